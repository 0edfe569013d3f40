#!/usr/bin/env python3
"""Does the single-launch path's speed depend on where its buffers sit?  Several contexts of the same problem alive at
once (distinct allocations), each timed in turn.  With NBMF_SMALL_PLACEMENT=1 the library also prints where the
dispatcher put each workgroup (XCD.SE.CU).  Found: the 11.5 / 13.3 us modes of configs[0] follow neither the allocation nor
the placement (same pattern at both speeds); a launch keeps one speed for its whole duration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import _hip, _dist
m, n, k = 100, 500, 6
X = (np.random.default_rng(0).random((m, n)) < 0.25).astype(np.float64)
W, H = _dist.global_init(m, n, k, random_state=0)
ctxs = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    ctx = _hip.Context(m, n, k)
    ctx.set_hyper(1.2, 1.2)
    ctx.upload(X)
    ctxs.append(ctx)
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    row = []
    for ctx in ctxs:
        ctx.set_factors(W, H)
        ctx.run(50, 0.0)
        ctx.set_factors(W, H)
        t0 = time.perf_counter(); losses, nit = ctx.run(5000, 0.0); dt = time.perf_counter() - t0
        row.append(1e6 * dt / nit)
    print("us/it per context:", " ".join("%6.2f" % v for v in row), flush=True)
for ctx in ctxs:
    ctx.close()
