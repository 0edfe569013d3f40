#!/usr/bin/env python3
"""Mid-size shapes on the single-launch path (split strips) against the five-kernel path: it/s in the loop and the
agreement of the two engines' loss curves."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import _hip, _dist
its = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for (m, n, k) in [(2000, 2000, 16), (2048, 2048, 32), (1500, 2000, 8), (2048, 1024, 16), (4000, 300, 8), (300, 2040, 12), (4080, 2040, 16)]:
    X = (np.random.default_rng(0).random((m, n)) < 0.25).astype(np.float64)
    W, H = _dist.global_init(m, n, k, random_state=0)
    row = []
    for mode in ("1", "0"):
        os.environ["NBMF_PERSISTENT"] = mode
        with _hip.Context(m, n, k) as ctx:
            ctx.set_hyper(1.2, 1.2); ctx.upload(X); ctx.set_factors(W, H); ctx.run(20, 0.0); ctx.set_factors(W, H)
            t0 = time.perf_counter(); l, nit = ctx.run(its, 0.0); dt = time.perf_counter() - t0
            row.append((nit / dt, l, ctx.small_stats()))
    rel = float(np.max(np.abs(row[0][1] - row[1][1]) / np.abs(row[1][1])))
    print(f"{m}x{n} K={k}: single launch {row[0][0]:.0f} it/s {row[0][2]}   five kernels {row[1][0]:.0f} it/s   max rel loss diff {rel:.1e}", flush=True)
