#!/usr/bin/env python3
"""Round 6: the end-of-run guard of the single-launch engine, end to end: a 200-iteration fit through the ctypes layer
(set_factors + run) with the guard on (default) or off (NBMF_SMALL_GUARD=0, given on the command line's environment)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import _hip, _dist
os.environ["NBMF_PERSISTENT"] = "1"
for name, m, n, k in (("configs[0] 100x500 K=6", 100, 500, 6), ("lastfm 1226x285 K=8", 1226, 285, 8)):
    X = (np.random.default_rng(0).random((m, n)) < 0.25).astype(np.float64)
    W, H = _dist.global_init(m, n, k, random_state=0)
    with _hip.Context(m, n, k) as ctx:
        ctx.set_hyper(1.2, 1.2); ctx.upload(X); ctx.set_factors(W, H); ctx.run(5000, 0.0)
        ts = []
        for _ in range(200):
            ctx.set_factors(W, H)
            t0 = time.perf_counter(); ctx.run(200, 0.0); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"guard {os.environ.get('NBMF_SMALL_GUARD', '1')}: {name:28s} 200-iteration run: median {1e3 * ts[len(ts) // 2]:.3f} ms, best {1e3 * ts[0]:.3f} ms, "
              f"runs/aborted {ctx.small_stats()}", flush=True)
