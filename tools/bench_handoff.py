#!/usr/bin/env python3
"""Round 6: what the memory model's release / acquire costs the single-launch engine's hand-off, and what the end-of-run
guard costs: iterations per second in the loop with NBMF_SMALL_FENCED=0 / 1 (the guard off, so that only the hand-off
differs), then a 200-iteration fit END TO END with the guard on and off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import _hip, _dist
its = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
os.environ["NBMF_PERSISTENT"] = "1"
os.environ["NBMF_SMALL_GUARD"] = "0"          # read once per process: off for the loop timings ...
first = True
for name, m, n, k in (("configs[0] 100x500 K=6", 100, 500, 6), ("animals 50x85 K=4", 50, 85, 4), ("lastfm 1226x285 K=8", 1226, 285, 8),
                      ("paleo 253x902 K=16", 253, 902, 16), ("2000x2000 K=16 (split strips)", 2000, 2000, 16)):
    X = (np.random.default_rng(0).random((m, n)) < 0.25).astype(np.float64)
    W, H = _dist.global_init(m, n, k, random_state=0)
    row = {}
    for fenced in ("0", "1"):
        os.environ["NBMF_SMALL_FENCED"] = fenced
        with _hip.Context(m, n, k) as ctx:
            ctx.set_hyper(1.2, 1.2); ctx.upload(X); ctx.set_factors(W, H); ctx.run(50, 0.0)
            if first:
                ctx.set_factors(W, H); ctx.run(20000, 0.0); first = False
            best = 0.0
            for _ in range(3):
                ctx.set_factors(W, H)
                t0 = time.perf_counter(); l, nit = ctx.run(its if m < 2000 else its // 10, 0.0); dt = time.perf_counter() - t0
                best = max(best, nit / dt)
            row[fenced] = (best, l[-1], ctx.small_stats())
    a, b = row["0"], row["1"]
    print(f"{name:32s} sc1 form {a[0]:9.0f} it/s ({1e6 / a[0]:6.2f} us/it)   fenced {b[0]:9.0f} it/s ({1e6 / b[0]:6.2f} us/it)   "
          f"fenced / sc1 time {a[0] / b[0]:.3f}   same loss bits {a[1] == b[1]}   runs/aborted {a[2]} {b[2]}", flush=True)
