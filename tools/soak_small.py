#!/usr/bin/env python3
"""Soak of the single-launch path's hand-offs: the same fit repeated many times must give the same bits every time
(a stale or torn read of another workgroup's data would show as a difference), and agree with the five-kernel path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import _hip, _dist
reps, its = int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 20000
bad = 0
for (m, n, k, real) in [(100, 500, 6, False), (1226, 285, 8, False), (253, 902, 8, True), (1024, 1024, 32, False), (2000, 2000, 16, False)]:
    r = np.random.default_rng(1)
    X = r.random((m, n)) if real else (r.random((m, n)) < 0.25).astype(np.float64)
    mask = r.random((m, n)) < 0.9
    W, H = _dist.global_init(m, n, k, random_state=0)
    n_it = max(200, its * 100 * 500 // (m * n))            # fewer iterations for the bigger shapes
    os.environ["NBMF_PERSISTENT"] = "0"
    with _hip.Context(m, n, k) as ctx:
        ctx.set_hyper(1.2, 1.2); ctx.upload(X, mask=mask); ctx.set_factors(W, H)
        lref, _ = ctx.run(n_it, 0.0)
    os.environ["NBMF_PERSISTENT"] = "1"
    first, t0 = None, time.perf_counter()
    with _hip.Context(m, n, k) as ctx:
        ctx.set_hyper(1.2, 1.2); ctx.upload(X, mask=mask)
        for rep in range(reps):
            ctx.set_factors(W, H)
            l, _ = ctx.run(n_it, 0.0)
            out = (l.tobytes(),) + tuple(a.tobytes() for a in ctx.get_factors())
            if first is None:
                first = out
            elif out != first:
                bad += 1
                print(f"  {m}x{n} K={k}: repetition {rep} differs from the first", flush=True)
        stats = ctx.small_stats()
    rel = float(np.nanmax(np.abs(l - lref) / np.abs(lref)))
    print(f"{m}x{n} K={k} real={real}: {reps} x {n_it} iterations in {time.perf_counter() - t0:.1f} s, runs/aborted {stats}, "
          f"max rel loss diff to the five-kernel path {rel:.1e}", flush=True)
    bad += rel > 1e-11 or stats[1] != 0
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
