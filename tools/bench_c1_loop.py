#!/usr/bin/env python3
"""Small problems through the ctypes layer, iterations per second in the loop: BASELINE configs[0] (100 x 500, K = 6)
and the shapes of the reference's own datasets (examples/reproduce_magron2022.py:49-73), by the single-launch
path (nbmf_small_kernel.inc) and by the five-kernel path (NBMF_PERSISTENT=0)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import _hip, _dist
from bench_c1_loop_cases import CASES
its = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
only = sys.argv[2] if len(sys.argv) > 2 else ""
warm = True
for name, m, n, k in [c for c in CASES if only in c[0]]:
    X = (np.random.default_rng(0).random((m, n)) < 0.25).astype(np.float64)
    W, H = _dist.global_init(m, n, k, random_state=0)
    row = []
    for mode in ("1", "0"):
        os.environ["NBMF_PERSISTENT"] = mode
        with _hip.Context(m, n, k) as ctx:
            ctx.set_hyper(1.2, 1.2)
            ctx.upload(X)
            ctx.set_factors(W, H)
            ctx.run(50, 0.0)
            if warm:   # the first long launch of a process can run ~15 % slower than every later one (clock state)
                ctx.set_factors(W, H); ctx.run(20000, 0.0); warm = False
            ctx.set_factors(W, H)
            t0 = time.perf_counter(); losses, nit = ctx.run(its, 0.0); dt = time.perf_counter() - t0
            row.append((nit / dt, 1e6 * dt / nit, losses[min(199, nit - 1)], ctx.small_stats()))
    (a, ua, la, sa), (b, ub, lb, sb) = row
    print(f"{name:24s} single launch {a:9.0f} it/s ({ua:6.2f} us/it, runs/aborted {sa})   five kernels {b:9.0f} it/s ({ub:6.2f} us/it)   "
          f"loss[199] {la:.15f} vs {lb:.15f}", flush=True)
