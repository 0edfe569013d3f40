#!/usr/bin/env python3
"""BASELINE configs[0] (100 x 500, K = 6) through the ctypes layer: 5000 iterations, for kernel-level profiles of
the latency-bound regime."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import _hip, _dist
X = (np.random.default_rng(0).random((100, 500)) < 0.25).astype(np.float64)
W, H = _dist.global_init(100, 500, 6, random_state=0)
with _hip.Context(100, 500, 6) as ctx:
    ctx.set_hyper(1.2, 1.2)
    ctx.upload(X)
    ctx.set_factors(W, H)
    ctx.run(50, 0.0)
    t0 = time.perf_counter(); losses, n = ctx.run(5000, 0.0); dt = time.perf_counter() - t0
    print(f"{n / dt:.0f} it/s, {1e6 * dt / n:.1f} us per iteration, loss {losses[199 - 50]:.15f}")
