#!/usr/bin/env python3
"""Time the general (real-valued / weighted) storage path against the binary one on the same shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import _hip
M, N, K = 16384, 8192, 64
g = np.random.default_rng(0)
Xb = (g.random((M, N)) < 0.25).astype(np.float64)
Xr = g.random((M, N))
Wt = g.random((M, N))
Mb = g.random((M, N)) < 0.9
np.random.seed(0)
W0 = np.random.uniform(0.1, 0.9, (K, M)); W0 /= W0.sum(axis=0, keepdims=True)
H0 = np.random.uniform(0.1, 0.9, (K, N))
for name, X, mk in [("binary+boolmask", Xb, Mb), ("real, no mask", Xr, None), ("real+boolmask", Xr, Mb), ("real+0/1 f64 mask", Xr, Mb.astype(np.float64)), ("real+weights", Xr, Wt), ("binary+weights", Xb, Wt)]:
    with _hip.Context(M, N, K) as ctx:
        ctx.set_hyper(1.2, 1.2)
        binp = ctx.upload(X, mask=mk)
        ctx.set_factors(W0, H0)
        ctx.run(2, 0.0)
        ctx.timing_enable(True)
        t0 = time.perf_counter(); losses, _ = ctx.run(10, 0.0); dt = time.perf_counter() - t0
        t = ctx.timing()
        print(f"{name:18s} binary_path={binp} {10/dt:8.1f} it/s  hpass {t['hpass_ms']/t['hpass_launches']:.3f} ms wpass {t['wpass_ms']/t['wpass_launches']:.3f} ms loss {losses[-1]:.12f}", flush=True)
