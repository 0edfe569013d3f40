#!/usr/bin/env python3
"""The likelihood-only sweep (pass_kernel<MODE_L>: nbmf_loss / score / the closing loss of a fit) on real-valued V at
K = 8 and K = 16 -- the kernels that carried scratch until round 5 -- timed per call.  Library: NBMF_HIP_LIBRARY."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import _hip
M, N = 16384, 8192
g = np.random.default_rng(0)
Xr = g.random((M, N))
Mb = g.random((M, N)) < 0.9
for K in (8, 16, 32):
    np.random.seed(0)
    W0 = np.random.uniform(0.1, 0.9, (K, M)); W0 /= W0.sum(axis=0, keepdims=True)
    H0 = np.random.uniform(0.1, 0.9, (K, N))
    for name, mk in (("no mask", None), ("bool mask", Mb)):
        with _hip.Context(M, N, K) as ctx:
            ctx.set_hyper(1.2, 1.2)
            ctx.upload(Xr, mask=mk)
            ctx.set_factors(W0, H0)
            ctx.run(2, 0.0)
            for fn_name in ("loss", "loglik_strict"):
                fn = getattr(ctx, fn_name)
                v = fn(); ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    v = fn()
                ctx.synchronize()
                dt = (time.perf_counter() - t0) / 50
                print(f"K={K:3d} real-valued V {M}x{N}, {name:9s} {fn_name:14s} {dt*1e3:8.3f} ms per call  value {v:.15g}", flush=True)
