#!/usr/bin/env python3
"""Host -> device upload + pack of the c3 workload handed over as uint8 V + bool mask (1.07 GB) and as float64 V + bool
mask (4.8 GB), repeated; NBMF_UPLOAD_TRACE=1 prints where each upload's time goes."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nbmf_mm_amd import _hip
M, N, K = 65536, 8192, 64
g = np.random.default_rng(0)
X8 = (g.random((M, N)) < 0.25).astype(np.uint8)
Mk = g.random((M, N)) < 0.9
with _hip.Context(M, N, K) as ctx:
    for rep in range(3):
        t0 = time.perf_counter(); ctx.upload(X8, mask=Mk); dt = time.perf_counter() - t0
        print("uint8 upload %.3f s = %.1f GB/s" % (dt, (X8.nbytes + Mk.nbytes) / dt / 1e9), flush=True)
X = X8.astype(np.float64)
with _hip.Context(M, N, K) as ctx:
    for rep in range(2):
        t0 = time.perf_counter(); ctx.upload(X, mask=Mk); dt = time.perf_counter() - t0
        print("float64 upload %.3f s = %.1f GB/s" % (dt, (X.nbytes + Mk.nbytes) / dt / 1e9), flush=True)
