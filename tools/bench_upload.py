#!/usr/bin/env python3
"""Host -> device upload + pack of the c3 workload (4.3 GB float64 + 0.5 GB bool), repeated; with and without
NBMF_NO_HOST_REGISTER (kept from an experiment: page-locking the arrays first bought 10 %, not worth it)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nbmf_mm_amd import _hip
M, N, K = 65536, 8192, 64
g = np.random.default_rng(0)
X = (g.random((M, N)) < 0.25).astype(np.float64)
Mk = g.random((M, N)) < 0.9
with _hip.Context(M, N, K) as ctx:
    for rep in range(3):
        t0 = time.perf_counter(); ctx.upload(X, mask=Mk); print("upload %.3f s" % (time.perf_counter() - t0), flush=True)
    os.environ["NBMF_NO_HOST_REGISTER"] = "1"
    t0 = time.perf_counter(); ctx.upload(X, mask=Mk); print("upload (no register) %.3f s" % (time.perf_counter() - t0), flush=True)
