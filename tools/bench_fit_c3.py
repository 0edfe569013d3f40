#!/usr/bin/env python3
"""End-to-end NBMF(...).fit on the c3 workload (what a user of the estimator pays: validation, upload, pack,
iterations, download), next to the time of the iterations alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import NBMF
M, N, K, its = 65536, 8192, 64, 50
g = np.random.default_rng(0)
X = (g.random((M, N)) < 0.25).astype(np.float64)
Mk = g.random((M, N)) < 0.9
for name, V in (("float64 V", X), ("uint8 V", X.astype(np.uint8))):
    for rep in range(2):
        t0 = time.perf_counter()
        m = NBMF(n_components=K, random_state=0, max_iter=its, tol=0, projection="duchi").fit(V, mask=Mk)
        dt = time.perf_counter() - t0
        print(f"{name}: fit {dt:.3f} s for {m.n_iter_} iterations ({its / 200.0:.3f} s of that in the iteration kernels), loss {m.loss_:.12f}", flush=True)
