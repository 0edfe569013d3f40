#!/usr/bin/env python3
"""Where the time of NBMF(...).fit goes on the configs[2] workload besides the iterations: input validation
(sklearn's check_array, as the reference does), upload + pack, factor download."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sklearn.utils import check_array
from nbmf_mm_amd import _hip, _dist
M, N, K = 65536, 8192, 64
g = np.random.default_rng(0)
X = (g.random((M, N)) < 0.25).astype(np.float64)
Mk = g.random((M, N)) < 0.9
W, H = _dist.global_init(M, N, K, random_state=0)
for rep in range(3):
    t0 = time.perf_counter(); check_array(X, accept_sparse="csr", dtype=np.float64); t1 = time.perf_counter()
    with _hip.Context(M, N, K) as ctx:
        t2 = time.perf_counter(); ctx.set_hyper(1.2, 1.2); ctx.upload(X, mask=Mk); ctx.synchronize(); t3 = time.perf_counter()
        ctx.set_factors(W, H); ctx.synchronize(); t4 = time.perf_counter()
        ctx.run(50, 0.0); t5 = time.perf_counter()
        ctx.get_factors(); t6 = time.perf_counter()
    print(f"check_array {t1-t0:.3f} s | create {t2-t1:.3f} | upload+pack {t3-t2:.3f} ({(X.nbytes+Mk.nbytes)/(t3-t2)/1e9:.1f} GB/s) | set_factors {t4-t3:.3f} | "
          f"run(50) {t5-t4:.3f} | get_factors {t6-t5:.3f}", flush=True)
