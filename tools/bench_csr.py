#!/usr/bin/env python3
"""Sparse upload at a size whose dense float64 form would not be practical on the host: V 360000 x 17000 (the
configs[4] shape) with 2 % ones given as CSR (122 M stored entries, 0.5 GB of index arrays instead of 49 GB dense),
dir-beta, K = 64: time of nbmf_upload_csr and of a few iterations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import _hip, _dist
M, N, K, per_row = 360000, 17000, 64, 340
r = np.random.default_rng(0)
t0 = time.perf_counter()
idx = r.integers(0, N, size=(M, per_row), dtype=np.int32)
idx.sort(axis=1)
keep = np.ones(idx.shape, dtype=bool)
keep[:, 1:] = idx[:, 1:] != idx[:, :-1]                       # drop duplicate columns within a row
indptr = np.zeros(M + 1, dtype=np.int64)
np.cumsum(keep.sum(axis=1), out=indptr[1:])
indices = idx[keep]
print(f"pattern: {indices.size / 1e6:.1f} M entries built in {time.perf_counter() - t0:.1f} s", flush=True)
W, H = _dist.global_init(N, M, K, random_state=0)            # dir-beta: internal matrix is V^T (17000 x 360000)
with _hip.Context(N, M, K) as ctx:
    ctx.set_hyper(1.2, 1.2)
    t0 = time.perf_counter()
    ctx.upload_csr((indptr, indices), None, transposed=True)
    print(f"upload_csr: {time.perf_counter() - t0:.2f} s, n_obs {ctx.n_obs():.0f}", flush=True)
    ctx.set_factors(W, H)
    ctx.run(1, 0.0)
    t0 = time.perf_counter()
    losses, n = ctx.run(5, 0.0)
    dt = time.perf_counter() - t0
    print(f"{n / dt:.2f} it/s ({1e3 * dt / n:.1f} ms per iteration), loss {losses[-1]:.6f}, monotone {all(np.diff(losses) <= 1e-12)}")
