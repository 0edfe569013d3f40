#!/usr/bin/env python3
"""Long runs (hand-run on a GPU box): 6000 iterations of BASELINE configs[2] and 3000 of a real-valued 16384 x 8192 matrix --
monotone and finite all the way, and a shorter run from the same start is a bitwise prefix.  (Round 4: 202.9 it/s over the
6000 iterations, every loss assembled inside the next sweep; no wait ever expired.)"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import numpy as np
from bench import make_shard, init_factors
from nbmf_mm_amd import _hip
M, N, K = 65536, 8192, 64
X, Mk = make_shard(M, N, 0, M, 0)
W, H = init_factors(M, N, K, 0)
with _hip.Context(M, N, K) as ctx:
    ctx.set_hyper(1.2, 1.2, 1e-8, 0)
    ctx.upload(X, mask=Mk)
    ctx.set_factors(W, H)
    t0 = time.perf_counter()
    l, n = ctx.run(6000, 0.0)
    dt = time.perf_counter() - t0
    l = np.array(l)
    print("c3: %d iterations in %.1f s (%.1f it/s), monotone %s, finite %s, loss %.12f -> %.12f" % (n, dt, n / dt, bool(np.all(l[1:] <= l[:-1] + 1e-12)), bool(np.all(np.isfinite(l))), l[0], l[-1]))
    ctx.set_factors(W, H)
    l2, _ = ctx.run(300, 0.0)
    print("prefix bitwise:", bool(np.array_equal(l2, l[:300])))
# general path soak
Xr = np.random.default_rng(1).random((16384, 8192))
W, H = init_factors(16384, N, K, 0)
with _hip.Context(16384, N, K) as ctx:
    ctx.upload(Xr, mask=Mk[:16384])
    ctx.set_factors(W, H)
    l, n = ctx.run(3000, 0.0)
    l = np.array(l)
    print("general 16384 rows: %d iterations, monotone %s, finite %s, loss %.12f -> %.12f" % (n, bool(np.all(l[1:] <= l[:-1] + 1e-12)), bool(np.all(np.isfinite(l))), l[0], l[-1]))
