"""Bring-up helper: nbmf_run_batch against the sequential calls, problem by problem (prints, asserts nothing)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nbmf_mm_amd import _hip
g = np.random.default_rng(1)
m, n, k, P = 60, 45, 4, 3
Y = (g.random((m, n)) < 0.3).astype(np.float64)
W0 = g.uniform(0.1, 0.9, (P, k, m)); W0 /= W0.sum(axis=1, keepdims=True)
H0 = g.uniform(0.1, 0.9, (P, k, n))
with _hip.Context(m, n, k) as ctx:
    ctx.set_hyper(1.2, 1.2, 1e-8)
    ctx.upload(Y)
    curves, nit, Ws, Hs = ctx.run_batch([1.2] * P, [1.2] * P, W0, H0, 30, 0.0)
    print("batch stats", ctx.batch_stats())
    for p in range(P):
        ctx.set_factors(W0[p], H0[p])
        l, ni = ctx.run(30, 0.0)
        W, H = ctx.get_factors()
        print(p, "batch", curves[p][-1], nit[p], "seq", l[-1], ni, "dW", np.abs(Ws[p] - W).max(), "dH", np.abs(Hs[p] - H).max(),
              "first losses", curves[p][0], l[0])
