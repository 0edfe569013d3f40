import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/
import numpy as np
from nbmf_mm_amd import _hip
from oracle import nbmf_oracle as orc
from conftest import config1_X, config1_mask
g = np.load("tests/golden/transform.npz")
X, mask = config1_X(), config1_mask().astype(float)
H = g["H"]
np.random.seed(5)
W0 = np.random.uniform(0.1, 0.9, (100, 6))
with _hip.Context(100, 500, 6) as ctx:
    ctx.set_hyper(1.2, 1.2)
    ctx.upload(X, mask=mask)
    ctx.set_factors(np.ascontiguousarray(W0.T), H)
    Wr = W0
    for it in range(50):
        ctx.w_only_steps(1)
        Wk, _ = ctx.get_factors()
        # oracle single step without the final clip
        Wt = Wr.T
        th = H.T @ Wt
        Wt = Wt * (H @ ((X.T * mask.T) / (th + 1e-8)) + (1 - H) @ (((1 - X).T * mask.T) / (1 - th + 1e-8)))
        Wt = Wt / 500
        Wt = Wt / Wt.sum(axis=0, keepdims=True)
        Wr = Wt.T
        d = np.abs(Wk.T - Wr)
        bad = np.where(d.max(axis=1) > 1e-9)[0]
        print(it, "maxdiff %.3e" % d.max(), "bad rows", bad[:10], "min W ref %.3e gpu %.3e" % (Wr.min(), Wk.min()))
        if len(bad) and it > 45:
            i = bad[0]; print(" ref", Wr[i], "\n gpu", Wk.T[i])
