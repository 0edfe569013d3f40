"""Four mid-size fits (millions of entries, K = 32 ... 128: real-valued with weights, binary with a mask, real, binary with\nthe Duchi projection) against the oracle: the figures it prints are the largest relative loss difference and the largest\nfactor differences after 6 iterations (round 3, final build: 2e-15 / 1e-16 / 1e-15).  Hand-run on a GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nbmf_mm_amd import nbmf_mm_solver
from oracle import nbmf_oracle as orc
r = np.random.default_rng(0)
for (m, n, k, kind) in [(3000, 5000, 64, "real+weights"), (4000, 3000, 32, "binary+mask"), (2500, 6000, 128, "real"), (5000, 2000, 64, "binary duchi")]:
    if kind.startswith("real"):
        Y = r.random((m, n))
    else:
        Y = (r.random((m, n)) < 0.3).astype(np.float64)
    mask = r.random((m, n)) if "weights" in kind else ((r.random((m, n)) < 0.85) if "mask" in kind else None)
    duchi = "duchi" in kind
    kw = dict(max_iter=6, tol=0, random_state=3, mask=mask, alpha=1.3, beta=1.1)
    t0 = time.time(); Wr, Hr, lr, _, _ = orc.solve(Y, k, step=orc.mm_step_duchi if duchi else None, **kw); t1 = time.time()
    W, H, l, _, _ = nbmf_mm_solver(Y, k, projection="duchi" if duchi else "normalize", **kw); t2 = time.time()
    print(f"{m}x{n} K={k} {kind}: rel loss {np.max(np.abs(np.asarray(l)/np.asarray(lr)-1)):.2e}  max|dW| {np.max(np.abs(W-Wr)):.2e} max|dH| {np.max(np.abs(H-Hr)):.2e}  (oracle {t1-t0:.1f} s, hip {t2-t1:.1f} s)", flush=True)
