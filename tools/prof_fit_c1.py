import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from nbmf_mm_amd import _hip, _dist, NBMF
X = (np.random.default_rng(0).random((100, 500)) < 0.25).astype(np.float64)
W, H = _dist.global_init(100, 500, 6, random_state=0)
def t(f, *a):
    t0 = time.perf_counter(); r = f(*a); return (time.perf_counter() - t0) * 1e3, r
for rep in range(3):
    a, ctx = t(lambda: _hip.Context(100, 500, 6))
    b, _ = t(lambda: ctx.set_hyper(1.2, 1.2))
    c, _ = t(lambda: ctx.upload(X))
    d, _ = t(lambda: ctx.set_factors(W, H))
    e, _ = t(lambda: ctx.run(200, 0.0))
    f, _ = t(lambda: ctx.get_factors())
    g, _ = t(lambda: ctx.close())
    print("create %.2f upload %.2f set_factors %.2f run(200) %.2f get %.2f close %.2f ms" % (a, c, d, e, f, g))
t0 = time.perf_counter(); NBMF(n_components=6, max_iter=200, tol=0, random_state=0).fit(X); print("estimator fit %.2f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); NBMF(n_components=6, max_iter=200, tol=0, random_state=0).fit(X); print("estimator fit %.2f ms" % ((time.perf_counter() - t0) * 1e3))
