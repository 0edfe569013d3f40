#!/usr/bin/env python3
"""Time the BASELINE.json single-GPU configs through the estimator API (fit only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import NBMF

def run(name, X, mask, reps=2, **kw):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        m = NBMF(**kw).fit(X, mask=mask)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{name}: fit {best:.3f} s for {m.n_iter_} its = {m.n_iter_/best:.1f} it/s (incl. upload/pack/download), final loss {m.loss_:.15f}", flush=True)

X = (np.random.default_rng(0).random((100, 500)) < 0.25).astype(np.float64)
run("c1 100x500 K=6 200 its", X, None, n_components=6, orientation="beta-dir", alpha=1.2, beta=1.2, random_state=0, max_iter=200, tol=0)
run("c1 default stop rule   ", X, None, n_components=6, random_state=0)
X = (np.random.default_rng(0).random((8192, 8192)) < 0.25).astype(np.float64)
run("c2 8192x8192 K=32 500 its", X, None, reps=1, n_components=32, random_state=0, max_iter=500, tol=0)
