#!/usr/bin/env python3
"""Throughput of the experiment driver on a reference-sized problem (lastfm: 1226 x 285, K=8, the 36-point
(alpha, beta) grid of examples/reproduce_magron2022.py run_figure1, 500 iterations each)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd.experiments import perplexity_grid
r = np.random.default_rng(0)
Y = (r.random((1226, 285)) < 0.1).astype(np.float64)
u = r.random(Y.shape)
train, val = (u < 0.7).astype(np.float64), ((u >= 0.7) & (u < 0.85)).astype(np.float64)
grid = [0.5, 1.0, 1.5, 2.0, 2.5, 3.0]
for conc in (1, 2, 4, 8):
    t0 = time.perf_counter()
    rows = perplexity_grid(Y, train, {"val": val}, 8, grid, grid, max_iter=500, tol=0, concurrency=conc)
    dt = time.perf_counter() - t0
    its = sum(r_["n_iter"] for r_ in rows)
    print(f"concurrency {conc}: {len(rows)} fits, {its} iterations in {dt:.2f} s = {its/dt:.0f} it/s "
          f"(reference: 409 s for this grid on its CPU, outputs/chauhan2025/figure1_lastfm_results.csv)", flush=True)
