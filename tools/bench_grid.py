#!/usr/bin/env python3
"""Throughput of the experiment driver on a reference-sized problem (lastfm: 1226 x 285, K=8, the 36-point
(alpha, beta) grid of examples/reproduce_magron2022.py run_figure1, 500 iterations each): the grid points go to the
library as ONE batched call per K (nbmf_run_batch: as many fits at a time as the chip holds, one persistent launch per
group); `concurrency` adds host threads on top (round 2's way of overlapping the fits)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import _hip
from nbmf_mm_amd.experiments import perplexity_grid, _init
r = np.random.default_rng(0)
Y = (r.random((1226, 285)) < 0.1).astype(np.float64)
u = r.random(Y.shape)
train, val = (u < 0.7).astype(np.float64), ((u >= 0.7) & (u < 0.85)).astype(np.float64)
grid = [0.5, 1.0, 1.5, 2.0, 2.5, 3.0]
perplexity_grid(Y, train, {"val": val}, 8, grid[:2], grid[:2], max_iter=20, tol=0)     # warm-up: code objects, pools
for conc in (1, 1, 2, 4):
    t0 = time.perf_counter()
    rows = perplexity_grid(Y, train, {"val": val}, 8, grid, grid, max_iter=500, tol=0, concurrency=conc)
    dt = time.perf_counter() - t0
    its = sum(r_["n_iter"] for r_ in rows)
    print(f"concurrency {conc}: {len(rows)} fits, {its} iterations in {dt:.3f} s = {its/dt:.0f} it/s "
          f"(reference: 409 s for this grid on its CPU, outputs/chauhan2025/figure1_lastfm_results.csv)", flush=True)
# where the time goes: the batched call alone, and one fit alone
W0, H0 = _init(1226, 285, 8, 12345)
al = [a for a in grid for _ in grid]
be = [b for _ in grid for b in grid]
with _hip.Context(1226, 285, 8) as ctx:
    ctx.set_hyper(1.0, 1.0, 1e-8)
    ctx.upload(Y, mask=train)
    ctx.run_batch(al[:3], be[:3], W0, H0, 20, 0.0)
    for P in (1, 2, 3, 6, 36):
        l0, p0 = ctx.batch_stats()
        t0 = time.perf_counter()
        ctx.run_batch(al[:P], be[:P], W0, H0, 500, 0.0)
        dt = time.perf_counter() - t0
        l1, p1 = ctx.batch_stats()
        print(f"run_batch of {P:2d} fits x 500 iterations: {dt*1e3:7.2f} ms in {l1-l0} launch(es) = {dt/500/max(1,l1-l0)*1e6:.1f} us per iteration of a launch", flush=True)
    ctx.set_factors(W0, H0)
    ctx.run(20, 0.0)
    t0 = time.perf_counter()
    ctx.run(500, 0.0)
    print(f"nbmf_run, one fit x 500 iterations: {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
