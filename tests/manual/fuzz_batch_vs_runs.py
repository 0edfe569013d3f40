#!/usr/bin/env python3
"""Randomised sweep of `nbmf_run_batch` against one `nbmf_run` per problem (hand-run on a GPU box:
`python tests/manual/fuzz_batch_vs_runs.py [cases] [seed] [max_dim]`).

The batch call is what the reference's experiment loops become (`examples/reproduce_magron2022.py:49-73`: a grid of
priors, restarts): P independent fits of one data set.  Its promise (`include/nbmf_hip.h`) is problem p's results
BITWISE those of set_hyper + set_factors + run + get_factors.  Every case draws a shape (small enough for the batched
launch, or not: the call then runs the problems one after the other), K, data and mask kinds, the projection, P in 1..9
with its own (alpha, beta, W0, H0) each, an iteration count and a stop rule that fires at different iterations for
different problems, and compares loss curves, iteration counts and factors bit for bit."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from nbmf_mm_amd import _hip


def run(cases, seed, max_dim=1500):
    r = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for case in range(cases):
        m = int(r.integers(1, max_dim))
        n = int(r.integers(1, max_dim))
        k = int(r.choice([1, 2, 4, 6, 8, 10, 16, 20, 32, 40, 64]))
        real = r.random() < 0.3
        Y = r.random((m, n)) if real else (r.random((m, n)) < r.uniform(0.05, 0.9)).astype(np.float64)
        mk = r.choice(["none", "bool", "weights"], p=[0.4, 0.45, 0.15])
        mask = None if mk == "none" else ((r.random((m, n)) < r.uniform(0.3, 0.99)) if mk == "bool" else r.uniform(0.05, 1.0, (m, n)))
        P = int(r.integers(1, 10))
        alphas = r.uniform(1.0, 2.0, P)
        betas = r.uniform(1.0, 2.0, P)
        W0 = r.uniform(0.05, 0.95, (P, k, m))
        W0 /= W0.sum(axis=1, keepdims=True)
        H0 = r.uniform(0.05, 0.95, (P, k, n))
        iters = int(r.integers(1, 40))
        tol = float(r.choice([0.0, 1e-3, 1e-2]))
        proj = int(r.random() < 0.25)
        tag = f"m={m} n={n} k={k} real={real} mask={mk} P={P} its={iters} tol={tol} proj={proj}"
        try:
            with _hip.Context(m, n, k) as ctx:
                ctx.set_hyper(1.2, 1.2, 1e-8, proj)
                ctx.upload(Y, mask)
                lb, nb, Wb, Hb = ctx.run_batch(alphas, betas, W0, H0, iters, tol)
                launches, served = ctx.batch_stats()
                ok, why = True, ""
                for p in range(P):
                    ctx.set_hyper(float(alphas[p]), float(betas[p]), 1e-8, proj)
                    ctx.set_factors(W0[p], H0[p])
                    l, it = ctx.run(iters, tol)
                    W, H = ctx.get_factors()
                    if it != nb[p] or not np.array_equal(l, lb[p], equal_nan=True):
                        ok, why = False, f"problem {p}: n_iter {it} / {nb[p]}, losses differ by {np.nanmax(np.abs(np.asarray(l)[:min(it, nb[p])] - lb[p][:min(it, nb[p])])) if min(it, nb[p]) else float('nan'):.2e}"
                        break
                    if not (np.array_equal(W, Wb[p], equal_nan=True) and np.array_equal(H, Hb[p], equal_nan=True)):
                        ok, why = False, f"problem {p}: factors differ by {np.nanmax(np.abs(W - Wb[p])):.2e} / {np.nanmax(np.abs(H - Hb[p])):.2e}"
                        break
        except Exception as e:   # noqa: BLE001
            ok, why, launches, served = False, f"EXCEPTION {e!r}", -1, -1
        if not ok:
            bad += 1
            print(f"case {case}: {why}  (batched launches {launches}, problems they served {served})  {tag}", flush=True)
        if case % 50 == 49:
            print(f"  ... {case + 1} cases, {bad} failures, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    return bad, time.time() - t0


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n_bad, secs = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 1500)
    print(f"{n_cases} cases, {n_bad} failures, {secs:.0f} s")
    sys.exit(1 if n_bad else 0)
