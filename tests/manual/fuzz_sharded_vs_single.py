#!/usr/bin/env python3
"""Randomised sweep of the in-process sharded fit against the single-context fit (hand-run on a GPU box:
`GPU_MAX_HW_QUEUES=32 python tests/manual/fuzz_sharded_vs_single.py [cases] [seed] [max_dim]`).

Every case draws a shape (down to fewer rows than ranks), K, 2-6 ranks (all on device 0: rank threads, the peer
transport between their arenas, or the host transport), data kind (binary / real / real on SOME shards only), mask kind,
orientation, projection, hyper-parameters, iteration count and a stop rule, runs `fit_in_process` (what `nbmf_mm_solver(..., n_gpus=R)` calls) and
`nbmf_mm_solver(...)` on the same inputs and compares: same number of iterations, losses rtol 1e-10, factors atol 1e-9
(the sums over ranks differ in order only: DESIGN.md 6).  Prints one line per failure and a summary."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from nbmf_mm_amd import nbmf_mm_solver
from nbmf_mm_amd._dist import fit_in_process


def run(cases, seed, max_dim=900):
    r = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for case in range(cases):
        ranks = int(r.integers(2, 7))
        m = int(r.integers(1, max_dim))
        n = int(r.integers(1, max_dim))
        k = int(r.choice([1, 3, 6, 10, 16, 20, 32, 40, 64, 100, 128, 130]))
        kind = r.choice(["binary", "real", "mixed"], p=[0.55, 0.25, 0.2])
        Y = (r.random((m, n)) < r.uniform(0.05, 0.9)).astype(np.float64)
        if kind == "real":
            Y = r.random((m, n))
        elif kind == "mixed":                       # real values on some row blocks only: the ranks choose different storage
            lo = int(r.integers(0, m))
            hi = int(r.integers(lo, m + 1))
            Y[lo:hi] = r.random((hi - lo, n))
        mk = r.choice(["none", "bool", "weights"], p=[0.4, 0.45, 0.15])
        mask = None if mk == "none" else ((r.random((m, n)) < r.uniform(0.3, 0.99)) if mk == "bool" else r.uniform(0.05, 1.0, (m, n)))
        kw = dict(max_iter=int(r.integers(1, 20)), tol=float(r.choice([0.0, 1e-4, 1e-3])), mask=mask, alpha=float(r.uniform(1.0, 2.0)),
                  beta=float(r.uniform(1.0, 2.0)), orientation=str(r.choice(["beta-dir", "dir-beta"])),
                  projection=str(r.choice(["normalize", "duchi"], p=[0.75, 0.25])))
        init = r.choice(["seed", "custom"], p=[0.6, 0.4])
        if init == "seed":
            kw["random_state"] = int(r.integers(0, 1000))
        else:
            W0 = r.uniform(0.05, 0.95, (m, k))
            H0 = r.uniform(0.05, 0.95, (k, n))
            if kw["orientation"] == "beta-dir":
                W0 /= W0.sum(axis=1, keepdims=True)
            else:
                H0 /= H0.sum(axis=0, keepdims=True)
            kw["W_init"], kw["H_init"] = W0, H0
        transport = str(r.choice(["auto", "host"], p=[0.7, 0.3]))
        tag = f"{transport} m={m} n={n} k={k} ranks={ranks} data={kind} mask={mk} init={init} {kw['orientation']} {kw['projection']} its={kw['max_iter']} tol={kw['tol']}"
        if m < ranks:                               # fewer rows than ranks: refused, cleanly
            try:
                fit_in_process(Y, k, ranks, devices=[0] * ranks, transport=transport, **kw)
                bad += 1
                print(f"case {case}: no ValueError  {tag}", flush=True)
            except ValueError as e:
                assert "cannot shard" in str(e)
            continue
        try:
            with np.errstate(all="ignore"):
                W1, H1, l1, _, n1 = nbmf_mm_solver(Y, k, **kw)
                W2, H2, l2, _, n2 = fit_in_process(Y, k, ranks, devices=[0] * ranks, transport=transport, **kw)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"case {case}: EXCEPTION {e!r}  {tag}", flush=True)
            continue
        l1, l2 = np.asarray(l1), np.asarray(l2)
        ok = n1 == n2 and l1.shape == l2.shape and np.array_equal(np.isnan(l1), np.isnan(l2))
        fin = ~np.isnan(l1)
        if ok and fin.any():
            ok = np.allclose(l2[fin], l1[fin], rtol=1e-10, atol=0)
        if ok and fin.all() and np.all(np.isfinite(W1)) and np.all(np.isfinite(H1)):
            ok = np.allclose(W2, W1, rtol=0, atol=1e-9) and np.allclose(H2, H1, rtol=0, atol=1e-9)
        if not ok:
            bad += 1
            dl = np.max(np.abs(l2[fin] / l1[fin] - 1)) if (l1.shape == l2.shape and fin.any()) else float("nan")
            print(f"case {case}: MISMATCH n_iter {n1} / {n2} rel loss {dl:.2e}  {tag}", flush=True)
        if case % 25 == 24:
            print(f"  ... {case + 1} cases, {bad} failures, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    return bad, time.time() - t0


if __name__ == "__main__":
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 12:
        sys.exit("start with GPU_MAX_HW_QUEUES=32: up to six ranks share the one GPU")
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n_bad, secs = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 900)
    print(f"{n_cases} cases, {n_bad} failures, {secs:.0f} s")
    sys.exit(1 if n_bad else 0)
