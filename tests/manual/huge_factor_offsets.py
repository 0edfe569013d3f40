#!/usr/bin/env python3
"""Factors of more than 2^31 elements (hand-run on a GPU box: `python tests/manual/huge_factor_offsets.py [tall|wide|both]`).

BASELINE's largest configuration (360000 x 17000, K = 128) has a 46-million-element factor; a 288 GB device holds far
more.  Here the streamed factor of a sweep crosses the 2^31-element line -- 16 781 312 rows (or columns) x K = 128 =
2 147 991 936 doubles, 17 GB -- so every index into W (tall) or into H and the H sweep's slabs (wide) that was formed in 32
bits would show: device-generated data (regenerated on the host slice by slice), two iterations, then
  * the simplex factor's columns sum to 1 and the Beta factor stays in [eps, 1 - eps] EVERYWHERE,
  * the H update of a few columns and the W update of the first, some random and the LAST 256 rows of iteration 2,
    slice-exact against the oracle's arithmetic from the device's own state after iteration 1 (1e-12),
  * the loss of both iterations finite and falling.
Needs ~80 GB of host memory (the factors come back whole)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from nbmf_mm_amd import _hip

ALPHA, BETA, EPS = 1.2, 1.2, 1e-8


def h_slice(Ycols, Mcols, W0, H0cols):
    Ym = Ycols * Mcols
    theta = W0.T @ H0cols
    num = H0cols * (W0 @ (Ym / (theta + EPS))) + (ALPHA - 1)
    den = (1 - H0cols) * (W0 @ ((1 - Ym) / (1 - theta + EPS))) + (BETA - 1)
    return np.clip(num / (num + den + EPS), EPS, 1 - EPS)


def w_rows(Yrows, Mrows, W0rows, H1, n):
    th_t = H1.T @ W0rows
    Wn = W0rows * (H1 @ ((Yrows.T * Mrows.T) / (th_t + EPS)) + (1 - H1) @ (((1 - Yrows).T * Mrows.T) / (1 - th_t + EPS))) / n
    return Wn / Wn.sum(axis=0, keepdims=True)


def case(name, m, n, k, seed):
    t0 = time.time()
    g = np.random.default_rng(seed)
    W0 = g.random((k, m))
    W0 /= W0.sum(axis=0, keepdims=True)
    H0 = g.uniform(0.1, 0.9, (k, n))
    print(f"{name}: {m} x {n}, K = {k}: W {W0.size} elements ({W0.nbytes / 2**30:.1f} GiB), H {H0.size}; inits drawn in {time.time() - t0:.0f} s", flush=True)
    with _hip.Context(m, n, k) as ctx:
        ctx.set_hyper(ALPHA, BETA, EPS)
        ctx.generate(seed=seed, density=0.25, observed=0.9)
        ctx.set_factors(W0, H0)
        del W0, H0
        t1 = time.time()
        l1, _ = ctx.run(1, 0.0)
        W1, H1 = ctx.get_factors()
        l2, _ = ctx.run(1, 0.0)
        W2, H2 = ctx.get_factors()
        print(f"  two iterations + factors back in {time.time() - t1:.0f} s; losses {l1[0]:.12f} {l2[0]:.12f}", flush=True)
    assert np.isfinite(l1[0]) and np.isfinite(l2[0]) and l2[0] <= l1[0]
    s = W2.sum(axis=0)
    assert np.abs(s - 1.0).max() <= 1e-12, f"simplex: {np.abs(s - 1.0).max():.2e} at {int(np.abs(s - 1.0).argmax())}"
    assert H2.min() >= EPS and H2.max() <= 1 - EPS and np.isfinite(W2).all()
    cols = np.unique(np.concatenate([[0, n - 1], g.choice(n, 6, replace=False)]))
    Yc, Mc = _hip.synthetic_reference(m, n, seed, 0.25, 0.9, cols=cols)
    want = h_slice(Yc, Mc.astype(np.float64), W1, H1[:, cols])
    dH = np.abs(H2[:, cols] - want).max()
    del Yc, Mc
    nr = 256 if n <= 4096 else 16                   # (a row of a wide problem costs the host n doubles per temporary)
    rows = np.unique(np.concatenate([np.arange(nr), np.arange(m - nr, m), g.choice(m, nr, replace=False)]))
    Yr, Mr = _hip.synthetic_reference(m, n, seed, 0.25, 0.9, rows=rows)
    dW = np.abs(W2[:, rows] - w_rows(Yr, Mr.astype(np.float64), W1[:, rows], H2, n)).max()
    print(f"  H update of {cols.size} columns: max deviation {dH:.2e}; W update of {rows.size} rows (first, last, random): {dW:.2e}", flush=True)
    assert dH <= 1e-12 and dW <= 1e-12
    print(f"{name}: ok ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    big = (1 << 24) + 4096
    if which in ("tall", "both"):
        case("tall", big, 256, 128, 11)
    if which in ("wide", "both"):
        case("wide", 256, big, 128, 12)
