#!/usr/bin/env python3
"""Randomised sweep of the EVALUATION calls against the oracle (hand-run on a GPU box:
`python tests/manual/fuzz_eval_vs_oracle.py [cases] [seed] [max_dim]`).

fuzz_vs_oracle.py drives fits; this one puts random factors into a context -- on the simplex or off it, H inside
(0, 1) or beyond, so that both variants of the sweeps and the clipped evaluation are met -- over random shapes, K (ragged
or not), data kinds and mask kinds, and compares with the oracle

  * `nbmf_loss`            with `mm_loss` (the reference's loss, `_solver.py:148-162`),
  * `nbmf_loglik(clip=1)`  with `score` x n_obs (`_base.py:235-247`),
  * `nbmf_loglik_strict`   with log(perplexity) x -n_obs (`examples/reproduce_magron2022.py:40-47`),
  * `nbmf_w_only_steps(3)` with three iterations of the transform loop from the same start (`_base.py:170-199`, before
    its closing clip and renormalisation), on factors where that loop is stable (W on the simplex, Theta < 1),

relative 1e-10 on the sums, absolute 1e-9 on W.  Prints one line per failure and a summary."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from nbmf_mm_amd import _hip
from oracle import nbmf_oracle as orc


def _w_steps(X, W_kxm, H, mask, steps):
    """`steps` iterations of the loop in oracle.w_only_transform (same arithmetic, factors in the internal layout)."""
    Wt = W_kxm.copy()
    n = X.shape[1]
    for _ in range(steps):
        theta_t = H.T @ Wt
        yt = X.T if mask is None else X.T * mask.T
        zt = (1 - X).T if mask is None else (1 - X).T * mask.T
        Wt = Wt * (H @ (yt / (theta_t + 1e-8)) + (1 - H) @ (zt / (1 - theta_t + 1e-8)))
        Wt = Wt / n
        Wt = Wt / Wt.sum(axis=0, keepdims=True)
    return Wt


def run(cases, seed, max_dim=900):
    r = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for case in range(cases):
        m = int(r.integers(1, max_dim))
        n = int(r.integers(1, max_dim))
        k = int(r.choice([1, 3, 5, 8, 10, 16, 20, 24, 32, 40, 48, 64, 100, 128, 130]))
        real = r.random() < 0.35
        Y = r.random((m, n)) if real else (r.random((m, n)) < r.uniform(0.05, 0.9)).astype(np.float64)
        mk = r.choice(["none", "bool", "weights"], p=[0.35, 0.45, 0.2])
        mask = None if mk == "none" else ((r.random((m, n)) < r.uniform(0.3, 0.99)) if mk == "bool" else r.uniform(0.05, 1.0, (m, n)))
        maskf = None if mask is None else np.asarray(mask, dtype=np.float64)
        al, be = float(r.uniform(1.0, 2.0)), float(r.uniform(1.0, 2.0))
        state = r.choice(["fit_like", "H_beyond", "W_off_simplex"], p=[0.6, 0.2, 0.2])
        W = r.uniform(0.05, 0.95, (k, m))
        W /= W.sum(axis=0, keepdims=True)
        H = r.uniform(0.02, 0.98, (k, n))
        if state == "H_beyond":
            H = r.uniform(0.0, 1.6, (k, n))
        if state == "W_off_simplex":
            W = r.uniform(0.0, 0.9, (k, m))
        what = "?"
        try:
            with _hip.Context(m, n, k) as ctx:
                ctx.set_hyper(al, be, 1e-8)
                ctx.upload(Y, mask)
                ctx.set_factors(W, H)
                n_obs = ctx.n_obs()
                with np.errstate(all="ignore"):
                    want_loss = orc.mm_loss(Y, W, H, maskf, al, be)
                    want_clip = orc.score(Y, W.T, H, maskf) * (Y.size if mask is None else np.count_nonzero(mask))
                    want_strict = -np.log(orc.heldout_perplexity(Y, np.clip(W.T @ H, 0, 1) if state != "fit_like" else W.T @ H,
                                                                   maskf)) * (Y.size if mask is None else np.count_nonzero(mask))
                fails = []
                what = "loss"
                got = ctx.loss()
                if np.isfinite(want_loss) and not abs(got - want_loss) <= 1e-10 * abs(want_loss):
                    fails.append(f"loss {got!r} vs {want_loss!r}")
                what = "loglik(clip)"
                got = ctx.loglik(clip_theta=True)
                if np.isfinite(want_clip) and not abs(got - want_clip) <= 1e-10 * abs(want_clip):
                    fails.append(f"loglik(clip) {got!r} vs {want_clip!r}")
                if state == "fit_like":
                    what = "loglik_strict"
                    got = ctx.loglik_strict()
                    if np.isfinite(want_strict) and not abs(got - want_strict) <= 1e-10 * abs(want_strict):
                        fails.append(f"loglik_strict {got!r} vs {want_strict!r}")
                if state == "fit_like":
                    what = "w_only_steps"
                    ctx.w_only_steps(3)
                    Wg, Hg = ctx.get_factors()
                    with np.errstate(all="ignore"):
                        Ww = _w_steps(Y, W, H, maskf, 3)
                    if np.all(np.isfinite(Ww)) and not (np.allclose(Wg, Ww, rtol=0, atol=1e-9) and np.array_equal(Hg, H)):
                        fails.append(f"w_only_steps max|dW| {np.max(np.abs(Wg - Ww)):.2e}")
                assert n_obs == (Y.size if mask is None else np.count_nonzero(mask)), "n_obs"
        except Exception as e:   # noqa: BLE001
            fails = [f"EXCEPTION in {what}: {e!r}"]
        if fails:
            bad += 1
            print(f"case {case}: {'; '.join(fails)}  m={m} n={n} k={k} real={real} mask={mk} state={state}", flush=True)
        if case % 100 == 99:
            print(f"  ... {case + 1} cases, {bad} failures, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    return bad, time.time() - t0


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n_bad, secs = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 900)
    print(f"{n_cases} cases, {n_bad} failures, {secs:.0f} s")
    sys.exit(1 if n_bad else 0)
