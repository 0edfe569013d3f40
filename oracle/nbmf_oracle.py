"""CPU oracle for the NBMF-MM hot path (TEST INFRASTRUCTURE — not product code).

This file is a NumPy restatement, in this repo's own words, of the one path the HIP library
accelerates.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; nothing under ``nbmf_mm_amd/`` does, and the product path raises
when the HIP library is missing instead of falling back to this.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference package from
``/root/reference/src`` (in the build container only) and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` holds this restatement to those vectors (bitwise for the
one-step vectors and ≤1e-13 for long runs) on every CPU test run.

Extensions with no reference code (Duchi projection, per-row observed-count normaliser):
PARITY UNPINNED — specified by the reference README (README.md:27-35) and Duchi et al. 2008;
checked by properties only.

Reference citations are relative to /root/reference/.
Internal layout everywhere: Y is (m, n); W is (k, m) with columns on the simplex; H is (k, n).
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "heldout_perplexity",
    "mm_step",
    "mm_loss",
    "solve",
    "w_only_transform",
    "score",
    "project_simplex_sort",
    "mm_step_duchi",
]


def _dense(a):
    """Sparse inputs are densified (src/nbmf_mm/_solver.py:28-29,106-107)."""
    return a.toarray() if hasattr(a, "toarray") else a


def mm_step(Y, W, H, mask, alpha, beta, eps=1e-8):
    """One MM iteration: Beta-factor (H) update, then simplex-factor (W) update.

    Follows src/nbmf_mm/_solver.py:5-59 operation for operation (the order of the floating
    point operations is what pins bit-parity):
      * the "zeros" term of the H update uses ``1 - Y*mask`` (not ``(1-Y)*mask``), :43;
      * the W update is strictly masked, :31-32,53;
      * ``/n`` then column renormalisation, :54,57; W is never clipped, H is, :47.
    """
    n = Y.shape[1]
    if mask is None:
        y_obs = Y                       # :23
        yt_obs = Y.T                    # :24
        zt_obs = (1 - Y).T              # :25
    else:
        mask = _dense(mask)
        y_obs = Y * mask                # :30
        yt_obs = Y.T * mask.T           # :31
        zt_obs = (1 - Y).T * mask.T     # :32
    a = np.ones_like(H) * (alpha - 1)   # :35
    b = np.ones_like(H) * (beta - 1)    # :36

    theta = W.T @ H                                             # :39
    num = H * (W @ (y_obs / (theta + eps))) + a                 # :42
    den = (1 - H) * (W @ ((1 - y_obs) / (1 - theta + eps))) + b  # :43
    H_new = num / (num + den + eps)                             # :46
    H_new = np.clip(H_new, eps, 1 - eps)                        # :47

    theta_t = H_new.T @ W                                       # :50
    W_new = W * (H_new @ (yt_obs / (theta_t + eps))
                 + (1 - H_new) @ (zt_obs / (1 - theta_t + eps)))  # :53
    W_new = W_new / n                                           # :54
    W_new = W_new / W_new.sum(axis=0, keepdims=True)            # :57
    return W_new, H_new


def mm_loss(Y, W, H, mask, alpha, beta, eps=1e-8):
    """Negative penalised log-likelihood per observed entry (src/nbmf_mm/_solver.py:148-162).

    The masked form keeps the reference's asymmetry: ``1 - Y*mask`` multiplies the second
    log (:153-154) and the divisor is ``count_nonzero(mask)`` (:155).
    """
    theta = W.T @ H
    if mask is None:
        ll = Y * np.log(theta + eps) + (1 - Y) * np.log(1 - theta + eps)
        n_obs = Y.size
    else:
        y_obs = Y * mask
        ll = y_obs * np.log(theta + eps) + (1 - y_obs) * np.log(1 - theta + eps)
        n_obs = np.count_nonzero(mask)
    pa = (alpha - 1) * np.sum(np.log(H + eps))
    pb = (beta - 1) * np.sum(np.log(1 - H + eps))
    return -(np.sum(ll) + pa + pb) / n_obs


def solve(Y, n_components, max_iter=500, tol=1e-5, alpha=1.2, beta=1.2, W_init=None,
          H_init=None, mask=None, random_state=None, verbose=0, orientation="beta-dir",
          eps=1e-8, step=None):
    """Outer loop: seeding, init draws, orientation transpose, stop rule, final touch-up.

    Restates src/nbmf_mm/_solver.py:61-216.  ``step`` lets tests swap in the extension step
    (``mm_step_duchi``); the default is the reference step.
    Returns ``(W (m,k), H (k,n), losses, 0.0, n_iter)`` exactly as the reference does (:216).
    """
    if step is None:
        step = mm_step
    if random_state is not None:
        np.random.seed(random_state)            # global legacy RNG, :102-103
    if mask is not None:
        mask = _dense(mask)                     # :106-107
    m, n = Y.shape
    k = n_components
    if orientation == "dir-beta":               # transpose trick, :113-123
        Y = Y.T
        m, n = n, m
        if mask is not None:
            mask = mask.T
        if W_init is not None and H_init is not None:
            W_init, H_init = H_init.T, W_init.T
    if W_init is None:
        W_init = np.random.uniform(0.1, 0.9, (m, k))   # W first, :126-127
    if H_init is None:
        H_init = np.random.uniform(0.1, 0.9, (k, n))   # then H, :128-129
    W = W_init.T
    H = H_init
    W = W / W.sum(axis=0, keepdims=True)        # :136

    losses = []
    prev = np.inf
    it = -1
    for it in range(max_iter):                  # :143-175
        W, H = step(Y, W, H, mask, alpha, beta, eps)
        loss = mm_loss(Y, W, H, mask, alpha, beta, eps)
        losses.append(loss)
        if verbose > 0 and it % 10 == 0:
            print(f"Iter {it:4d}: Loss = {loss:.6f}")
        if it > 0:
            if abs(prev - loss) / abs(prev) < tol:
                if verbose > 0:
                    print(f"Converged at iteration {it}")
                break
        prev = loss
    if it < 0:
        raise ValueError("max_iter must be >= 1")   # reference raises UnboundLocalError, :215

    W_out, H_out = W.T, H
    if orientation == "dir-beta":
        W_out, H_out = H_out.T, W_out.T         # :182-184
    # final touch-up, only when the simplex sums drifted by more than 1e-9 (:192-213)
    if orientation == "beta-dir":
        sums = W_out.sum(axis=1, keepdims=True)
        dev = np.max(np.abs(sums - 1.0)) if sums.size else 0.0
        if np.isfinite(dev) and dev > 1e-9:
            ok = (sums > 1e-12).ravel()
            if np.any(ok):
                W_out = np.array(W_out)
                W_out[ok, :] = W_out[ok, :] / sums[ok]
    else:
        sums = H_out.sum(axis=0, keepdims=True)
        dev = np.max(np.abs(sums - 1.0)) if sums.size else 0.0
        if np.isfinite(dev) and dev > 1e-9:
            ok = (sums > 1e-12).ravel()
            if np.any(ok):
                H_out = np.array(H_out)
                H_out[:, ok] = H_out[:, ok] / sums[:, ok]
    return W_out, H_out, losses, 0.0, it + 1


def w_only_transform(X, H, mask=None, W0=None, n_iter=50, track_positive=False):
    """Simplex-factor-only loop with the Beta factor frozen (src/nbmf_mm/_base.py:170-199).

    ``W0`` (m,k) defaults to a draw from the GLOBAL legacy RNG, as the reference does (:175).
    Always the simplex-W form, whatever the fitted orientation; eps is hard-coded 1e-8.
    ``track_positive``: also return which rows kept every entry positive through all iterations -- the start is NOT
    on the simplex (:175), so W @ H can exceed 1, ratios turn negative, and a row that goes through that is on a
    chaotic trajectory in the reference itself (tests/test_oracle_golden.py::test_transform_start_is_chaotic_on_a_few_rows).
    """
    m = X.shape[0]
    k = H.shape[0]
    W = np.random.uniform(0.1, 0.9, (m, k)) if W0 is None else W0
    positive = np.ones(m, dtype=bool)
    for _ in range(n_iter):
        Wt = W.T
        theta_t = H.T @ Wt
        if mask is None:
            yt = X.T
            zt = (1 - X).T
        else:
            yt = X.T * mask.T
            zt = (1 - X).T * mask.T
        Wt = Wt * (H @ (yt / (theta_t + 1e-8)) + (1 - H) @ (zt / (1 - theta_t + 1e-8)))
        Wt = Wt / X.shape[1]
        Wt = Wt / Wt.sum(axis=0, keepdims=True)
        W = Wt.T
        positive &= (W > 0).all(axis=1)
    W = np.clip(W, 1e-8, 1.0)                   # :196
    W = W / W.sum(axis=1, keepdims=True)        # :198
    return (W, positive) if track_positive else W


def score_rows(X, W, H, mask=None):
    """Per-row sums of the log-likelihood terms of :func:`score` (their total / n_obs is the score)."""
    recon = np.clip(W @ H, 0.0, 1.0)
    eps = 1e-8
    xm = X if mask is None else X * mask
    return np.sum(xm * np.log(recon + eps) + (1 - xm) * np.log(1 - recon + eps), axis=1)


def score(X, W, H, mask=None):
    """Mean log-likelihood per observed entry of clip(W@H) (src/nbmf_mm/_base.py:235-247)."""
    recon = np.clip(W @ H, 0.0, 1.0)            # :208-210
    eps = 1e-8
    if mask is None:
        ll = X * np.log(recon + eps) + (1 - X) * np.log(1 - recon + eps)
        n_obs = X.size
    else:
        xm = X * mask
        ll = xm * np.log(recon + eps) + (1 - xm) * np.log(1 - recon + eps)
        n_obs = np.count_nonzero(mask)
    return np.sum(ll) / n_obs


def heldout_perplexity(Y, Y_hat, mask=None, eps=1e-8):
    """Strictly masked perplexity of the reference's experiment script
    (examples/reproduce_magron2022.py:40-47, `compute_perplexity`)."""
    if mask is None:
        mask = np.ones_like(Y)
    ll = Y * np.log(Y_hat + eps) + (1 - Y) * np.log(1 - Y_hat + eps)
    return np.exp(-np.sum(mask * ll) / np.count_nonzero(mask))


# ----------------------------------------------------------------------------------------
# Extensions (no reference code; PARITY UNPINNED).  Spec: /root/reference/README.md:27-35.
# ----------------------------------------------------------------------------------------

def project_simplex_sort(v):
    """Euclidean projection of each COLUMN of ``v`` (k, cols) onto the probability simplex.

    Sort-based algorithm of Duchi et al. 2008 (Fig. 1) / Wang & Carreira-Perpinan 2013:
    with u = sort(v) descending, rho = max{j : u_j - (sum_{r<=j} u_r - 1)/j > 0},
    tau = (sum_{r<=rho} u_r - 1)/rho, projection = max(v - tau, 0).
    """
    k = v.shape[0]
    u = -np.sort(-v, axis=0)
    css = np.cumsum(u, axis=0) - 1.0
    j = np.arange(1, k + 1, dtype=np.float64)[:, None]
    cond = u - css / j > 0
    rho = k - 1 - np.argmax(cond[::-1, :], axis=0)      # last index where cond holds
    tau = css[rho, np.arange(v.shape[1])] / (rho + 1.0)
    return np.maximum(v - tau[None, :], 0.0)


def mm_step_duchi(Y, W, H, mask, alpha, beta, eps=1e-8):
    """Extension step: same Beta-factor update; the simplex factor takes the multiplicative
    step divided by the per-row OBSERVED count (README.md:32-35; equals ``n`` when unmasked)
    and is then projected onto the simplex (README.md:27-30) instead of renormalised."""
    n = Y.shape[1]
    if mask is None:
        y_obs, yt_obs, zt_obs = Y, Y.T, (1 - Y).T
        counts = np.full((1, Y.shape[0]), float(n))
    else:
        mask = _dense(mask)
        y_obs = Y * mask
        yt_obs = Y.T * mask.T
        zt_obs = (1 - Y).T * mask.T
        counts = np.maximum(np.asarray(mask, dtype=np.float64).sum(axis=1), 1.0)[None, :]
    a = alpha - 1
    b = beta - 1
    theta = W.T @ H
    num = H * (W @ (y_obs / (theta + eps))) + a
    den = (1 - H) * (W @ ((1 - y_obs) / (1 - theta + eps))) + b
    H_new = np.clip(num / (num + den + eps), eps, 1 - eps)
    theta_t = H_new.T @ W
    W_new = W * (H_new @ (yt_obs / (theta_t + eps)) + (1 - H_new) @ (zt_obs / (1 - theta_t + eps)))
    W_new = W_new / counts
    W_new = project_simplex_sort(W_new)
    return W_new, H_new
