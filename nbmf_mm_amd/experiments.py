"""GPU-resident equivalent of the reference's experiment driver (examples/reproduce_magron2022.py):
masked training followed by held-out perplexity over a grid of (alpha, beta) prior settings and/or
component counts K.  The data matrix and each mask are packed onto the device ONCE per K; every grid
point reuses them (`train_nbmf_mm`, :49-73; `compute_perplexity`, :40-47; the figure loops, :75-340).
With a process group (nbmf_mm_amd._rendezvous) the grid points are dealt round-robin over the ranks (one GPU each).
"""
from __future__ import annotations

import time

import numpy as np

from . import _hip


def heldout_perplexity(ctx_eval, W_kxm, H_kxn):
    """exp(-sum(mask * loglik) / count_nonzero(mask)) of W^T H on the entries `ctx_eval` holds as
    observed (examples/reproduce_magron2022.py:40-47), evaluated by the Theta-only sweep."""
    ctx_eval.set_factors(W_kxm, H_kxn)
    return float(np.exp(-ctx_eval.loglik_strict() / ctx_eval.n_obs()))


def _run_points(points, Y, train_mask, eval_masks, max_iter, tol, random_state, device):
    """Grid points (K, alpha, beta) on one set of device contexts per K (data packed once per K)."""
    m, n = Y.shape
    rows = []
    for k in sorted({p[0] for p in points}):
        with _hip.Context(m, n, k, device=device) as train:
            train.upload(Y, mask=train_mask)
            evals = {}
            try:
                for name, mk in eval_masks.items():
                    evals[name] = _hip.Context(m, n, k, device=device)
                    evals[name].set_hyper(1.0, 1.0, 1e-8)
                    evals[name].upload(Y, mask=mk)
                mine = [p for p in points if p[0] == k]
                W0, H0 = _init(m, n, k, random_state)                   # same init for every grid point
                train.set_hyper(1.0, 1.0, 1e-8, _hip.PROJ_NORMALIZE)
                # all grid points of this K in one call: the library runs as many of these small fits at a time as the
                # chip holds, one persistent launch per group (nbmf_run_batch); `time` is each fit's share of the call
                t0 = time.perf_counter()
                curves, n_iters, Ws, Hs = train.run_batch([a for _, a, _ in mine], [b for _, _, b in mine], W0, H0,
                                                          int(max_iter), float(tol))
                dt = (time.perf_counter() - t0) / max(1, len(mine))
                for i, (_, a, b) in enumerate(mine):
                    row = {"K": k, "alpha": a, "beta": b, "n_iter": int(n_iters[i]), "loss": float(curves[i][-1]), "time": dt}
                    for name, ev in evals.items():
                        row[name + "_perplexity"] = heldout_perplexity(ev, Ws[i], Hs[i])
                    rows.append(row)
            finally:
                for ev in evals.values():
                    ev.close()
    return rows


def _init(m, n, k, random_state):
    """Reference init rule with a PRIVATE legacy generator seeded like np.random.seed(random_state): the
    same numbers as the global-RNG draw of _solver.py:102-129, but safe to call from several threads."""
    rs = np.random.RandomState(random_state)
    W0 = rs.uniform(0.1, 0.9, (m, k))
    H0 = rs.uniform(0.1, 0.9, (k, n))
    W = W0.T / W0.T.sum(axis=0, keepdims=True)
    return np.ascontiguousarray(W), H0


def perplexity_grid(Y, train_mask, eval_masks, n_components, alphas, betas, max_iter=500, tol=1e-5,
                    random_state=12345, device=0, group=None, concurrency=1):
    """Fit beta-dir NBMF-MM on `train_mask` for every (K, alpha, beta) and report held-out perplexities.

    eval_masks: dict name -> mask (e.g. {"val": val_mask, "test": test_mask}).
    n_components: int or iterable of ints.  Returns a list of dict rows sorted by (K, alpha, beta) with
    keys K, alpha, beta, n_iter, loss, time, and one perplexity per eval mask; every rank gets all rows.
    concurrency > 1 runs that many grid points at once on this GPU (one host thread, HIP stream and set
    of contexts each): the reference's datasets are tiny (50x85 ... 1226x285), a single fit cannot fill
    an MI355X, several independent ones can.  Results do not depend on it.
    """
    Y = np.asarray(Y, dtype=np.float64)
    Ks = [int(n_components)] if np.isscalar(n_components) else [int(k) for k in n_components]
    points = [(k, float(a), float(b)) for k in Ks for a in alphas for b in betas]
    world, rank = (group.world, group.rank) if group is not None else (1, 0)
    mine = points[rank::world]
    args = (Y, train_mask, eval_masks, max_iter, tol, random_state, device)
    conc = max(1, min(int(concurrency), len(mine)))
    if conc == 1:
        rows = _run_points(mine, *args)
    else:
        from concurrent.futures import ThreadPoolExecutor
        parts = [sorted(mine)[i::conc] for i in range(conc)]
        with ThreadPoolExecutor(max_workers=conc) as pool:     # ctypes releases the GIL inside the library
            rows = [r for part in pool.map(lambda pts: _run_points(pts, *args), parts) for r in part]
    if group is not None and world > 1:
        rows = [r for part in group.all_gather(rows) for r in part]
    return sorted(rows, key=lambda r: (r["K"], r["alpha"], r["beta"]))
