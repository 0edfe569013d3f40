"""Host side of the solver: same signature and return value as the reference's
``nbmf_mm_solver`` (src/nbmf_mm/_solver.py:61-216), with the hot loop (:143-175) executed by
libnbmf_hip on an MI355X.  Seeding, init draws, the orientation transpose and the final
simplex touch-up stay on the host exactly as the reference orders them.
"""
from __future__ import annotations

import numpy as np

from . import _hip

_PROJECTIONS = {"normalize": _hip.PROJ_NORMALIZE, "duchi": _hip.PROJ_DUCHI}


def _projection_code(projection):
    if projection not in _PROJECTIONS:
        raise ValueError(f"Unknown projection: {projection}. Must be one of {list(_PROJECTIONS)}")
    return _PROJECTIONS[projection]


def _draw_init(m, k, n, W_init, H_init):
    """Init draws in the reference's order: W (m,k) first, then H (k,n); each only if absent
    (src/nbmf_mm/_solver.py:126-129)."""
    if W_init is None:
        W_init = np.random.uniform(0.1, 0.9, (m, k))
    if H_init is None:
        H_init = np.random.uniform(0.1, 0.9, (k, n))
    return W_init, H_init


def _binary_pattern(A):
    """``(indptr, indices)`` of a scipy sparse matrix whose stored values are all 1 (after summing duplicates
    and dropping explicit zeros), else None."""
    A = A.tocsr(copy=True)
    A.sum_duplicates()
    A.eliminate_zeros()
    if A.nnz and not np.all(A.data == 1):
        return None
    if A.shape[1] >= 2 ** 31:
        return None
    return A.indptr.astype(np.int64), A.indices.astype(np.int32)


def _dense(X):
    """Dense data as the library takes it: bool / uint8 arrays as they are (one byte per entry up to the device,
    ``nbmf_upload_v``), float32 as it is (four; the device's conversion is the exact one ``astype(float64)`` does),
    everything else as float64 (the reference converts every input, _base.py:83)."""
    X = np.asarray(X)
    return X if X.dtype in (np.bool_, np.uint8, np.float32) else np.asarray(X, dtype=np.float64)


def upload_any(ctx, X, mask=None, transposed=False):
    """Upload dense or scipy-sparse data: binary sparse patterns (with no mask or a sparse pattern mask) go up as
    CSR (``nbmf_upload_csr``); everything else is densified first, as the reference does (:28-29,106-107)."""
    if hasattr(X, "toarray"):
        csr = _binary_pattern(X)
        csr_mask = None
        if csr is not None and mask is not None:
            csr_mask = _binary_pattern(mask) if hasattr(mask, "toarray") else None
            if csr_mask is None:
                csr = None
        if csr is not None:
            return ctx.upload_csr(csr, csr_mask, transposed=transposed)
        X = X.toarray()
    if mask is not None and hasattr(mask, "toarray"):
        mask = mask.toarray()
    return ctx.upload(_dense(X), mask=mask, transposed=transposed)


def nbmf_mm_solver(Y, n_components, max_iter=500, tol=1e-5, alpha=1.2, beta=1.2, W_init=None,
                   H_init=None, mask=None, random_state=None, verbose=0, orientation="beta-dir",
                   eps=1e-8, projection="normalize", device=0, n_gpus=1, devices=None, _ctx_hook=None):
    """NBMF-MM on the GPU.  Returns ``(W (m,k), H (k,n), losses, 0.0, n_iter)``.

    Mirrors src/nbmf_mm/_solver.py:61-216 argument for argument; ``projection``, ``device`` and ``n_gpus`` / ``devices``
    are extensions.  ``time_elapsed`` is 0.0 as in the reference (:216).  ``n_gpus > 1``: the same fit with the rows of
    Y sharded over that many GPUs from this one process (``_dist.fit_in_process``: one host thread and context per GPU,
    one exchange of the K x N H-step products per iteration); ``devices`` names the GPUs (default 0 .. n_gpus-1).
    """
    if int(n_gpus) != 1 or devices is not None:
        n = int(n_gpus) if devices is None else len(list(devices))
        if devices is not None and int(n_gpus) not in (1, n):
            raise ValueError(f"devices names {n} GPUs, n_gpus is {n_gpus}")
        if n < 1:
            raise ValueError("n_gpus must be >= 1")
        if n > 1:
            from ._dist import fit_in_process
            return fit_in_process(Y, n_components, n, devices=devices, orientation=orientation, max_iter=max_iter, tol=tol,
                                  alpha=alpha, beta=beta, W_init=W_init, H_init=H_init, mask=mask, random_state=random_state,
                                  verbose=verbose, eps=eps, projection=projection)
        if devices is not None:
            device = int(list(devices)[0])
    proj = _projection_code(projection)
    if int(max_iter) < 1:
        raise ValueError("max_iter must be >= 1")      # the reference dies with UnboundLocalError here (:215)
    if random_state is not None:
        np.random.seed(random_state)                   # GLOBAL legacy RNG, as :102-103
    # (sparse input: the reference densifies, :28-29,106-107; here upload_any keeps binary patterns sparse)
    if not hasattr(Y, "toarray"):
        Y = np.asarray(Y)
    m, n = Y.shape
    k = int(n_components)
    transposed = orientation == "dir-beta"
    if transposed:                                     # transpose trick, :113-123
        m, n = n, m
        if W_init is not None and H_init is not None:
            W_init, H_init = np.asarray(H_init).T, np.asarray(W_init).T
    W_init, H_init = _draw_init(m, k, n, W_init, H_init)
    W = np.asarray(W_init, dtype=np.float64).T         # (k, m), :132
    H = np.asarray(H_init, dtype=np.float64)           # (k, n), :133
    W = W / W.sum(axis=0, keepdims=True)               # :136 (a wrong-shaped init raises here, as in the reference)
    if W.shape != (k, m) or H.shape != (k, n):
        raise ValueError(f"operands could not be broadcast together: W_init/H_init give {W.shape}, {H.shape}; "
                         f"expected ({k},{m}), ({k},{n})")

    with _hip.Context(m, n, k, device=device) as ctx:
        ctx.set_hyper(alpha, beta, eps, proj)
        # the user's array goes up untransposed; the pack kernel applies the orientation
        upload_any(ctx, Y, mask, transposed=transposed)
        if _ctx_hook is not None:
            _ctx_hook(ctx)
        ctx.set_factors(W, H)
        if verbose > 0:
            # the reference prints inside its loop (:165-166): the library reports the losses every ten
            # iterations while it runs (one iteration behind the device, see nbmf_set_progress)
            def _report(first, values):
                for it, loss in enumerate(values, start=first):
                    if it % 10 == 0:
                        print(f"Iter {it:4d}: Loss = {loss:.6f}", flush=True)
            ctx.set_progress(_report, every=10)
        losses, n_iter = ctx.run(int(max_iter), float(tol))
        Wk, Hk = ctx.get_factors()

    losses = [float(v) for v in losses]
    if verbose > 0 and (n_iter < int(max_iter) or (n_iter > 1 and losses[-2] != 0 and
                                                   abs(losses[-2] - losses[-1]) / abs(losses[-2]) < tol)):
        print(f"Converged at iteration {n_iter - 1}")   # :172-173

    W_final, H_final = Wk.T, Hk                        # :178-179
    if transposed:
        W_final, H_final = H_final.T, W_final.T        # :182-184
    W_final, H_final = _touch_up(W_final, H_final, orientation)
    return W_final, H_final, losses, 0.0, n_iter


def nbmf_mm_restarts(Y, n_components, n_init, max_iter=500, tol=1e-5, alpha=1.2, beta=1.2, W_init=None, H_init=None,
                     mask=None, random_state=None, orientation="beta-dir", eps=1e-8, projection="normalize", device=0):
    """``n_init`` restarts of :func:`nbmf_mm_solver` (README.md:144; seeds ``random_state + r``) with ONE upload of the
    data and ONE ``nbmf_run_batch`` call: small problems run several restarts at a time in one launch.  Every restart is
    bit for bit the sequential call with that seed (same draws from the global generator, in the same order).  Returns
    ``(W, H, losses, 0.0, n_iter)`` of the restart with the lowest final loss (the first one on ties) and its index."""
    proj = _projection_code(projection)
    if int(max_iter) < 1:
        raise ValueError("max_iter must be >= 1")
    if not hasattr(Y, "toarray"):
        Y = np.asarray(Y)
    m, n = Y.shape
    k = int(n_components)
    transposed = orientation == "dir-beta"
    if transposed:
        m, n = n, m
        if W_init is not None and H_init is not None:
            W_init, H_init = np.asarray(H_init).T, np.asarray(W_init).T
    W0s, H0s = [], []
    for r in range(int(n_init)):
        if random_state is not None:
            np.random.seed(random_state + r)           # restarts: consecutive seeds, each seeding the GLOBAL generator (:102-103)
        Wi, Hi = _draw_init(m, k, n, W_init, H_init)
        W = np.asarray(Wi, dtype=np.float64).T
        W0s.append(W / W.sum(axis=0, keepdims=True))
        H0s.append(np.asarray(Hi, dtype=np.float64))
        if W0s[-1].shape != (k, m) or H0s[-1].shape != (k, n):
            raise ValueError(f"operands could not be broadcast together: W_init/H_init give {W0s[-1].shape}, {H0s[-1].shape}; "
                             f"expected ({k},{m}), ({k},{n})")
    with _hip.Context(m, n, k, device=device) as ctx:
        ctx.set_hyper(alpha, beta, eps, proj)
        upload_any(ctx, Y, mask, transposed=transposed)
        curves, n_iters, Ws, Hs = ctx.run_batch([alpha] * len(W0s), [beta] * len(W0s), np.stack(W0s), np.stack(H0s),
                                                int(max_iter), float(tol))
    best = min(range(len(curves)), key=lambda r: (curves[r][-1], r))
    W_final, H_final = Ws[best].T, Hs[best]
    if transposed:
        W_final, H_final = H_final.T, W_final.T
    W_final, H_final = _touch_up(W_final, H_final, orientation)
    return (W_final, H_final, [float(v) for v in curves[best]], 0.0, int(n_iters[best])), best


def _touch_up(W_final, H_final, orientation):
    """Final renormalisation only where the simplex sums drifted by more than 1e-9
    (src/nbmf_mm/_solver.py:192-213); normally a no-op."""
    if orientation == "beta-dir":
        sums = W_final.sum(axis=1, keepdims=True)
        dev = np.max(np.abs(sums - 1.0)) if sums.size else 0.0
        if np.isfinite(dev) and dev > 1e-9:
            ok = (sums > 1e-12).ravel()
            if np.any(ok):
                W_final = np.array(W_final)
                W_final[ok, :] = W_final[ok, :] / sums[ok]
    else:
        sums = H_final.sum(axis=0, keepdims=True)
        dev = np.max(np.abs(sums - 1.0)) if sums.size else 0.0
        if np.isfinite(dev) and dev > 1e-9:
            ok = (sums > 1e-12).ravel()
            if np.any(ok):
                H_final = np.array(H_final)
                H_final[:, ok] = H_final[:, ok] / sums[:, ok]
    return W_final, H_final


def w_only_transform(X, H, mask=None, W0=None, n_iter=50, device=0):
    """The loop of ``NBMFMM.transform`` (src/nbmf_mm/_base.py:170-199) on the GPU: ``n_iter``
    simplex-factor updates with ``H`` frozen, then clip to [1e-8, 1] and row-renormalise."""
    if not hasattr(X, "toarray"):
        X = _dense(X)
    m, n = X.shape
    k = H.shape[0]
    if W0 is None:
        W0 = np.random.uniform(0.1, 0.9, (m, k))      # global RNG, :175
    with _hip.Context(m, n, k, device=device) as ctx:
        ctx.set_hyper(1.2, 1.2, 1e-8, _hip.PROJ_NORMALIZE)   # eps is hard-coded 1e-8 at :190
        upload_any(ctx, X, mask)
        ctx.set_factors(np.ascontiguousarray(W0.T), H)
        ctx.w_only_steps(int(n_iter))
        Wk, _ = ctx.get_factors()
    W = Wk.T
    W = np.clip(W, 1e-8, 1.0)                          # :196
    W = W / W.sum(axis=1, keepdims=True)               # :198
    return W


def device_score(X, H, mask=None, n_iter=50, device=0):
    """``NBMFMM.score`` on the GPU (src/nbmf_mm/_base.py:212-247): the inner transform runs WITHOUT
    the mask (:235), then the mean log-likelihood per observed entry of W @ H under ``mask``.
    The reference clips W @ H to [0, 1] first (:210); so does the device sweep (clip_theta)."""
    if not hasattr(X, "toarray"):
        X = _dense(X)
    m, n = X.shape
    k = H.shape[0]
    W0 = np.random.uniform(0.1, 0.9, (m, k))          # global RNG, :175
    with _hip.Context(m, n, k, device=device) as ctx:
        ctx.set_hyper(1.2, 1.2, 1e-8, _hip.PROJ_NORMALIZE)
        upload_any(ctx, X, None)
        ctx.set_factors(np.ascontiguousarray(W0.T), H)
        ctx.w_only_steps(int(n_iter))
        Wk, _ = ctx.get_factors()
        W = np.clip(Wk.T, 1e-8, 1.0)                  # :196
        W = W / W.sum(axis=1, keepdims=True)          # :198
        if mask is not None:
            upload_any(ctx, X, mask)
        ctx.set_factors(np.ascontiguousarray(W.T), H)
        ll = ctx.loglik(clip_theta=True)
        n_obs = ctx.n_obs()
    return float(ll / n_obs)
