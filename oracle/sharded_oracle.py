"""CPU statement of the ROW-SHARDED iteration (TEST INFRASTRUCTURE).

What the multi-GPU path computes, written with NumPy and an injected ``allreduce`` so that the
world_size-2 gloo tests can check that sharding changes nothing but summation order:
each rank holds Y[r0:r1], mask[r0:r1], W[:, r0:r1]; H is replicated; per iteration ONE all-reduce
of [P1 | P2 | loglik].  Formulas are those of nbmf_oracle.mm_step / mm_loss
(src/nbmf_mm/_solver.py:39-57,148-162 of the reference).
"""
import numpy as np


def sharded_solve(Y_loc, mask_loc, W_loc, H, alpha, beta, n_obs_global, allreduce, max_iter, tol=0.0, eps=1e-8):
    """Returns (W_loc (k, m_loc), H, losses).  ``allreduce(arr)`` sums a float64 array over ranks in place."""
    n = Y_loc.shape[1]
    k = H.shape[0]
    a, b = alpha - 1, beta - 1
    y_obs = Y_loc if mask_loc is None else Y_loc * mask_loc
    yt = Y_loc.T if mask_loc is None else Y_loc.T * mask_loc.T
    zt = (1 - Y_loc).T if mask_loc is None else (1 - Y_loc).T * mask_loc.T
    losses = []
    prev = np.inf

    def products(W_loc, H):
        theta = W_loc.T @ H
        buf = np.empty(2 * k * n + 1)
        buf[:k * n] = (W_loc @ (y_obs / (theta + eps))).ravel()
        buf[k * n:2 * k * n] = (W_loc @ ((1 - y_obs) / (1 - theta + eps))).ravel()
        buf[-1] = np.sum(y_obs * np.log(theta + eps) + (1 - y_obs) * np.log(1 - theta + eps))
        allreduce(buf)
        return buf[:k * n].reshape(k, n), buf[k * n:2 * k * n].reshape(k, n), buf[-1]

    def finish(ll, H):
        pa = a * np.sum(np.log(H + eps))
        pb = b * np.sum(np.log(1 - H + eps))
        return -(ll + pa + pb) / n_obs_global

    for it in range(max_iter):
        P1, P2, ll = products(W_loc, H)          # ll belongs to the factors of iteration it-1
        if it > 0:
            loss = finish(ll, H)
            losses.append(loss)
            if it - 1 > 0 and abs(prev - loss) / abs(prev) < tol:
                return W_loc, H, losses
            prev = loss
        num = H * P1 + a
        den = (1 - H) * P2 + b
        H = np.clip(num / (num + den + eps), eps, 1 - eps)
        theta_t = H.T @ W_loc
        W_new = W_loc * (H @ (yt / (theta_t + eps)) + (1 - H) @ (zt / (1 - theta_t + eps)))
        W_new = W_new / n
        W_loc = W_new / W_new.sum(axis=0, keepdims=True)
    _, _, ll = products(W_loc, H)
    losses.append(finish(ll, H))
    return W_loc, H, losses


def sharded_solve_cols(Y_loc, mask_loc, W, H_loc, alpha, beta, n_obs_global, n_global, allreduce, max_iter, tol=0.0,
                       eps=1e-8):
    """The other split: each rank holds the COLUMNS Y[:, j0:j1] and H[:, j0:j1]; W is replicated.  The H-step
    is local; the W-step bracket (k x m) is all-reduced; [loglik, prior A, prior B] travel before the stop
    test.  Returns (W, H_loc, losses)."""
    a, b = alpha - 1, beta - 1
    y_obs = Y_loc if mask_loc is None else Y_loc * mask_loc
    yt = Y_loc.T if mask_loc is None else Y_loc.T * mask_loc.T
    zt = (1 - Y_loc).T if mask_loc is None else (1 - Y_loc).T * mask_loc.T
    losses = []
    prev = np.inf

    def scalars(W, H_loc):
        theta = W.T @ H_loc
        buf = np.array([np.sum(y_obs * np.log(theta + eps) + (1 - y_obs) * np.log(1 - theta + eps)),
                        np.sum(np.log(H_loc + eps)), np.sum(np.log(1 - H_loc + eps))])
        allreduce(buf)
        return -(buf[0] + a * buf[1] + b * buf[2]) / n_obs_global

    for it in range(max_iter):
        if it > 0:
            loss = scalars(W, H_loc)
            losses.append(loss)
            if it - 1 > 0 and abs(prev - loss) / abs(prev) < tol:
                return W, H_loc, losses
            prev = loss
        theta = W.T @ H_loc
        num = H_loc * (W @ (y_obs / (theta + eps))) + a
        den = (1 - H_loc) * (W @ ((1 - y_obs) / (1 - theta + eps))) + b
        H_loc = np.clip(num / (num + den + eps), eps, 1 - eps)
        theta_t = H_loc.T @ W
        Q = H_loc @ (yt / (theta_t + eps)) + (1 - H_loc) @ (zt / (1 - theta_t + eps))
        Q = np.ascontiguousarray(Q)
        allreduce(Q.reshape(-1))
        W_new = W * Q
        W_new = W_new / n_global
        W = W_new / W_new.sum(axis=0, keepdims=True)
    losses.append(scalars(W, H_loc))
    return W, H_loc, losses
