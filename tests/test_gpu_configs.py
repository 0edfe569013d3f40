"""BASELINE.json configs[1], [3] (its per-rank shard) and [4] (its shape, on one GPU) at full size,
plus the benchmarked Duchi projection at configs[2]'s size.  What a CPU cannot recompute in seconds is
checked through slices: the H-update of a column subset depends only on those columns of Y, the W-update
of a row subset only on those rows, so an oracle restatement on the slice is an exact check of the full
run (reference arithmetic: src/nbmf_mm/_solver.py:39-57; Duchi: README.md:27-35, parity unpinned)."""
import os
import sys

import numpy as np
import pytest

from oracle import nbmf_oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALPHA, BETA, EPS = 1.2, 1.2, 1e-8


def _h_slice(Ycols, Mcols, W0, H0cols):
    """Oracle H-update of a column subset (_solver.py:39-47); Mcols None = unmasked."""
    Ym = Ycols if Mcols is None else Ycols * Mcols
    theta = W0.T @ H0cols
    num = H0cols * (W0 @ (Ym / (theta + EPS))) + (ALPHA - 1)
    den = (1 - H0cols) * (W0 @ ((1 - Ym) / (1 - theta + EPS))) + (BETA - 1)
    return np.clip(num / (num + den + EPS), EPS, 1 - EPS)


def _w_bracket(Yrows, Mrows, W0rows, H1):
    """W0 * bracket of _solver.py:53 for a row subset (strictly masked, :31-32); returns (k, rows)."""
    Mr = np.ones_like(Yrows) if Mrows is None else Mrows
    th_t = H1.T @ W0rows
    return W0rows * (H1 @ ((Yrows.T * Mr.T) / (th_t + EPS)) + (1 - H1) @ (((1 - Yrows).T * Mr.T) / (1 - th_t + EPS)))


def _monotone(l):
    return all(l[i] <= l[i - 1] + 1e-12 for i in range(1, len(l)))


# ------------------------------------------------------------------------------------------------------
# configs[1]: dense V 8192 x 8192, K=32, beta-dir / normalize, 500 iterations, unmasked
# ------------------------------------------------------------------------------------------------------
def test_config2_full_size_500_iterations():
    from bench import init_factors, make_shard
    from nbmf_mm_amd import _hip
    M = N = 8192
    K = 32
    X, _ = make_shard(M, N, 0, M, seed=0, masked=False)
    W0, H0 = init_factors(M, N, K, seed=0)
    with _hip.Context(M, N, K) as ctx:
        ctx.set_hyper(ALPHA, BETA, EPS)
        assert ctx.upload(X) is True
        ctx.set_factors(W0, H0)
        l3, _ = ctx.run(3, 0.0)
        W3, H3 = ctx.get_factors()
        # the first three iterations against the oracle on the WHOLE matrix (seconds on the box's host cores)
        Wr, Hr, lr = W0, H0, []
        for _ in range(3):
            Wr, Hr = orc.mm_step(X, Wr, Hr, None, ALPHA, BETA, EPS)
            lr.append(orc.mm_loss(X, Wr, Hr, None, ALPHA, BETA, EPS))
        np.testing.assert_allclose(l3, lr, rtol=1e-10, atol=0)          # loss curve tolerance
        np.testing.assert_allclose(W3, Wr, rtol=0, atol=1e-9)           # factor tolerance
        np.testing.assert_allclose(H3, Hr, rtol=0, atol=1e-9)
        # the full 500: monotone, constraints, and bitwise repeatable from the same state
        ctx.set_factors(W0, H0)
        la, na = ctx.run(500, 0.0)
        Wa, Ha = ctx.get_factors()
        assert na == 500 and _monotone(la)
        np.testing.assert_array_equal(la[:3], l3)                       # a 3-iteration run is a prefix of the 500
        np.testing.assert_allclose(Wa.sum(axis=0), 1.0, atol=1e-12)
        assert Ha.min() >= EPS and Ha.max() <= 1 - EPS
        ctx.set_factors(W0, H0)
        lb, _ = ctx.run(500, 0.0)
        Wb, Hb = ctx.get_factors()
        np.testing.assert_array_equal(la, lb)
        np.testing.assert_array_equal(Wa, Wb)
        np.testing.assert_array_equal(Ha, Hb)
        # iteration 501 from the device's own state 500, slice-exact against the oracle
        ctx.run(1, 0.0)
        W1, H1 = ctx.get_factors()
    cols = np.random.default_rng(1).choice(N, 128, replace=False)
    rows = np.random.default_rng(2).choice(M, 128, replace=False)
    np.testing.assert_allclose(H1[:, cols], _h_slice(X[:, cols], None, Wa, Ha[:, cols]), rtol=0, atol=1e-12)
    Wn = _w_bracket(X[rows], None, Wa[:, rows], H1) / N
    np.testing.assert_allclose(W1[:, rows], Wn / Wn.sum(axis=0, keepdims=True), rtol=0, atol=1e-12)


# ------------------------------------------------------------------------------------------------------
# configs[2] with the projection bench.py runs: Duchi + per-row observed counts at 65536 x 8192, K=64
# ------------------------------------------------------------------------------------------------------
def test_config3_full_size_duchi_slices():
    from bench import init_factors, make_shard
    from nbmf_mm_amd import _hip
    M, N, K = 65536, 8192, 64
    X, Mk = make_shard(M, N, 0, M, seed=0)
    W0, H0 = init_factors(M, N, K, seed=0)
    with _hip.Context(M, N, K) as ctx:
        ctx.set_hyper(ALPHA, BETA, EPS, _hip.PROJ_DUCHI)
        ctx.upload(X, mask=Mk)
        ctx.set_factors(W0, H0)
        l2, _ = ctx.run(2, 0.0)
        W2, H2 = ctx.get_factors()
        ctx.run(1, 0.0)
        W3, H3 = ctx.get_factors()
        more, _ = ctx.run(5, 0.0)
    rows = np.random.default_rng(3).choice(M, 256, replace=False)
    cols = np.random.default_rng(4).choice(N, 128, replace=False)
    Mf = Mk.astype(np.float64)
    # iteration 3 from the device's state after 2 (W2 is already a Duchi-projected factor)
    np.testing.assert_allclose(H3[:, cols], _h_slice(X[:, cols], Mf[:, cols], W2, H2[:, cols]), rtol=0, atol=1e-12)
    counts = np.maximum(Mf[rows].sum(axis=1), 1.0)[None, :]                     # README.md:32-35
    want = orc.project_simplex_sort(_w_bracket(X[rows], Mf[rows], W2[:, rows], H3) / counts)   # README.md:27-30
    np.testing.assert_allclose(W3[:, rows], want, rtol=0, atol=1e-12)
    # projection properties on ALL rows: on the simplex, non-negative, idempotent
    np.testing.assert_allclose(W3.sum(axis=0), 1.0, atol=1e-12)
    assert W3.min() >= 0.0
    samp = np.random.default_rng(5).choice(M, 4096, replace=False)
    np.testing.assert_allclose(orc.project_simplex_sort(W3[:, samp]), W3[:, samp], rtol=0, atol=1e-14)
    assert _monotone(np.concatenate([l2, more]))


# ------------------------------------------------------------------------------------------------------
# configs[3]: the per-rank shard of the 8-GPU row-sharded run (32768 x 8192, K=64) with a communicator
# attached (one rank: the sharded code path minus the wires), both transports, against the plain run
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("transport", ["rccl", "peer"])
def test_config4_shard_with_communicator(transport):
    from bench import init_factors, make_shard
    from nbmf_mm_amd import _hip
    M, N, K = 32768, 8192, 64
    X, Mk = make_shard(M, N, 0, M, seed=0)
    W0, H0 = init_factors(M, N, K, seed=0)
    outs = []
    for attach in (False, True):
        with _hip.Context(M, N, K) as ctx:
            ctx.set_hyper(ALPHA, BETA, EPS)
            ctx.upload(X, mask=Mk)
            ctx.set_factors(W0, H0)
            if attach:
                if transport == "peer":
                    ctx.comm_init_peer(ctx.peer_export(0), 1, 0)
                else:
                    ctx.comm_init(_hip.comm_unique_id(), 1, 0)
            losses, _ = ctx.run(6, 0.0)
            outs.append((losses,) + ctx.get_factors())
    (l0, Wp, Hp), (l1, Ws, Hs) = outs
    # one rank: the exchange is the identity, but the sums are formed in a different order
    np.testing.assert_allclose(l1, l0, rtol=1e-12, atol=0)
    np.testing.assert_allclose(Ws, Wp, rtol=0, atol=1e-12)
    np.testing.assert_allclose(Hs, Hp, rtol=0, atol=1e-12)
    assert _monotone(l1)
    cols = np.random.default_rng(7).choice(N, 96, replace=False)
    # first iteration of the sharded path, slice-exact against the oracle
    with _hip.Context(M, N, K) as ctx:
        ctx.set_hyper(ALPHA, BETA, EPS)
        ctx.upload(X, mask=Mk)
        ctx.set_factors(W0, H0)
        if transport == "peer":
            ctx.comm_init_peer(ctx.peer_export(0), 1, 0)
        else:
            ctx.comm_init(_hip.comm_unique_id(), 1, 0)
        ctx.run(1, 0.0)
        _, H1 = ctx.get_factors()
    Mf = Mk[:, cols].astype(np.float64)
    np.testing.assert_allclose(H1[:, cols], _h_slice(X[:, cols], Mf, W0, H0[:, cols]), rtol=0, atol=1e-12)


# ------------------------------------------------------------------------------------------------------
# configs[4]: V 360000 x 17000, K=128, dir-beta (internal Y = V.T, 17000 x 360000), 90 % observed;
# data generated on the device, slices regenerated by the NumPy twin of the generator
# ------------------------------------------------------------------------------------------------------
def test_config5_shape_on_one_gpu():
    from nbmf_mm_amd import _hip
    m, n, K = 17000, 360000, 128            # internal layout of dir-beta on V 360000 x 17000 (_solver.py:113-123)
    seed, dens, obs = 5, 0.02, 0.9
    r = np.random.default_rng(8)
    W0 = r.uniform(0.1, 0.9, (K, m))
    W0 /= W0.sum(axis=0, keepdims=True)
    H0 = r.uniform(0.1, 0.9, (K, n))
    with _hip.Context(m, n, K) as ctx:
        ctx.set_hyper(ALPHA, BETA, EPS)
        ctx.generate(seed, density=dens, observed=obs)
        n_obs = ctx.n_obs()
        assert abs(n_obs / (m * n) - obs) < 1e-3
        ctx.set_factors(W0, H0)
        loss0 = ctx.loss()
        l1, _ = ctx.run(1, 0.0)
        W1, H1 = ctx.get_factors()
        more, _ = ctx.run(3, 0.0)
        Wa, Ha = ctx.get_factors()
        ctx.set_factors(W1, H1)
        again, _ = ctx.run(3, 0.0)
        Wb, Hb = ctx.get_factors()
    cols = np.sort(r.choice(n, 96, replace=False))
    rows = np.sort(r.choice(m, 24, replace=False))
    Yc, Mc = _hip.synthetic_reference(m, n, seed, dens, obs, cols=cols)
    np.testing.assert_allclose(H1[:, cols], _h_slice(Yc, Mc.astype(np.float64), W0, H0[:, cols]), rtol=0, atol=1e-12)
    Yr, Mr = _hip.synthetic_reference(m, n, seed, dens, obs, rows=rows)
    Wn = _w_bracket(Yr, Mr.astype(np.float64), W0[:, rows], H1) / n
    np.testing.assert_allclose(W1[:, rows], Wn / Wn.sum(axis=0, keepdims=True), rtol=0, atol=1e-12)
    np.testing.assert_allclose(W1.sum(axis=0), 1.0, atol=1e-12)
    assert H1.min() >= EPS and H1.max() <= 1 - EPS
    assert _monotone(np.concatenate([[loss0], l1, more]))
    np.testing.assert_array_equal(more, again)
    np.testing.assert_array_equal(Wa, Wb)
    np.testing.assert_array_equal(Ha, Hb)


def test_config5_sparse_as_dense_with_restarts():
    """configs[4] as the reference would be handed it: a scipy CSR V 360000 x 17000 (2 % ones, never densified:
    nbmf_upload_csr), K=128, dir-beta, n_init restarts (README.md:144; spread over the ranks by
    _dist.fit_restarts -- one rank here)."""
    import scipy.sparse as sp
    from nbmf_mm_amd import _dist, _rendezvous, nbmf_mm_solver
    Mv, Nv, K = 360000, 17000, 128
    blocks, r = [], np.random.default_rng(9)
    for b0 in range(0, Mv, 8192):
        rows = min(8192, Mv - b0)
        nnz = r.binomial(rows * Nv, 0.02)
        flat = np.unique(r.integers(0, rows * Nv, nnz))
        blocks.append(sp.csr_matrix((np.ones(flat.size), (flat // Nv, flat % Nv)), shape=(rows, Nv)))
    V = sp.vstack(blocks, format="csr")
    assert V.shape == (Mv, Nv) and 0.015 < V.nnz / (Mv * Nv) < 0.025
    kw = dict(max_iter=3, tol=0, orientation="dir-beta")
    group = _rendezvous.SingleGroup()
    W, H, losses, n_iter, best = _dist.fit_restarts(V, K, group, n_init=2, random_state=11, **kw)
    singles = [nbmf_mm_solver(V, K, random_state=11 + i, **kw) for i in range(2)]
    finals = [s[2][-1] for s in singles]
    assert best == int(np.argmin(finals)) and n_iter == 3
    np.testing.assert_array_equal(losses, singles[best][2])
    np.testing.assert_array_equal(W, singles[best][0])
    np.testing.assert_array_equal(H, singles[best][1])
    assert W.shape == (Mv, K) and H.shape == (K, Nv)
    np.testing.assert_allclose(H.sum(axis=0), 1.0, atol=1e-12)          # dir-beta: columns of H on the simplex
    assert W.min() >= EPS and W.max() <= 1 - EPS and _monotone(losses)
    # a row sample of the Beta factor after one iteration against the oracle (internal: a column subset of H)
    Wi, Hi, _, _, _ = nbmf_mm_solver(V, K, random_state=11, max_iter=1, tol=0, orientation="dir-beta")
    np.random.seed(11)                                                   # the init the solver drew (_solver.py:126-136)
    W_init = np.random.uniform(0.1, 0.9, (Nv, K))
    H_init = np.random.uniform(0.1, 0.9, (K, Mv))
    Wint = W_init.T / W_init.T.sum(axis=0, keepdims=True)
    samp = np.sort(np.random.default_rng(10).choice(Mv, 64, replace=False))
    Ycols = np.asarray(V[samp].todense()).T                              # internal Y = V.T: columns = V's rows
    np.testing.assert_allclose(Wi[samp].T, _h_slice(Ycols, None, Wint, H_init[:, samp]), rtol=0, atol=1e-12)
