"""BASELINE-size checks (V 65536 x 8192, K=64, 90 % observed) through properties that do not need
a full CPU run: slice-exact parity (the H-update of a column subset depends only on those columns,
the W-update of a row subset only on those rows), additivity of the log-likelihood over row shards,
monotone loss, simplex/range constraints and bitwise run-to-run determinism."""
import numpy as np
import pytest

from oracle import nbmf_oracle as orc

pytestmark = pytest.mark.gpu

M, N, K = 65536, 8192, 64
ALPHA, BETA, EPS = 1.2, 1.2, 1e-8


@pytest.fixture(scope="module")
def problem():
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_shard, init_factors
    X, Mk = make_shard(M, N, 0, M, seed=0)
    W0, H0 = init_factors(M, N, K, seed=0)
    return X, Mk, W0, H0


def _prior(H):
    return (ALPHA - 1) * np.sum(np.log(H + EPS)) + (BETA - 1) * np.sum(np.log(1 - H + EPS))


def test_full_size_iteration_properties(problem):
    from nbmf_mm_amd import _hip
    X, Mk, W0, H0 = problem
    with _hip.Context(M, N, K) as ctx:
        ctx.set_hyper(ALPHA, BETA, EPS)
        assert ctx.upload(X, mask=Mk) is True                      # 1 byte/entry path
        n_obs = ctx.n_obs()
        assert n_obs == float(np.count_nonzero(Mk))
        ctx.set_factors(W0, H0)
        loss_init = ctx.loss()
        losses1, _ = ctx.run(1, 0.0)
        W1, H1 = ctx.get_factors()
        # ---- slice-exact parity of one whole iteration against the oracle
        cols = np.random.default_rng(1).choice(N, 192, replace=False)
        rows = np.random.default_rng(2).choice(M, 384, replace=False)
        Ym = X[:, cols] * Mk[:, cols]
        theta = W0.T @ H0[:, cols]
        num = H0[:, cols] * (W0 @ (Ym / (theta + EPS))) + (ALPHA - 1)
        den = (1 - H0[:, cols]) * (W0 @ ((1 - Ym) / (1 - theta + EPS))) + (BETA - 1)
        H_ref = np.clip(num / (num + den + EPS), EPS, 1 - EPS)
        np.testing.assert_allclose(H1[:, cols], H_ref, rtol=0, atol=1e-12)
        Yr, Mr = X[rows], Mk[rows].astype(np.float64)
        th_t = H1.T @ W0[:, rows]
        Wn = W0[:, rows] * (H1 @ ((Yr.T * Mr.T) / (th_t + EPS)) + (1 - H1) @ (((1 - Yr).T * Mr.T) / (1 - th_t + EPS)))
        Wn = Wn / N
        Wn = Wn / Wn.sum(axis=0, keepdims=True)
        np.testing.assert_allclose(W1[:, rows], Wn, rtol=0, atol=1e-12)
        # ---- constraints
        np.testing.assert_allclose(W1.sum(axis=0), 1.0, atol=1e-12)
        assert H1.min() >= EPS and H1.max() <= 1 - EPS
        # ---- more iterations: monotone, and bitwise reproducible from the same state
        more, _ = ctx.run(6, 0.0)
        curve = np.concatenate([[loss_init], losses1, more])
        assert all(curve[i] <= curve[i - 1] + 1e-12 for i in range(1, len(curve)))
        Wa, Ha = ctx.get_factors()
        ctx.set_factors(W1, H1)
        again, _ = ctx.run(6, 0.0)
        Wb, Hb = ctx.get_factors()
        np.testing.assert_array_equal(more, again)
        np.testing.assert_array_equal(Wa, Wb)
        np.testing.assert_array_equal(Ha, Hb)
        full_loss = curve[-1]

    # ---- additivity of the log-likelihood over row shards ("checksum of checksums"), and one shard
    #      against the oracle
    ll_total, nobs_total = 0.0, 0.0
    bounds = [(0, 2048), (2048, 20000), (20000, 47001), (47001, M)]
    for (r0, r1) in bounds:
        with _hip.Context(r1 - r0, N, K) as sh:
            sh.set_hyper(ALPHA, BETA, EPS)
            sh.upload(X[r0:r1], mask=Mk[r0:r1])
            sh.set_factors(np.ascontiguousarray(Wa[:, r0:r1]), Ha)
            li, ni = sh.loss(), sh.n_obs()
            ll_total += -li * ni - _prior(Ha)
            nobs_total += ni
            if r0 == 0:
                want = orc.mm_loss(X[r0:r1], np.ascontiguousarray(Wa[:, r0:r1]), Ha, Mk[r0:r1].astype(np.float64),
                                   ALPHA, BETA, EPS)
                assert abs(li - want) <= 1e-12 * abs(want)
    assert nobs_total == n_obs
    assembled = -(ll_total + _prior(Ha)) / nobs_total
    assert abs(assembled - full_loss) <= 1e-12 * abs(full_loss)


@pytest.mark.parametrize("weights", [False, True], ids=["bool-mask-folded-in", "float64-weights"])
def test_full_size_real_valued_data(weights):
    """configs[2]'s shape with GENUINELY real-valued V (the reference accepts any V in [0, 1], _base.py:90; its tests feed
    np.random.rand) -- the general path of the sweeps at full size: 8 bytes per entry with the bool mask folded in, 16 with
    float64 weights (half the rows: 2 x 4.3 GB of tiles per image).  Slice-exact H and W updates (<= 1e-12), the loss of
    the start against the oracle on the WHOLE matrix (1e-10: what the Theta-only sweep sums), monotone, bitwise repeat, and
    a k-iteration run as a prefix of a (k+1)-iteration one."""
    from nbmf_mm_amd import _hip
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import init_factors
    Mr = M // 2 if weights else M
    r = np.random.default_rng(77)
    X = r.random((Mr, N))
    Mk = r.random((Mr, N)) if weights else (r.random((Mr, N)) < 0.9)
    W0, H0 = init_factors(Mr, N, K, seed=1)
    Mf = Mk if weights else Mk.astype(np.float64)
    with _hip.Context(Mr, N, K) as ctx:
        ctx.set_hyper(ALPHA, BETA, EPS)
        assert ctx.upload(X, mask=Mk) is False
        ctx.set_factors(W0, H0)
        want = orc.mm_loss(X, W0, H0, Mf, ALPHA, BETA, EPS)
        loss0 = ctx.loss()
        assert abs(loss0 - want) <= 1e-10 * abs(want)
        l1, _ = ctx.run(1, 0.0)
        W1, H1 = ctx.get_factors()
        cols = np.random.default_rng(1).choice(N, 160, replace=False)
        rows = np.random.default_rng(2).choice(Mr, 320, replace=False)
        Ym = X[:, cols] * Mf[:, cols]
        theta = W0.T @ H0[:, cols]
        num = H0[:, cols] * (W0 @ (Ym / (theta + EPS))) + (ALPHA - 1)
        den = (1 - H0[:, cols]) * (W0 @ ((1 - Ym) / (1 - theta + EPS))) + (BETA - 1)
        np.testing.assert_allclose(H1[:, cols], np.clip(num / (num + den + EPS), EPS, 1 - EPS), rtol=0, atol=1e-12)
        Yr, Mrr = X[rows], Mf[rows]
        th_t = H1.T @ W0[:, rows]
        Wn = W0[:, rows] * (H1 @ ((Yr.T * Mrr.T) / (th_t + EPS)) + (1 - H1) @ (((1 - Yr).T * Mrr.T) / (1 - th_t + EPS)))
        Wn = Wn / N
        np.testing.assert_allclose(W1[:, rows], Wn / Wn.sum(axis=0, keepdims=True), rtol=0, atol=1e-12)
        np.testing.assert_allclose(W1.sum(axis=0), 1.0, atol=1e-12)
        assert H1.min() >= EPS and H1.max() <= 1 - EPS
        more, _ = ctx.run(4, 0.0)
        curve = np.concatenate([[loss0], l1, more])
        assert all(curve[i] <= curve[i - 1] + 1e-12 for i in range(1, len(curve)))
        Wa, Ha = ctx.get_factors()
        ctx.set_factors(W1, H1)
        again, _ = ctx.run(4, 0.0)
        np.testing.assert_array_equal(more, again)
        np.testing.assert_array_equal(Ha, ctx.get_factors()[1])
        ctx.set_factors(W1, H1)
        three, _ = ctx.run(3, 0.0)
        np.testing.assert_array_equal(three, more[:3])


def test_tall_narrow_and_wide_short_shapes():
    """Index ranges at the extremes: 1.2 M x 24 (more pack tile-rows than one launch's grid.y holds)
    and its transpose through dir-beta; both against each other (the transpose identity is bitwise)
    and a row sample against the oracle's W-step."""
    from nbmf_mm_amd import nbmf_mm_solver
    r = np.random.default_rng(5)
    Mt, Nt, Kt = 1_200_003, 24, 5
    V = (r.random((Mt, Nt)) < 0.3).astype(np.float64)
    W, H, l, _, _ = nbmf_mm_solver(V, Kt, max_iter=4, tol=0, random_state=1)
    H2, W2, l2, _, _ = nbmf_mm_solver(V.T, Kt, max_iter=4, tol=0, random_state=1, orientation="dir-beta")
    np.testing.assert_array_equal(l, l2)
    np.testing.assert_array_equal(W, W2.T)
    np.testing.assert_array_equal(H, H2.T)
    np.testing.assert_allclose(W.sum(axis=1), 1.0, atol=1e-12)
    assert all(l[i] <= l[i - 1] + 1e-12 for i in range(1, len(l)))
    # one more iteration from the returned state, checked on a row sample (the W-update is row-local)
    rows = r.choice(Mt, 500, replace=False)
    from nbmf_mm_amd import _hip
    with _hip.Context(Mt, Nt, Kt) as ctx:
        ctx.set_hyper(1.2, 1.2)
        ctx.upload(V)
        ctx.set_factors(np.ascontiguousarray(W.T), H)
        ctx.run(1, 0.0)
        W1, H1 = ctx.get_factors()
    th_t = H1.T @ W.T[:, rows]
    Yr = V[rows]
    Wn = W.T[:, rows] * (H1 @ (Yr.T / (th_t + EPS)) + (1 - H1) @ ((1 - Yr).T / (1 - th_t + EPS)))
    Wn = Wn / Nt
    Wn = Wn / Wn.sum(axis=0, keepdims=True)
    np.testing.assert_allclose(W1[:, rows], Wn, rtol=0, atol=1e-12)


def test_k128_dir_beta_large():
    """configs[4]-like orientation and K at a size one GPU runs in seconds: V 24576 x 12288, K=128,
    dir-beta, masked; transpose identity against beta-dir on V.T (bitwise), constraints, monotone loss."""
    from nbmf_mm_amd import nbmf_mm_solver
    r = np.random.default_rng(6)
    V = (r.random((24576, 12288)) < 0.05).astype(np.float64)
    mask = r.random(V.shape) < 0.9
    W, H, l, _, _ = nbmf_mm_solver(V, 128, max_iter=5, tol=0, random_state=2, orientation="dir-beta", mask=mask)
    Ht, Wt, lt, _, _ = nbmf_mm_solver(np.ascontiguousarray(V.T), 128, max_iter=5, tol=0, random_state=2,
                                      mask=np.ascontiguousarray(mask.T))
    np.testing.assert_array_equal(l, lt)
    np.testing.assert_array_equal(W, Wt.T)
    np.testing.assert_array_equal(H, Ht.T)
    np.testing.assert_allclose(H.sum(axis=0), 1.0, atol=1e-12)          # dir-beta: columns of H on the simplex
    assert W.min() >= 1e-8 and W.max() <= 1 - 1e-8
    assert all(l[i] <= l[i - 1] + 1e-12 for i in range(1, len(l)))
