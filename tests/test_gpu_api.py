"""The reference's behavioural contract (its tests/ directory) re-stated against the drop-in:
shapes, constraints, monotone loss, aliases, reproducibility, sklearn protocol, extensions."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from nbmf_mm_amd import _hip
    assert _hip.device_count() >= 1, "no MI355X visible: the GPU tests must run on the GPU box"
    return _hip


@pytest.fixture(scope="module")
def data():
    from nbmf_mm_amd._utils import generate_synthetic_binary_data
    X, _, _ = generate_synthetic_binary_data(60, 45, 5, random_state=42)
    return X


def test_shapes_constraints_both_orientations(data):
    from nbmf_mm_amd import NBMF
    for orient, simplex_rows in [("beta-dir", True), ("dir-beta", False)]:
        m = NBMF(n_components=5, orientation=orient, max_iter=100, tol=1e-8, random_state=0).fit(data)
        assert m.W_.shape == (60, 5) and m.components_.shape == (5, 45)
        if simplex_rows:       # tests/test_algorithm_correctness.py:25-39,5-23
            np.testing.assert_allclose(m.W_.sum(axis=1), 1.0, rtol=1e-5)
            cont = m.components_
        else:                  # :129-143,109-127
            np.testing.assert_allclose(m.components_.sum(axis=0), 1.0, rtol=1e-5)
            cont = m.W_
        assert cont.min() >= 0 and cont.max() <= 1 and len(np.unique(cont)) > 100
        l = m.loss_curve_
        assert all(l[i] <= l[i - 1] + 1e-12 for i in range(1, len(l)))          # :41-60
        assert len(m.objective_history_) == m.n_iter_ and m.loss_ == l[-1] == m.reconstruction_err_
        assert isinstance(m.reconstruction_err_, float)
        recon = m.inverse_transform(m.W_)
        assert recon.shape == data.shape and len(np.unique(recon)) > 100           # :62-81


def test_aliases_are_normalised_and_written_back(data):
    from nbmf_mm_amd import NBMF
    for alias, norm in [("Dir-Beta", "dir-beta"), ("Aspect Bernoulli", "dir-beta"), ("binary ICA", "beta-dir"),
                        ("Binary ICA", "beta-dir"), ("bICA", "beta-dir"), ("Beta-Dir", "beta-dir"), ("Dir Beta", "dir-beta")]:
        m = NBMF(n_components=3, orientation=alias, max_iter=5, random_state=0).fit(data)
        assert m.orientation == norm                                              # tests/test_api.py:139-153
    a = NBMF(n_components=4, orientation="Aspect Bernoulli", max_iter=20, tol=0, random_state=0).fit(data)
    b = NBMF(n_components=4, orientation="binary ICA", max_iter=20, tol=0, random_state=0).fit(data.T)
    np.testing.assert_array_equal(a.W_, b.components_.T)                          # tests/test_symmetry.py (bitwise here)


def test_fit_transform_transform_score(data):
    from nbmf_mm_amd import NBMF
    m = NBMF(n_components=5, max_iter=60, random_state=0)
    Wt = m.fit_transform(data)
    np.testing.assert_array_equal(Wt, m.W_)                                       # tests/test_public_api.py:34-41
    W2 = m.transform(data[:7])
    assert W2.shape == (7, 5)
    np.testing.assert_allclose(W2.sum(axis=1), 1.0, atol=1e-12)
    mask = np.random.default_rng(0).random(data.shape) < 0.8
    mm = NBMF(n_components=5, max_iter=60, random_state=0).fit(data, mask=mask)
    s = mm.score(data, mask=mask)
    assert isinstance(s, float) and np.isfinite(s) and mm.perplexity(data, mask=mask) >= 1.0   # tests/test_api.py:73-85


def test_reproducibility_and_seed_dependence(data):
    from nbmf_mm_amd import NBMF
    a = NBMF(n_components=5, random_state=42, max_iter=50).fit(data)
    b = NBMF(n_components=5, random_state=42, max_iter=50).fit(data)
    c = NBMF(n_components=5, random_state=43, max_iter=50).fit(data)
    np.testing.assert_array_equal(a.components_, b.components_)                   # tests/test_nbmf_mm.py:127-138
    np.testing.assert_array_equal(a.W_, b.W_)
    assert abs(a.loss_ - b.loss_) < 1e-8 and not np.allclose(a.components_, c.components_)   # tests/test_reproducibility.py


def test_tolerance_controls_iterations(data):
    from nbmf_mm_amd import NBMFMM
    hi = NBMFMM(n_components=5, tol=0.1, max_iter=1000, random_state=42).fit(data)
    lo = NBMFMM(n_components=5, tol=1e-8, max_iter=1000, random_state=42).fit(data)
    assert hi.n_iter_ < 50 < lo.n_iter_                                           # tests/test_nbmf_mm.py:113-125


def test_prior_effect_ordering(data):
    from nbmf_mm_amd import NBMF
    lo = NBMF(n_components=4, alpha=1.0, beta=3.0, max_iter=150, random_state=0).fit(data).components_.mean()
    hi = NBMF(n_components=4, alpha=3.0, beta=1.0, max_iter=150, random_state=0).fit(data).components_.mean()
    assert hi > lo                                                                # tests/test_algorithm_correctness.py:83-107


def test_extensions_n_init_projection_verbose(data, capsys):
    from nbmf_mm_amd import NBMF
    single = [NBMF(n_components=4, max_iter=30, tol=0, random_state=5 + i).fit(data).loss_ for i in range(3)]
    best = NBMF(n_components=4, max_iter=30, tol=0, random_state=5, n_init=3).fit(data)
    assert best.loss_ == min(single)
    d = NBMF(n_components=4, max_iter=30, tol=0, random_state=5, projection_method="duchi").fit(data)
    np.testing.assert_allclose(d.W_.sum(axis=1), 1.0, atol=1e-12)
    NBMF(n_components=3, max_iter=25, tol=0, random_state=0, verbose=1).fit(data)
    out = capsys.readouterr().out
    assert "Iter    0: Loss = " in out and "Iter   20: Loss = " in out           # _solver.py:165-166 text
    with pytest.raises(ValueError, match="n_components"):
        NBMF(n_components=513, max_iter=2).fit(data)                              # beyond NBMF_MAX_K is refused loudly


def test_large_input_range_check_happens_on_the_device():
    """Above 2^24 entries fit() leaves "X must be binary" (_base.py:90-91) to the device pack, which sees every
    entry anyway; the error and its wording are the reference's."""
    from nbmf_mm_amd import NBMF
    X = np.zeros((4100, 4100))
    X[::7, ::5] = 1.0
    assert X.size > (1 << 24)
    m = NBMF(n_components=4, max_iter=2, tol=0, random_state=0).fit(X)
    assert m.n_iter_ == 2
    X[4099, 4097] = 1.5
    with pytest.raises(ValueError, match="must be binary"):
        NBMF(n_components=4, max_iter=2).fit(X)
    # NaN / inf: sklearn's check_array error, as in the reference (:83) -- for an input this big its pass over the data
    # runs only once the device pack has found something (the pack reads every entry anyway), so a clean fit does not pay
    # for it; the error, its wording and its precedence over "must be binary" and over a bad orientation are the same
    for bad, word in ((np.nan, "NaN"), (np.inf, "infinity")):
        X[4099, 4097] = bad
        with pytest.raises(ValueError, match=word):
            NBMF(n_components=4, max_iter=2).fit(X)
        with pytest.raises(ValueError, match=word):
            NBMF(n_components=4, max_iter=2, orientation="no such orientation").fit(X)
        X[5, 5] = 1.5                                      # out of range AND not finite: sklearn's error comes first
        with pytest.raises(ValueError, match=word):
            NBMF(n_components=4, max_iter=2).fit(X)
        X[5, 5] = 0.0
    X[4099, 4097] = 0.0
    m2 = NBMF(n_components=4, max_iter=2, tol=0, random_state=0).fit(X)
    X[4099, 4097] = np.nan
    for call in (m2.transform, m2.score):
        with pytest.raises(ValueError, match="NaN"):
            call(X)


def test_sparse_input_stays_sparse_and_matches_dense():
    """scipy CSR input with binary values (and an optional sparse pattern mask) is packed on the device from the
    CSR arrays; the fit is bit-identical to the dense one.  Non-binary values or a dense mask take the
    reference's route (densify) and still agree."""
    import scipy.sparse as sp
    from nbmf_mm_amd import NBMF, nbmf_mm_solver, _hip
    r = np.random.default_rng(5)
    m, n, k = 333, 517, 12
    Xd = (r.random((m, n)) < 0.07).astype(np.float64)
    Xd[17, :] = 0.0                                           # an empty row
    Md = r.random((m, n)) < 0.8
    X = sp.csr_matrix(Xd)
    X.indices = X.indices.astype(np.int64)                    # any index width
    M = sp.coo_matrix(Md.astype(np.float64))                  # any sparse format
    calls = []
    orig = _hip.Context.upload_csr
    _hip.Context.upload_csr = lambda self, *a, **kw: (calls.append(1), orig(self, *a, **kw))[1]
    try:
        for kw_s, kw_d in [(dict(), dict()), (dict(mask=M), dict(mask=Md)),
                           (dict(mask=M, orientation="dir-beta"), dict(mask=Md, orientation="dir-beta"))]:
            a = nbmf_mm_solver(X, k, max_iter=15, tol=0, random_state=1, **kw_s)
            b = nbmf_mm_solver(Xd, k, max_iter=15, tol=0, random_state=1, **kw_d)
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
            np.testing.assert_array_equal(a[2], b[2])
        assert len(calls) == 3
        os.environ["NBMF_CSR_PIECE"] = "1000"                 # the staged upload in many pieces
        try:
            a = nbmf_mm_solver(X, k, max_iter=15, tol=0, random_state=1, mask=M, orientation="dir-beta")
        finally:
            del os.environ["NBMF_CSR_PIECE"]
        np.testing.assert_array_equal(a[2], b[2])
        calls.pop()
        e = NBMF(n_components=k, max_iter=15, tol=0, random_state=1).fit(X, mask=M)
        d = NBMF(n_components=k, max_iter=15, tol=0, random_state=1).fit(Xd, mask=Md)
        np.testing.assert_array_equal(e.components_, d.components_)
        assert len(calls) == 4
        # weights in the sparse matrix, or a dense mask: densified as in the reference, same answer as dense input
        Xw = X.copy()
        Xw.data[::3] = 0.5
        a = nbmf_mm_solver(Xw, k, max_iter=5, tol=0, random_state=1)
        b = nbmf_mm_solver(Xw.toarray(), k, max_iter=5, tol=0, random_state=1)
        np.testing.assert_array_equal(a[2], b[2])
        a = nbmf_mm_solver(X, k, max_iter=5, tol=0, random_state=1, mask=Md)
        b = nbmf_mm_solver(Xd, k, max_iter=5, tol=0, random_state=1, mask=Md)
        np.testing.assert_array_equal(a[2], b[2])
        assert len(calls) == 4
        # transform / score / perplexity and the sharded helpers take the same route
        np.random.seed(3)
        ts = e.transform(X[:64], mask=sp.csr_matrix(Md[:64].astype(np.float64)))
        np.random.seed(3)
        td = d.transform(Xd[:64], mask=Md[:64])
        np.testing.assert_array_equal(ts, td)
        np.random.seed(4)
        ss = e.score(X, mask=M)
        np.random.seed(4)
        assert ss == d.score(Xd, mask=Md)
        assert len(calls) == 4 + 1 + 2                        # transform once; score: unmasked transform + masked sweep
        Xbad = X.copy()
        Xbad.data[0] = 1.5
        with pytest.raises(ValueError, match="must be binary"):
            NBMF(n_components=k, max_iter=2).fit(Xbad)
    finally:
        _hip.Context.upload_csr = orig


def test_progress_reports_arrive_during_the_run(monkeypatch):
    """verbose's live prints (_solver.py:165-166) rest on nbmf_set_progress: every loss is reported exactly once,
    in order, in several calls made from inside nbmf_run, with the values the run returns -- also when the stop
    rule ends the run early -- and the run itself is unchanged by reporting.  (Reporting runs on the five-kernel
    path; the comparison run is kept on it too, so that "unchanged" can mean bit for bit.)"""
    from nbmf_mm_amd import _hip
    monkeypatch.setenv("NBMF_PERSISTENT", "0")
    r = np.random.default_rng(3)
    Y = (r.random((150, 220)) < 0.3).astype(np.float64)
    W0 = r.uniform(0.1, 0.9, (7, 150))
    W0 /= W0.sum(axis=0, keepdims=True)
    H0 = r.uniform(0.1, 0.9, (7, 220))
    for max_iter, tol in [(95, 0.0), (400, 1e-4)]:
        with _hip.Context(150, 220, 7) as ctx:
            ctx.set_hyper(1.2, 1.2)
            ctx.upload(Y)
            ctx.set_factors(W0, H0)
            plain, n_plain = ctx.run(max_iter, tol)
            Wp, Hp = ctx.get_factors()
            calls = []
            ctx.set_progress(lambda first, vals: calls.append((first, list(vals))), every=10)
            ctx.set_factors(W0, H0)
            losses, n_iter = ctx.run(max_iter, tol)
            Wq, Hq = ctx.get_factors()
            ctx.set_progress(None)
            ctx.set_factors(W0, H0)
            again, _ = ctx.run(max_iter, tol)
        assert n_iter == n_plain and (tol == 0.0 or 5 < n_iter < max_iter)
        np.testing.assert_array_equal(losses, plain)
        np.testing.assert_array_equal(again, plain)
        np.testing.assert_array_equal(Wq, Wp)
        np.testing.assert_array_equal(Hq, Hp)
        assert len(calls) >= n_iter // 10 and all(len(v) <= 11 for _, v in calls)
        assert [f for f, _ in calls] == list(np.cumsum([0] + [len(v) for _, v in calls[:-1]]))
        np.testing.assert_array_equal(np.concatenate([v for _, v in calls]), losses)


def test_loss_assembled_by_the_sweeps_last_workgroup_is_the_finalize_launch_bit_for_bit(monkeypatch):
    """Single GPU, five-kernel path: the loss and stop test of iteration t ride in the H-pass of iteration t+1 (every
    workgroup hands its partial in with one atomic store; the sweep's last workgroup in launch order waits for all of them
    and sums in finalize_kernel's fixed order: nbmf_pass_kernel.inc, PassFin) instead of a launch of their own.
    Same losses, same stop iteration and same factors as with the separate launch (NBMF_NO_FUSED_FINALIZE=1), for
    binary and real-valued data, with and without the stop rule, over shapes with 1 ... many workgroups per sweep."""
    from nbmf_mm_amd import _hip
    monkeypatch.setenv("NBMF_PERSISTENT", "0")
    r = np.random.default_rng(11)
    for (m, n, k), real in [((150, 220, 7), False), ((150, 220, 7), True), ((1500, 900, 20), False), ((40, 3000, 33), False)]:
        Y = r.random((m, n)) if real else (r.random((m, n)) < 0.3).astype(np.float64)
        mask = r.random((m, n)) < 0.8
        W0 = r.uniform(0.1, 0.9, (k, m))
        W0 /= W0.sum(axis=0, keepdims=True)
        H0 = r.uniform(0.1, 0.9, (k, n))
        for max_iter, tol in [(40, 0.0), (400, 2e-4)]:
            out = {}
            for fused in (True, False):
                if fused:
                    monkeypatch.delenv("NBMF_NO_FUSED_FINALIZE", raising=False)
                else:
                    monkeypatch.setenv("NBMF_NO_FUSED_FINALIZE", "1")
                with _hip.Context(m, n, k) as ctx:
                    ctx.set_hyper(1.2, 1.3)
                    ctx.upload(Y, mask=mask)
                    ctx.set_factors(W0, H0)
                    losses, n_iter = ctx.run(max_iter, tol)
                    W, H = ctx.get_factors()
                    ctx.set_factors(W0, H0)            # (a second run on the same context: the loss slots are empty again)
                    again, n_again = ctx.run(max_iter, tol)
                out[fused] = (losses, n_iter, W, H)
                np.testing.assert_array_equal(again, losses)
                assert n_again == n_iter
            assert out[True][1] == out[False][1] and (tol == 0.0 or 3 < out[True][1] < max_iter)
            for a, b in zip(out[True], out[False]):
                np.testing.assert_array_equal(a, b)


def test_sweep_timing_counts_launches_in_both_forms_and_with_a_stride(monkeypatch):
    """nbmf_timing_enable: the events of a sweep that is one launch ride in its own dispatch packet
    (NBMF_TIMING_BRACKET=1: recorded around it); either way every sweep is counted once, its time is positive and of the
    same size, the run's results do not depend on timing, and enable = n > 1 times the sweeps of every n-th iteration."""
    from nbmf_mm_amd import _hip
    monkeypatch.setenv("NBMF_PERSISTENT", "0")
    r = np.random.default_rng(21)
    m, n, k = 900, 1100, 24
    Y = (r.random((m, n)) < 0.3).astype(np.float64)
    W0 = r.uniform(0.1, 0.9, (k, m))
    W0 /= W0.sum(axis=0, keepdims=True)
    H0 = r.uniform(0.1, 0.9, (k, n))
    out = {}
    for form in ("plain", "attached", "bracket", "stride4"):
        if form == "bracket":
            monkeypatch.setenv("NBMF_TIMING_BRACKET", "1")
        else:
            monkeypatch.delenv("NBMF_TIMING_BRACKET", raising=False)
        with _hip.Context(m, n, k) as ctx:
            ctx.set_hyper(1.2, 1.2)
            ctx.upload(Y)
            ctx.set_factors(W0, H0)
            if form != "plain":
                ctx.timing_enable(4 if form == "stride4" else True)
            losses, n_iter = ctx.run(20, 0.0)
            t = ctx.timing() if form != "plain" else None
            out[form] = (losses, ctx.get_factors(), t)
    for form in ("attached", "bracket", "stride4"):
        np.testing.assert_array_equal(out[form][0], out["plain"][0])
        np.testing.assert_array_equal(out[form][1][0], out["plain"][1][0])
        t = out[form][2]
        want = 5 if form == "stride4" else 20
        assert t["hpass_launches"] == want and t["wpass_launches"] == want
        assert 0.0 < t["hpass_ms"] / want < 5.0 and 0.0 < t["wpass_ms"] / want < 5.0
    ta, tb = out["attached"][2], out["bracket"][2]
    assert 0.5 < ta["hpass_ms"] / tb["hpass_ms"] < 2.0 and 0.5 < ta["wpass_ms"] / tb["wpass_ms"] < 2.0


def test_eps_range_and_log_of_a_negative_number(monkeypatch):
    """eps is a public argument (nbmf_mm_solver(eps=...)): far below the default the binary path's likelihood
    product and its shared reciprocal take their per-entry forms (below 1e-70; 1e-60 and 1e-69 are the last values
    of the shared forms, where the product of four denominators is down to 1e-276) and still follow the oracle, by
    either engine; a factor pair with Theta > 1 + eps gives the reference's NaN loss (log of a negative number,
    _solver.py:150), also when an even number of entries are negative; a denormal eps is refused."""
    from nbmf_mm_amd import _hip, nbmf_mm_solver
    from oracle import nbmf_oracle as orc
    r = np.random.default_rng(4)
    Y = (r.random((90, 140)) < 0.3).astype(np.float64)
    for eps in (1e-60, 1e-69, 1e-80, 1e-200):
        Wr, Hr, lr, _, _ = orc.solve(Y, 5, max_iter=12, tol=0, random_state=1, eps=eps)
        for engine in ("1", "0"):
            monkeypatch.setenv("NBMF_PERSISTENT", engine)
            W, H, l, _, _ = nbmf_mm_solver(Y, 5, max_iter=12, tol=0, random_state=1, eps=eps)
            np.testing.assert_allclose(l, lr, rtol=1e-10, atol=0)
            np.testing.assert_allclose(W, Wr, rtol=0, atol=1e-9)
            np.testing.assert_allclose(H, Hr, rtol=0, atol=1e-9)
    monkeypatch.delenv("NBMF_PERSISTENT", raising=False)
    W0 = r.uniform(0.5, 0.9, (5, 90))                  # columns sum to ~3.5: Theta > 1 in many places
    H0 = r.uniform(0.5, 0.9, (5, 140))
    for Yc in (Y, r.random((90, 140))):                # binary path and general path
        with _hip.Context(90, 140, 5) as ctx:
            ctx.set_hyper(1.2, 1.2)
            ctx.upload(Yc)
            ctx.set_factors(W0, H0)
            with np.errstate(all="ignore"):
                want = orc.mm_loss(Yc, W0, H0, None, 1.2, 1.2)
            assert np.isnan(want) and np.isnan(ctx.loss())
    with _hip.Context(90, 140, 5) as ctx:
        with pytest.raises(ValueError, match="normal"):
            ctx.set_hyper(1.2, 1.2, eps=1e-310)


def test_upload_is_refused_while_a_communicator_is_attached():
    from nbmf_mm_amd import _hip
    r = np.random.default_rng(5)
    Y = (r.random((64, 80)) < 0.3).astype(np.float64)
    with _hip.Context(64, 80, 4) as ctx:
        ctx.set_hyper(1.2, 1.2)
        ctx.upload(Y)
        ctx.comm_init_host(lambda arr: None, 1, 0)
        with pytest.raises(_hip.NBMFHipError, match="before attaching"):
            ctx.upload(Y)
        ctx.comm_detach()
        ctx.upload(Y)


def test_single_launch_path_serves_small_problems_and_falls_back_cleanly(monkeypatch):
    """Small fits run inside one persistent kernel (nbmf_small_kernel.inc); the five-kernel path gives the same
    curve to rounding and takes over, from the factors as they were at entry, when a grid barrier is abandoned
    (provoked here by raising the abort word before the launch)."""
    from nbmf_mm_amd import _hip
    from oracle import nbmf_oracle as orc
    r = np.random.default_rng(8)
    stopped_early = False
    for (m, n, k, real) in [(100, 500, 6, False), (253, 902, 8, False), (1226, 285, 16, False), (77, 130, 20, True), (300, 200, 32, False)]:
        Y = r.random((m, n)) if real else (r.random((m, n)) < 0.3).astype(np.float64)
        mask = r.random((m, n)) < 0.85
        W0 = r.uniform(0.1, 0.9, (k, m))
        W0 /= W0.sum(axis=0, keepdims=True)
        H0 = r.uniform(0.1, 0.9, (k, n))
        out = {}
        for mode in ("single", "five", "abort"):
            monkeypatch.setenv("NBMF_PERSISTENT", "0" if mode == "five" else "1")
            if mode == "abort":
                monkeypatch.setenv("NBMF_SMALL_FORCE_ABORT", "1")
            else:
                monkeypatch.delenv("NBMF_SMALL_FORCE_ABORT", raising=False)
            with _hip.Context(m, n, k) as ctx:
                ctx.set_hyper(1.1, 1.3, 1e-8, _hip.PROJ_DUCHI if k == 8 else _hip.PROJ_NORMALIZE)
                ctx.upload(Y, mask=mask)
                ctx.set_factors(W0, H0)
                l1, n1 = ctx.run(25, 0.0)
                l2, n2 = ctx.run(400, 1e-4)                  # continues from the state the first run left
                out[mode] = (np.concatenate([l1, l2]), n2) + ctx.get_factors() + (ctx.small_stats(),)
        assert out["single"][4] == (2, 0) and out["five"][4] == (0, 0) and out["abort"][4] == (1, 1)
        assert 2 < out["single"][1] <= 400 and out["single"][1] == out["five"][1] == out["abort"][1]
        stopped_early = stopped_early or out["single"][1] < 400
        np.testing.assert_allclose(out["single"][0], out["five"][0], rtol=1e-12, atol=0)
        np.testing.assert_allclose(out["single"][2], out["five"][2], rtol=0, atol=1e-12)
        np.testing.assert_allclose(out["single"][3], out["five"][3], rtol=0, atol=1e-12)
        for j in (0, 2, 3):
            np.testing.assert_array_equal(out["abort"][j], out["five"][j])     # the fall-back IS the five-kernel path
    assert stopped_early
    # and against the oracle, both stop-rule bookkeeping and the factors of the stop iteration
    Yb = (r.random((120, 90)) < 0.3).astype(np.float64)
    monkeypatch.setenv("NBMF_PERSISTENT", "1")
    monkeypatch.delenv("NBMF_SMALL_FORCE_ABORT", raising=False)
    from nbmf_mm_amd import nbmf_mm_solver
    W, H, l, _, nit = nbmf_mm_solver(Yb, 5, max_iter=500, tol=1e-5, random_state=3)
    Wr, Hr, lr, _, nr = orc.solve(Yb, 5, max_iter=500, tol=1e-5, random_state=3)
    assert nit == nr and 2 < nit < 500
    np.testing.assert_allclose(l, lr, rtol=1e-10, atol=0)
    np.testing.assert_allclose(W, Wr, rtol=0, atol=1e-9)
    np.testing.assert_allclose(H, Hr, rtol=0, atol=1e-9)


def test_single_launch_hand_offs_are_reproducible(monkeypatch):
    """The persistent kernel's workgroups hand the factors to each other through memory with write-through stores,
    L1-bypassing loads and flag words instead of cache-maintenance fences: a stale read anywhere would show as a
    run that differs from its repetitions.  Ten repetitions of long runs on shapes with unsplit and split strips,
    bit for bit, and against the five-kernel path (tools/soak_small.py is the long form)."""
    from nbmf_mm_amd import _hip
    for (m, n, k, real, its) in [(100, 500, 6, False, 3000), (1226, 285, 8, False, 800), (253, 902, 12, True, 800),
                                 (1000, 1000, 32, False, 300)]:
        r = np.random.default_rng(1)
        X = r.random((m, n)) if real else (r.random((m, n)) < 0.25).astype(np.float64)
        mask = r.random((m, n)) < 0.9
        W0 = r.uniform(0.1, 0.9, (k, m))
        W0 /= W0.sum(axis=0, keepdims=True)
        H0 = r.uniform(0.1, 0.9, (k, n))
        monkeypatch.setenv("NBMF_PERSISTENT", "0")
        with _hip.Context(m, n, k) as ctx:
            ctx.set_hyper(1.2, 1.2)
            ctx.upload(X, mask=mask)
            ctx.set_factors(W0, H0)
            lref, _ = ctx.run(its, 0.0)
        monkeypatch.setenv("NBMF_PERSISTENT", "1")
        with _hip.Context(m, n, k) as ctx:
            ctx.set_hyper(1.2, 1.2)
            ctx.upload(X, mask=mask)
            first = None
            for _ in range(10):
                ctx.set_factors(W0, H0)
                l, _ = ctx.run(its, 0.0)
                out = (l,) + ctx.get_factors()
                if first is None:
                    first = out
                for a, b in zip(out, first):
                    np.testing.assert_array_equal(a, b)
            assert ctx.small_stats() == (10, 0)
        np.testing.assert_allclose(l, lref, rtol=1e-12, atol=0)


@pytest.mark.parametrize("m,n,k,single", [
    (4096, 300, 8, True),      # 256 row strips: one workgroup on every CU of the chip
    (300, 2048, 12, True),     # 128 column strips: the most the loss-forming wave takes
    (300, 2064, 12, False),    # 129 column strips: the five launches
    (200, 300, 33, False),     # more than 32 components: the five launches
])
def test_single_launch_admission_edges(monkeypatch, m, n, k, single):
    """At the edges of what the single-launch path admits (small_eligible): the problem either runs there or through
    the five launches -- the statistics say which -- and both give the same curve and factors."""
    from nbmf_mm_amd import _hip
    r = np.random.default_rng(m + n + k)
    Y = (r.random((m, n)) < 0.3).astype(np.float64)
    mask = r.random((m, n)) < 0.9
    W0 = r.uniform(0.1, 0.9, (k, m))
    W0 /= W0.sum(axis=0, keepdims=True)
    H0 = r.uniform(0.1, 0.9, (k, n))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NBMF_PERSISTENT", mode)
        with _hip.Context(m, n, k) as ctx:
            ctx.set_hyper(1.2, 1.2)
            ctx.upload(Y, mask=mask)
            ctx.set_factors(W0, H0)
            losses, nit = ctx.run(30, 0.0)
            out[mode] = (losses, nit) + ctx.get_factors() + (ctx.small_stats(),)
    assert out["1"][4] == ((1, 0) if single else (0, 0)) and out["0"][4] == (0, 0)
    assert out["1"][1] == out["0"][1] == 30
    assert all(out["1"][0][i] <= out["1"][0][i - 1] + 1e-12 for i in range(1, 30))
    np.testing.assert_allclose(out["1"][0], out["0"][0], rtol=1e-12, atol=0)
    np.testing.assert_allclose(out["1"][2], out["0"][2], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out["1"][3], out["0"][3], rtol=0, atol=1e-12)


@pytest.mark.parametrize("shape,k,batch_max", [((300, 200), 6, None), ((300, 200), 6, "2"), ((1226, 285), 8, None), ((2600, 2300), 40, None)])
def test_run_batch_is_bitwise_the_sequential_runs(monkeypatch, shape, k, batch_max):
    """nbmf_run_batch: a grid of priors / several starts on one context's data.  Small problems share persistent
    launches (as many problems at a time as the chip holds; NBMF_BATCH_MAX caps it), large ones run one after the
    other inside the call -- either way every problem's loss curve, iteration count and factors are bit for bit
    what set_hyper + set_factors + run + get_factors give on the same context."""
    from nbmf_mm_amd import _hip
    if batch_max:
        monkeypatch.setenv("NBMF_BATCH_MAX", batch_max)
    m, n = shape
    g = np.random.default_rng(31)
    Y = (g.random((m, n)) < 0.15).astype(np.float64)
    mask = g.random((m, n)) < 0.8
    P = 7 if m < 2000 else 3
    alphas = np.array([0.5, 1.0, 1.2, 1.5, 2.0, 2.5, 3.0])[:P]
    betas = np.array([3.0, 1.0, 1.2, 0.5, 2.0, 1.1, 0.7])[:P]
    W0 = g.uniform(0.1, 0.9, (P, k, m))
    W0 /= W0.sum(axis=1, keepdims=True)
    H0 = g.uniform(0.1, 0.9, (P, k, n))
    max_iter, tol = (300, 2e-3) if m < 2000 else (6, 0.0)
    with _hip.Context(m, n, k) as ctx:
        ctx.set_hyper(1.2, 1.2, 1e-8)
        ctx.upload(Y, mask=mask)
        curves, n_iters, Ws, Hs = ctx.run_batch(alphas, betas, W0, H0, max_iter, tol)
        launches, served = ctx.batch_stats()
        seq = []
        for p in range(P):
            ctx.set_hyper(alphas[p], betas[p], 1e-8)
            ctx.set_factors(W0[p], H0[p])
            l, ni = ctx.run(max_iter, tol)
            seq.append((l, ni) + ctx.get_factors())
        # the same start for every problem (one (k, m) / (k, n) pair): what the experiment driver does
        c2, n2, W2, H2 = ctx.run_batch(alphas[:2], betas[:2], W0[0], H0[0], max_iter, tol)
    for p in range(P):
        l, ni, W, H = seq[p]
        assert n_iters[p] == ni
        np.testing.assert_array_equal(curves[p], l)
        np.testing.assert_array_equal(Ws[p], W)
        np.testing.assert_array_equal(Hs[p], H)
    np.testing.assert_array_equal(c2[0], curves[0])
    np.testing.assert_array_equal(W2[0], Ws[0])
    if m < 2000:
        assert len(set(int(v) for v in n_iters)) > 1          # the problems stopped at different iterations, each on its own
        assert served >= P and launches < served                # several problems per persistent launch
        if batch_max:
            assert launches >= (P + 1) // 2
    else:
        assert (launches, served) == (0, 0)                     # too large for the persistent kernel: one after the other


def test_run_batch_falls_back_when_a_launch_gives_up(monkeypatch):
    from nbmf_mm_amd import _hip
    g = np.random.default_rng(32)
    m, n, k, P = 200, 150, 5, 4
    Y = (g.random((m, n)) < 0.2).astype(np.float64)
    W0 = g.uniform(0.1, 0.9, (P, k, m))
    W0 /= W0.sum(axis=1, keepdims=True)
    H0 = g.uniform(0.1, 0.9, (P, k, n))
    al, be = np.linspace(1.0, 2.0, P), np.linspace(2.0, 1.0, P)
    with _hip.Context(m, n, k) as ctx:
        ctx.set_hyper(1.2, 1.2, 1e-8)
        ctx.upload(Y)
        good = ctx.run_batch(al, be, W0, H0, 25, 0.0)
    monkeypatch.setenv("NBMF_SMALL_FORCE_ABORT", "1")
    with _hip.Context(m, n, k) as ctx:
        ctx.set_hyper(1.2, 1.2, 1e-8)
        ctx.upload(Y)
        redo = ctx.run_batch(al, be, W0, H0, 25, 0.0)           # the launch is abandoned; every problem is redone by the five kernels
        assert ctx.small_stats()[1] >= 1
    for p in range(P):
        np.testing.assert_allclose(redo[0][p], good[0][p], rtol=1e-12, atol=0)   # another engine: another order of additions
        np.testing.assert_allclose(redo[2][p], good[2][p], rtol=0, atol=1e-12)
        np.testing.assert_allclose(redo[3][p], good[3][p], rtol=0, atol=1e-12)


def test_single_launch_guard_and_fenced_handoff(monkeypatch):
    """Round 6, the single-launch engine's hand-off.  (a) NBMF_SMALL_FENCED=1 puts an agent-scope release in
    front of every flag store and an acquire behind every poll -- the memory model's own form on top of the sc1 form --:
    the same arithmetic, so the same bits, still served by the persistent kernel.  (b) The end-of-run guard: the last loss
    the persistent kernel reported against the loss of the final factors recomputed by the launch-per-kernel engine (1e-12
    relative, the two engines' tested agreement); made to trip (NBMF_SMALL_GUARD_FAULT=1) the run is redone by the launches
    from the state at entry -- the five-kernel path's own bits -- and counted as given up; left alone it never trips over
    shapes, storage paths, stop rules and continued runs."""
    from nbmf_mm_amd import _hip
    r = np.random.default_rng(18)
    for (m, n, k, real) in [(100, 500, 6, False), (1226, 285, 8, False), (253, 902, 16, False), (77, 130, 20, True), (1500, 900, 12, False)]:
        Y = r.random((m, n)) if real else (r.random((m, n)) < 0.3).astype(np.float64)
        mask = r.random((m, n)) < 0.85
        W0 = r.uniform(0.1, 0.9, (k, m))
        W0 /= W0.sum(axis=0, keepdims=True)
        H0 = r.uniform(0.1, 0.9, (k, n))
        out = {}
        for mode in ("plain", "fenced", "five", "guard_trips"):
            monkeypatch.setenv("NBMF_PERSISTENT", "0" if mode == "five" else "1")
            for var, on in (("NBMF_SMALL_FENCED", mode == "fenced"), ("NBMF_SMALL_GUARD_FAULT", mode == "guard_trips")):
                if on:
                    monkeypatch.setenv(var, "1")
                else:
                    monkeypatch.delenv(var, raising=False)
            with _hip.Context(m, n, k) as ctx:
                ctx.set_hyper(1.1, 1.3, 1e-8)
                ctx.upload(Y, mask=mask)
                ctx.set_factors(W0, H0)
                l1, _ = ctx.run(30, 0.0)
                l2, n2 = ctx.run(300, 1e-5)
                out[mode] = (np.concatenate([l1, l2]), n2) + ctx.get_factors() + (ctx.small_stats(),)
        assert out["plain"][4] == (2, 0) and out["fenced"][4] == (2, 0) and out["five"][4] == (0, 0)
        assert out["guard_trips"][4] == (2, 2)                  # both runs served, both refused by the guard and redone
        for j in (0, 2, 3):
            np.testing.assert_array_equal(out["fenced"][j], out["plain"][j])
            np.testing.assert_array_equal(out["guard_trips"][j], out["five"][j])
        assert out["plain"][1] == out["fenced"][1] == out["five"][1] == out["guard_trips"][1]
        np.testing.assert_allclose(out["plain"][0], out["five"][0], rtol=1e-12, atol=0)
    # batched fits: one problem of every persistent launch is checked the same way, on its own images (the context's factors
    # stay what they were); a trip sends ALL problems of the call to the launches
    g = np.random.default_rng(19)
    m, n, k, P = 180, 140, 5, 6
    Y = (g.random((m, n)) < 0.25).astype(np.float64)
    W0 = g.uniform(0.1, 0.9, (P, k, m))
    W0 /= W0.sum(axis=1, keepdims=True)
    H0 = g.uniform(0.1, 0.9, (P, k, n))
    al, be = np.linspace(1.0, 2.0, P), np.linspace(2.0, 1.0, P)
    res = {}
    for mode in ("plain", "guard_trips", "five"):
        monkeypatch.setenv("NBMF_PERSISTENT", "0" if mode == "five" else "1")
        monkeypatch.setenv("NBMF_BATCH_MAX", "2")                     # three persistent launches of two problems each
        if mode == "guard_trips":
            monkeypatch.setenv("NBMF_SMALL_GUARD_FAULT", "1")
        else:
            monkeypatch.delenv("NBMF_SMALL_GUARD_FAULT", raising=False)
        with _hip.Context(m, n, k) as ctx:
            ctx.set_hyper(1.2, 1.2, 1e-8)
            ctx.upload(Y)
            ctx.set_factors(W0[0], H0[0])
            before = [a.copy() for a in ctx.get_factors()]
            res[mode] = ctx.run_batch(al, be, W0, H0, 40, 0.0) + (ctx.small_stats(), ctx.batch_stats())
            if mode == "plain":
                after = ctx.get_factors()
                np.testing.assert_array_equal(after[0], before[0])    # the guard borrowed the context's pointers, not its factors
                np.testing.assert_array_equal(after[1], before[1])
    assert res["plain"][5] == (3, 6) and res["plain"][4][1] == 0      # three launches served six problems, nothing given up
    assert res["guard_trips"][4][1] >= 1
    for p_ in range(P):
        np.testing.assert_array_equal(res["guard_trips"][0][p_], res["five"][0][p_])
        np.testing.assert_array_equal(res["guard_trips"][2][p_], res["five"][2][p_])
        np.testing.assert_allclose(res["plain"][0][p_], res["five"][0][p_], rtol=1e-12, atol=0)


def test_n_init_restarts_share_one_batched_call():
    """NBMF(n_init=r): the restarts go up as one nbmf_run_batch call; the winner is bit for bit the best of the
    sequential fits with seeds random_state, random_state + 1, ... (README.md:144)."""
    from nbmf_mm_amd import NBMF, nbmf_mm_solver
    g = np.random.default_rng(33)
    X = (g.random((90, 70)) < 0.3).astype(np.float64)
    mask = g.random(X.shape) < 0.85
    for orient in ("beta-dir", "dir-beta"):
        est = NBMF(n_components=5, n_init=4, random_state=9, max_iter=80, tol=1e-6, orientation=orient).fit(X, mask=mask)
        singles = [nbmf_mm_solver(X, 5, max_iter=80, tol=1e-6, mask=mask, random_state=9 + r, orientation=orient) for r in range(4)]
        best = min(range(4), key=lambda r: (singles[r][2][-1], r))
        np.testing.assert_array_equal(est.W_, singles[best][0])
        np.testing.assert_array_equal(est.components_, singles[best][1])
        np.testing.assert_array_equal(est.loss_curve_, singles[best][2])
        assert est.n_iter_ == singles[best][4]


def test_storage_paths_agree_and_can_be_chosen_through_the_api():
    """nbmf_set_storage: the same binary data and bool mask on the byte-code path (automatic), forced onto the 8-byte
    path (the reference's arithmetic for real-valued V: two quotients and two logarithms per entry) and onto the 16-byte
    path (float64 weight tiles) -- three kernels families, one fit.  Also without a mask (all-ones weights)."""
    from nbmf_mm_amd import _hip
    from oracle import nbmf_oracle as orc
    g = np.random.default_rng(41)
    m, n, k = 333, 270, 20
    Y = (g.random((m, n)) < 0.3).astype(np.float64)
    mask = g.random((m, n)) < 0.85
    W0 = g.uniform(0.1, 0.9, (k, m))
    W0 /= W0.sum(axis=0, keepdims=True)
    H0 = g.uniform(0.1, 0.9, (k, n))
    for mk in (mask, None):
        got = {}
        for storage in ("auto", "f64", "f64w"):
            with _hip.Context(m, n, k) as ctx:
                ctx.set_hyper(1.3, 1.1, 1e-8)
                ctx.set_storage(storage)
                binary = ctx.upload(Y, mask=mk)
                assert binary == (storage == "auto")
                ctx.set_factors(W0, H0)
                losses, _ = ctx.run(12, 0.0)
                got[storage] = (losses,) + ctx.get_factors()
        Wr, Hr, lr = W0, H0, []
        mf = None if mk is None else mk.astype(np.float64)
        for _ in range(12):
            Wr, Hr = orc.mm_step(Y, Wr, Hr, mf, 1.3, 1.1, 1e-8)
            lr.append(orc.mm_loss(Y, Wr, Hr, mf, 1.3, 1.1, 1e-8))
        for storage, (l, W, H) in got.items():
            np.testing.assert_allclose(l, lr, rtol=1e-10, atol=0, err_msg=storage)
            np.testing.assert_allclose(W, Wr, rtol=0, atol=1e-9, err_msg=storage)
            np.testing.assert_allclose(H, Hr, rtol=0, atol=1e-9, err_msg=storage)
    with pytest.raises(ValueError):
        _hip.Context(8, 8, 2).set_storage("f32")


def test_generate_slice_is_the_slice_of_the_global_matrix():
    """nbmf_generate_slice: a context holding rows [row0, row0+m) x columns [col0, col0+n) of a matrix n_global wide
    generates exactly those entries of what nbmf_generate gives a context holding the whole matrix (checked through
    the NumPy twin of the generator and one MM iteration against the oracle) -- the basis of the sharded runs at sizes
    no host array can hold."""
    from nbmf_mm_amd import _hip
    from oracle import nbmf_oracle as orc
    M, N, k, seed = 1500, 900, 12, 77
    g = np.random.default_rng(42)
    for (r0, m, c0, n) in [(0, 1500, 0, 900), (640, 500, 0, 900), (1000, 500, 37, 300), (1499, 1, 899, 1)]:
        Yg, Mg = _hip.synthetic_reference(M, N, seed, 0.2, 0.8, rows=np.arange(r0, r0 + m), cols=np.arange(c0, c0 + n))
        W0 = g.uniform(0.1, 0.9, (k, m))
        W0 /= W0.sum(axis=0, keepdims=True)
        H0 = g.uniform(0.1, 0.9, (k, n))
        with _hip.Context(m, n, k) as ctx:
            ctx.set_hyper(1.2, 1.2, 1e-8)
            ctx.generate(seed, density=0.2, observed=0.8, row0=r0, col0=c0, n_global=N)
            assert ctx.n_obs() == float(Mg.sum())
            ctx.set_factors(W0, H0)
            l, _ = ctx.run(1, 0.0)
            W1, H1 = ctx.get_factors()
        Wr, Hr = orc.mm_step(Yg, W0, H0, Mg.astype(np.float64), 1.2, 1.2, 1e-8)
        np.testing.assert_allclose(W1, Wr, rtol=0, atol=1e-12)
        np.testing.assert_allclose(H1, Hr, rtol=0, atol=1e-12)
        np.testing.assert_allclose(l[0], orc.mm_loss(Yg, Wr, Hr, Mg.astype(np.float64), 1.2, 1.2, 1e-8), rtol=1e-12)
    with pytest.raises(ValueError):
        with _hip.Context(10, 10, 2) as ctx:
            ctx.generate(1, row0=0, col0=5, n_global=12)          # the slice does not fit the global width


@pytest.mark.parametrize("orientation", ["beta-dir", "dir-beta"])
@pytest.mark.parametrize("dtype", [np.bool_, np.uint8])
def test_bool_and_uint8_data_go_up_as_bytes_and_fit_the_same_bits(hip, dtype, orientation):
    """Dense bool / uint8 V is handed to the library as it is (nbmf_upload_v, one byte per entry) instead of through the
    float64 copy of _base.py:83: the fit, transform and score are those of the float64 array, bit for bit."""
    from nbmf_mm_amd import NBMF
    r = np.random.default_rng(21)
    Xf = (r.random((150, 210)) < 0.3).astype(np.float64)
    Xb = Xf.astype(dtype)
    mask = r.random((150, 210)) < 0.85
    kw = dict(n_components=7, max_iter=25, tol=0, random_state=3, orientation=orientation)
    for mk in (None, mask):
        a = NBMF(**kw).fit(Xf, mask=mk)
        b = NBMF(**kw).fit(Xb, mask=mk)
        np.testing.assert_array_equal(a.loss_curve_, b.loss_curve_)
        np.testing.assert_array_equal(a.W_, b.W_)
        np.testing.assert_array_equal(a.components_, b.components_)
    np.random.seed(4)
    Ta = a.transform(Xf[:40])
    np.random.seed(4)
    Tb = a.transform(Xb[:40])
    np.testing.assert_array_equal(Ta, Tb)
    np.random.seed(4)
    sa = a.score(Xf, mask=mask)
    np.random.seed(4)
    assert a.score(Xb, mask=mask) == sa
    # the library sees bytes: the byte-code storage path, whatever nbmf_set_storage would pick for doubles
    with hip.Context(150, 210, 7) as ctx:
        assert ctx.upload(Xb, mask=mask) is True
        ctx.set_storage("f64")                       # forced 8-byte storage from 1-byte host data works too
        assert ctx.upload(Xb, mask=mask) is False
    if dtype is np.uint8:
        bad = Xb.copy()
        bad[3, 5] = 7
        with pytest.raises(ValueError, match="must be binary"):
            NBMF(**kw).fit(bad)


def test_measured_mfma_peak_is_near_the_datasheet(hip):
    """nbmf_selftest_mfma_peak (bench.py's roofline.peak_measured): a loop of bare f64 MFMAs on this device.  The
    datasheet says 64 cycles per MFMA and SIMD (78.6 TFLOP/s at 2.4 GHz); round 3 measured 64.5-64.7 with VGPR
    accumulators.  A reading far off either way means the kernel no longer measures what it says (e.g. hipcc has moved
    the accumulators to AGPRs: ~70 cycles)."""
    p = hip.mfma_peak(0, 50.0)
    assert 30.0 <= p["launch_ms"] <= 120.0
    assert 63.0 <= p["cycles_per_mfma_at_2p4GHz"] <= 68.0, p
    assert 74.0 <= p["tflops"] <= 80.0, p
    a, b, c = hip.engine_stats()
    assert a >= 0 and b >= 0 and c >= 0


def test_loss_assembly_inside_the_sweep_is_bounded_and_the_run_resumes(hip, monkeypatch):
    """The loss of iteration t-1 is put together inside the H sweep of iteration t: every workgroup hands in its partial,
    the sweep's last workgroup waits for all of them (nbmf_pass_kernel.inc, PassFin).  That wait is bounded -- no hung
    grid.  Round 6: when it runs out (here: one workgroup withholds its partial, NBMF_PASSFIN_FAULT, and the bound is cut
    to 0.2 s; in the field: a GPU shared with a tenant that saturates it, tests/test_gpu_configs.py) nothing is assembled,
    every later kernel of the run returns at once, and nbmf_run RESUMES from the factors that sweep read with the loss in a
    launch of its own: the same losses and factors bit for bit, with and without a stop rule, counted in
    nbmf_variant_stats; the context stays usable."""
    import time
    monkeypatch.setenv("NBMF_PERSISTENT", "0")          # the launch-per-kernel engine is the one that has this hand-off
    r = np.random.default_rng(8)
    Y = (r.random((600, 400)) < 0.3).astype(np.float64)
    W = r.uniform(0.1, 0.9, (12, 600)); W /= W.sum(axis=0, keepdims=True)
    H = r.uniform(0.1, 0.9, (12, 400))
    with hip.Context(600, 400, 12) as ctx:
        ctx.upload(Y)
        # fault 1: one workgroup withholds its partial, the others are done when the wait (0.2 s) runs out; fault 2: nobody
        # withholds anything but the assembling workgroup does not wait at all -- it gives up while other workgroups of its
        # sweep may still be running, as beside a tenant (the form that exposed a non-uniform give-up in round 6)
        for iters, tol, fault in ((6, 0.0, "1"), (300, 1e-4, "1"), (1, 0.0, "1"), (6, 0.0, "2"), (300, 1e-4, "2")):
            monkeypatch.delenv("NBMF_PASSFIN_FAULT", raising=False)
            ctx.set_factors(W, H)
            good, n_good = ctx.run(iters, tol)
            Wg, Hg = ctx.get_factors()
            monkeypatch.setenv("NBMF_PASSFIN_FAULT", fault)
            ctx.set_factors(W, H)
            before = hip.variant_stats()[2]
            t0 = time.perf_counter()
            again, n_again = ctx.run(iters, tol)
            assert time.perf_counter() - t0 < 5.0
            assert hip.variant_stats()[2] - before == (1 if fault == "1" else hip.variant_stats()[2] - before) <= 1   # (fault 2 on a sweep whose workgroups are all done: nothing to resume)
            Wa, Ha = ctx.get_factors()
            assert n_again == n_good and (tol == 0.0 or 2 < n_good < iters)
            np.testing.assert_array_equal(again, good)
            np.testing.assert_array_equal(Wa, Wg)
            np.testing.assert_array_equal(Ha, Hg)
        monkeypatch.delenv("NBMF_PASSFIN_FAULT")
        ctx.set_factors(W, H)
        once_more, _ = ctx.run(6, 0.0)
        ctx.set_factors(W, H)
        np.testing.assert_array_equal(once_more, ctx.run(6, 0.0)[0])
    # fault 2 where it bites: thousands of workgroups, many still on the chip when the last one gives up
    with hip.Context(17000, 60000, 128) as ctx:
        ctx.generate(seed=5, density=0.05, observed=0.9)
        Wb = r.uniform(0.1, 0.9, (128, 17000)); Wb /= Wb.sum(axis=0, keepdims=True)
        Hb = r.uniform(0.1, 0.9, (128, 60000))
        ctx.set_factors(Wb, Hb)
        good, _ = ctx.run(5, 0.0)
        Hg = ctx.get_factors()[1]
        monkeypatch.setenv("NBMF_PASSFIN_FAULT", "2")
        before = hip.variant_stats()[2]
        ctx.set_factors(Wb, Hb)
        again, _ = ctx.run(5, 0.0)
        assert hip.variant_stats()[2] == before + 1
        np.testing.assert_array_equal(again, good)
        np.testing.assert_array_equal(ctx.get_factors()[1], Hg)


def test_float32_data_goes_up_as_four_bytes_per_entry(hip):
    """A float32 V -- real-valued or binary -- is converted to float64 by the reference before anything else
    (check_array(dtype=float64), _base.py:83: exact).  Here a big one goes to the device as it is (nbmf_upload_v,
    NBMF_DATA_F32) and the pack kernel converts: the same fit, bit for bit, as from the converted array; NaN and
    out-of-range values raise what the reference raises."""
    from nbmf_mm_amd import NBMF
    r = np.random.default_rng(33)
    X32 = r.random((4100, 4100)).astype(np.float32)
    assert X32.size > (1 << 24)
    mask = r.random(X32.shape) < 0.9
    kw = dict(n_components=20, max_iter=4, tol=0, random_state=1)
    a = NBMF(**kw).fit(X32.astype(np.float64), mask=mask)
    b = NBMF(**kw).fit(X32, mask=mask)
    np.testing.assert_array_equal(a.loss_curve_, b.loss_curve_)
    np.testing.assert_array_equal(a.W_, b.W_)
    np.testing.assert_array_equal(a.components_, b.components_)
    with hip.Context(300, 200, 8) as ctx:                 # the library entry itself, any size; binary float32 data -> byte codes
        Xb = (r.random((300, 200)) < 0.3).astype(np.float32)
        assert ctx.upload(Xb) is True and ctx.upload(X32[:300, :200]) is False
    X32[7, 9] = 1.25
    with pytest.raises(ValueError, match="must be binary"):
        NBMF(**kw).fit(X32)
    X32[7, 9] = np.nan
    with pytest.raises(ValueError, match="NaN"):
        NBMF(**kw).fit(X32)
