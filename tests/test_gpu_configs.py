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


def test_config5_shape_as_dense_uint8_equals_the_csr_upload():
    """configs[4] "sparse-as-dense": the SAME 360000 x 17000 matrix handed over as a dense uint8 array (6.1 GB of host
    memory, nbmf_upload_v; as float64 it would be 49 GB) and as scipy CSR (nbmf_upload_csr): both are byte codes on the
    device, so the two fits are the same bits -- K=128, dir-beta, through the estimator, which passes uint8 on without
    the float64 copy of _base.py:83."""
    import scipy.sparse as sp
    from nbmf_mm_amd import NBMF, _hip
    Mv, Nv, K = 360000, 17000, 128
    X8 = np.zeros((Mv, Nv), dtype=np.uint8)
    r = np.random.default_rng(9)
    for b0 in range(0, Mv, 8192):
        rows = min(8192, Mv - b0)
        flat = np.unique(r.integers(0, rows * Nv, r.binomial(rows * Nv, 0.02)))
        X8[b0:b0 + rows].reshape(-1)[flat] = 1
    assert X8.nbytes == Mv * Nv and 0.015 < X8.mean(dtype=np.float64) < 0.025
    V = sp.csr_matrix(X8)
    kw = dict(n_components=K, max_iter=3, tol=0, orientation="dir-beta", random_state=11)
    served0 = _hip.engine_stats()
    dense = NBMF(**kw).fit(X8)
    sparse = NBMF(**kw).fit(V)
    assert _hip.engine_stats()[2] - served0[2] == 2
    np.testing.assert_array_equal(dense.loss_curve_, sparse.loss_curve_)
    np.testing.assert_array_equal(dense.W_, sparse.W_)
    np.testing.assert_array_equal(dense.components_, sparse.components_)
    assert _monotone(dense.loss_curve_) and dense.W_.shape == (Mv, K)
    # ... and the context says which path it took: byte codes, straight from the bytes
    with _hip.Context(Nv, Mv, K) as ctx:
        assert ctx.upload(X8, transposed=True) is True and ctx.n_obs() == float(Mv) * Nv
    # a uint8 value that is not 0 or 1 is the reference's "X must be binary" (_base.py:90-91), found by the device pack
    X8[Mv // 2, Nv // 3] = 2
    with pytest.raises(ValueError, match="must be binary"):
        NBMF(**kw).fit(X8)


# ------------------------------------------------------------------------------------------------------
# configs[3] WHOLE: V 262144 x 8192, K=64 -- on one context, and as 8 ranks (32768 rows each) that share the one
# GPU of the box: 4 processes x 2 ranks (the box admits at most 6 GPU processes at once; ranks of one process are
# host threads with a context and stream each, and the peer transport addresses their arenas directly, the other
# processes' through HIP IPC -- both kinds of peer in every exchange).  Every rank generates ITS rows of the same
# global matrix on the device (nbmf_generate_slice), so the sharded runs can be held against the one-context run.
# ------------------------------------------------------------------------------------------------------
C4 = dict(M=262144, N=8192, K=64, seed=3, dens=0.25, obs=0.9, tol_stop=2e-3, it_stop=12)


def _c4_init():
    from bench import init_factors
    return init_factors(C4["M"], C4["N"], C4["K"], seed=0)


def _c4_rank(rank, world, port, transports, W0, H0, out):
    """One rank of the 8: returns per transport the tol=0 losses, the stop-rule run's (n_iter, losses), the
    replicated H of both and a row sample of its W."""
    from nbmf_mm_amd import _dist, _hip, _rendezvous
    M, N, K = C4["M"], C4["N"], C4["K"]
    group = _rendezvous.Group(rank, world, ("tcp", "127.0.0.1", port), timeout=120, secret=b"tests-%d" % port)
    try:
        r0, r1 = _dist.shard_bounds(M, world, rank)
        with _hip.Context(r1 - r0, N, K) as ctx:
            ctx.set_hyper(ALPHA, BETA, EPS)
            ctx.generate(C4["seed"], density=C4["dens"], observed=C4["obs"], row0=r0, col0=0, n_global=N)
            res = {"n_obs": ctx.n_obs(), "rows": (r0, r1)}
            for tr in transports:
                used = _dist.attach_comm(ctx, group, tr)
                assert used == tr
                ctx.set_factors(np.ascontiguousarray(W0[:, r0:r1]), H0)
                l0, n0 = ctx.run(3, 0.0)
                Wa, Ha = ctx.get_factors()
                entry = {"tol0": (l0, Ha, Wa[:, ::61].copy())}
                if tr != "host":
                    ctx.set_factors(np.ascontiguousarray(W0[:, r0:r1]), H0)
                    ls, ns = ctx.run(C4["it_stop"], C4["tol_stop"])
                    Ws, Hs = ctx.get_factors()
                    entry["stop"] = (ns, ls, Hs, Ws[:, ::61].copy())
                res[tr] = entry
                ctx.comm_detach()
        out[rank] = res
    finally:
        group.close()


def _c4_process(first_rank, ranks_here, world, port, transports, q):
    import threading
    import traceback
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # in-kernel waits between ranks of this process: no shared hardware queue
    os.environ.setdefault("NBMF_PEER_TIMEOUT_MS", "60000")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    W0, H0 = _c4_init()
    out, errors = {}, []

    def body(rank):
        try:
            _c4_rank(rank, world, port, transports, W0, H0, out)
        except BaseException:
            errors.append(traceback.format_exc())
    threads = [threading.Thread(target=body, args=(first_rank + t,)) for t in range(ranks_here)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    q.put((first_rank, out, errors))


def test_config4_whole_on_one_gpu_and_as_eight_ranks_sharing_it():
    import multiprocessing as mp
    import socket
    from nbmf_mm_amd import _hip
    M, N, K = C4["M"], C4["N"], C4["K"]
    W0, H0 = _c4_init()
    # ---- one context holds the whole matrix
    with _hip.Context(M, N, K) as ctx:
        ctx.set_hyper(ALPHA, BETA, EPS)
        ctx.generate(C4["seed"], density=C4["dens"], observed=C4["obs"])
        n_obs = ctx.n_obs()
        assert abs(n_obs / (M * N) - C4["obs"]) < 1e-3
        ctx.set_factors(W0, H0)
        loss0 = ctx.loss()
        l1, _ = ctx.run(1, 0.0)
        W1, H1 = ctx.get_factors()
        ctx.set_factors(W0, H0)
        l3, _ = ctx.run(3, 0.0)
        W3, H3 = ctx.get_factors()
        ctx.set_factors(W0, H0)
        l3b, _ = ctx.run(3, 0.0)
        W3b, H3b = ctx.get_factors()
        ctx.set_factors(W0, H0)
        ls, ns = ctx.run(C4["it_stop"], C4["tol_stop"])
        Ws, Hs = ctx.get_factors()
    # slice-exact first iteration against the NumPy twin of the generator + the oracle's arithmetic
    r = np.random.default_rng(12)
    cols = np.sort(r.choice(N, 64, replace=False))
    rows = np.sort(r.choice(M, 96, replace=False))
    Yc, Mc = _hip.synthetic_reference(M, N, C4["seed"], C4["dens"], C4["obs"], cols=cols)
    np.testing.assert_allclose(H1[:, cols], _h_slice(Yc, Mc.astype(np.float64), W0, H0[:, cols]), rtol=0, atol=1e-12)
    Yr, Mr = _hip.synthetic_reference(M, N, C4["seed"], C4["dens"], C4["obs"], rows=rows)
    Wn = _w_bracket(Yr, Mr.astype(np.float64), W0[:, rows], H1) / N
    np.testing.assert_allclose(W1[:, rows], Wn / Wn.sum(axis=0, keepdims=True), rtol=0, atol=1e-12)
    np.testing.assert_array_equal(l1, l3[:1])
    assert _monotone(np.concatenate([[loss0], l3])) and _monotone(ls)
    np.testing.assert_array_equal(l3, l3b)
    np.testing.assert_array_equal(W3, W3b)
    np.testing.assert_array_equal(H3, H3b)
    assert 2 <= ns <= C4["it_stop"]
    # ---- the same matrix as 8 ranks on this GPU: 4 processes x 2 ranks, three transports
    world, per = 8, 2
    transports = ("peer", "peer2", "host")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_c4_process, args=(p0, per, world, port, transports, q)) for p0 in range(0, world, per)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ranks = {}
    for _, out, errors in got:
        assert not errors, "\n".join(errors)
        ranks.update(out)
    assert sorted(ranks) == list(range(world))
    assert sum(ranks[k]["n_obs"] for k in ranks) == n_obs                     # the shards tile the same matrix
    for tr in transports:
        for k in range(world):
            l0, Ha, Wsamp = ranks[k][tr]["tol0"]
            r0, r1 = ranks[k]["rows"]
            np.testing.assert_allclose(l0, l3, rtol=1e-12, atol=0)               # the one-context curve
            np.testing.assert_allclose(Ha, H3, rtol=0, atol=1e-12)
            np.testing.assert_allclose(Wsamp, W3[:, r0:r1][:, ::61], rtol=0, atol=1e-12)
            np.testing.assert_array_equal(l0, ranks[0][tr]["tol0"][0])           # identical on every rank, bit for bit
            np.testing.assert_array_equal(Ha, ranks[0][tr]["tol0"][1])
            if tr != "host":
                n_k, l_k, H_k, W_k = ranks[k][tr]["stop"]
                assert n_k == ns                                                  # the stop rule fires at the same iteration
                np.testing.assert_allclose(l_k, ls, rtol=1e-12, atol=0)
                np.testing.assert_allclose(H_k, Hs, rtol=0, atol=1e-12)          # factors of iteration ns, not ns + 1
                np.testing.assert_allclose(W_k, Ws[:, r0:r1][:, ::61], rtol=0, atol=1e-12)
                np.testing.assert_array_equal(H_k, ranks[0][tr]["stop"][2])
    # peer and peer2 move the same sums in the same (rank) order: the same bits
    np.testing.assert_array_equal(ranks[0]["peer"]["tol0"][1], ranks[0]["peer2"]["tol0"][1])


def test_sweeps_are_cut_so_that_the_last_round_leaves_no_cu_empty():
    """nbmf_sweep_info: the chunking rules of DESIGN.md 4.1 on the shapes they were measured on -- configs[2] (2048
    workgroups per sweep: 4 full rounds of two per CU), its 8192-row shard (a short sweep at K = 64: ONE round of 512),
    configs[1] (K = 32 keeps 1024), and a shape with few strip groups (configs[4]'s W-pass: 266 x 8 = 2128 workgroups
    would run a fifth round for 80 of them; more chunks, a whole multiple of 8, until the rounds are >= 95 % full)."""
    from nbmf_mm_amd import _hip
    def info(m, n, k):
        with _hip.Context(m, n, k) as ctx:
            ctx.generate(1, density=0.25, observed=1.0)
            return ctx.sweep_info()
    i = info(65536, 8192, 64)
    assert (i["h_chunks"], i["w_chunks"]) == (16, 2) and i["h_blocks"] == 256 and i["w_blocks"] == 256
    i = info(8192, 8192, 64)
    assert (i["h_chunks"], i["w_chunks"]) == (4, 8) and i["h_blocks"] == 128     # (W: may be launched in two parts, per exchange panel)
    i = info(8192, 8192, 32)
    assert (i["h_chunks"], i["w_chunks"]) == (8, 8) and i["h_blocks"] == 64
    i = info(17000, 60000, 128)                      # (configs[4]'s row count; fewer columns keep the test light)
    groups_w = -(-17000 // 64)
    wgs = groups_w * i["w_chunks"]
    assert i["w_chunks"] % 8 == 0 and wgs / (-(-wgs // 512) * 512) >= 0.95


def test_fit_beside_a_second_tenant_that_saturates_the_chip_at_the_longest_sweeps():
    """configs[4]'s shape (internal 17000 x 360000, K = 128: 64 ms H sweeps, the longest of any configuration) while a SECOND
    TENANT keeps every SIMD of the chip busy with 40 ms launches of bare f64 MFMAs on another stream (a host thread calling
    nbmf_selftest_mfma_peak in a loop; a second PROCESS was tried first: this pool's GPUs serve one process's queues at a
    time and the two never met).  The fit's kernels then get onto the chip at the tenant's launch boundaries only -- with
    0.3 s tenant launches a 4096 x 8192 fit of 3 ms took 5.7 s and four iterations of this shape 336 s instead of 0.5 -- and
    a sweep's workgroups can be held back for longer than the 3 s the sweep's last workgroup waits for their
    log-likelihood partials (nbmf_pass_kernel.inc, PassFin).  Until round 6 that was an error ("loss assembly timed out":
    found by this test's first form); now the run resumes with the loss in a launch of its own (that form of the test:
    "resumed 1 time(s)", the fit's bits; the give-up path itself is tested deterministically by
    test_loss_assembly_inside_the_sweep_is_bounded_and_the_run_resumes).  Here, with launches short enough for a suite:
    the fit beside the tenant must come back -- slower -- with the losses and factors of the fit alone, bit for bit;
    whether it had to resume is reported, not asserted (it depends on how the two interleave)."""
    import threading
    import time
    from nbmf_mm_amd import _hip
    m, n, k = 17000, 360000, 128
    stop, rows = threading.Event(), []
    th = None
    try:
        with _hip.Context(m, n, k) as ctx:
            ctx.set_hyper(ALPHA, BETA, EPS)
            ctx.generate(seed=5, density=0.05, observed=0.9)
            g = np.random.default_rng(2)
            W = g.uniform(0.1, 0.9, (k, m))
            W /= W.sum(axis=0, keepdims=True)
            H = g.uniform(0.1, 0.9, (k, n))

            def fit():
                ctx.set_factors(W, H)
                t0 = time.perf_counter()
                losses, _ = ctx.run(2, 0.0)
                return np.array(losses), ctx.get_factors()[1][:, :4096].copy(), time.perf_counter() - t0
            fit()                                              # (warm-up: code objects, clocks)
            alone, H_alone, t_alone = fit()
            base = _hip.mfma_peak(0, 40.0)["cycles_per_mfma_at_2p4GHz"]

            def tenant():
                while not stop.is_set():
                    t0 = time.perf_counter()
                    p = _hip.mfma_peak(0, 40.0)
                    rows.append((t0, time.perf_counter(), p["cycles_per_mfma_at_2p4GHz"]))
            th = threading.Thread(target=tenant, daemon=True)
            th.start()
            time.sleep(1.0)                                    # (its launches are on the chip)
            resumed = _hip.variant_stats()[2]
            w0 = time.perf_counter()
            shared, H_shared, t_shared = fit()
            w1 = time.perf_counter()
            resumed = _hip.variant_stats()[2] - resumed
            stop.set()
            th.join(60)
    finally:
        stop.set()
        if th is not None:
            th.join(60)
    np.testing.assert_array_equal(shared, alone)
    np.testing.assert_array_equal(H_shared, H_alone)
    assert _monotone(list(shared))
    during = [c for (t, te, c) in rows if t <= w1 and te >= w0]      # tenant calls that overlap the two iterations
    print(f"2 iterations alone {t_alone:.2f} s, beside the tenant {t_shared:.2f} s, resumed {resumed} time(s); the tenant's cycles per MFMA "
          f"alone {base:.1f}, during the run: {len(during)} launches, the slowest at {max(during) if during else 0:.1f}")
    assert during and (max(during) > 1.2 * base or t_shared > 1.2 * t_alone), "the tenant and the fit never met on the chip"
