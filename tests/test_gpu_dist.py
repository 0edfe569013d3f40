"""Multi-rank path on the GPU box: two processes share the one MI355X, all-reduce over the host
transport (gloo); plus the RCCL plumbing with a single rank.  Compared with the single-process HIP
run and the CPU oracle."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _single_process_reference_on_the_five_kernel_path(monkeypatch):
    """The sharded runs below are compared, some of them bit for bit, with single-process runs of the same small
    problems; those would take the single-launch path (another order of additions), so it is switched off here --
    tests/test_gpu_parity.py and tests/test_gpu_api.py hold the two engines against each other and the oracle."""
    monkeypatch.setenv("NBMF_PERSISTENT", "0")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    g = np.random.default_rng(21)
    M, N, K = 700, 333, 24
    Y = (g.random((M, N)) < 0.3).astype(np.float64)
    mask = g.random((M, N)) < 0.9
    return M, N, K, Y, mask


def _worker_dirbeta_restarts(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from nbmf_mm_amd import _dist, _rendezvous
    dist = _rendezvous.Group(rank, world, ("tcp", "127.0.0.1", port), secret=b"tests-%d" % port)
    try:
        M, N, K, Y, mask = _problem()
        V, Vmask = Y[:300, :], mask[:300, :]                 # dir-beta on a 300 x 333 matrix, column shards
        c0, c1 = _dist.shard_bounds(V.shape[1], world, rank)
        W, Hl, losses, n_iter = _dist.fit_sharded(V[:, c0:c1], V.shape, c0, 9, dist, orientation="dir-beta", shard="cols",
                                                  max_iter=25, tol=0, mask_local=Vmask[:, c0:c1], random_state=4,
                                                  device=0, transport="host")
        best = _dist.fit_restarts(V, 9, dist, n_init=3, random_state=10, device=0, max_iter=15, tol=0, mask=Vmask)
        q.put((rank, c0, c1, W, Hl, losses, best))
    finally:
        dist.close()


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from nbmf_mm_amd import _dist, _rendezvous
    dist = _rendezvous.Group(rank, world, ("tcp", "127.0.0.1", port), secret=b"tests-%d" % port)
    try:
        M, N, K, Y, mask = _problem()
        r0, r1 = _dist.shard_bounds(M, world, rank)
        Wl, H, losses, n_iter = _dist.fit_row_sharded(Y[r0:r1], M, r0, K, dist, max_iter=30, tol=0, alpha=1.2, beta=1.3,
                                                      mask_local=mask[r0:r1], random_state=5, device=0, transport="host")
        q.put((rank, r0, r1, Wl, H, losses, n_iter))
    finally:
        dist.close()


def test_two_ranks_one_gpu_host_transport():
    import multiprocessing as mp
    from nbmf_mm_amd import nbmf_mm_solver
    from oracle import nbmf_oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    M, N, K, Y, mask = _problem()
    W1, H1, l1, _, n1 = nbmf_mm_solver(Y, K, max_iter=30, tol=0, alpha=1.2, beta=1.3, mask=mask, random_state=5)
    Wr, Hr, lr, _, _ = orc.solve(Y, K, max_iter=30, tol=0, alpha=1.2, beta=1.3, mask=mask, random_state=5)
    W = np.concatenate([r[3] for r in res], axis=0)
    for ref_W, ref_H, ref_l, tol in [(W1, H1, l1, 1e-12), (Wr, Hr, lr, 1e-9)]:
        np.testing.assert_allclose(W, ref_W, rtol=0, atol=tol)
        np.testing.assert_allclose(res[0][4], ref_H, rtol=0, atol=tol)
        np.testing.assert_allclose(res[0][5], ref_l, rtol=1e-10, atol=0)
    np.testing.assert_array_equal(res[0][4], res[1][4])     # replicated factor bitwise identical on both ranks
    np.testing.assert_array_equal(res[0][5], res[1][5])
    assert res[0][6] == res[1][6] == 30 == n1


def test_two_ranks_stop_rule_agrees():
    """Sharded stop rule: every rank stops at the same iteration as the single-process run."""
    from nbmf_mm_amd import _hip, _dist, nbmf_mm_solver
    M, N, K, Y, mask = _problem()
    # emulate two ranks in ONE process: two contexts, the host callback sums their buffers
    r = [_dist.shard_bounds(M, 2, i) for i in range(2)]
    W, H = _dist.global_init(M, N, K, random_state=5)
    _, _, l1, _, n1 = nbmf_mm_solver(Y, K, max_iter=400, tol=1e-4, mask=mask, random_state=5)
    assert 5 < n1 < 400
    # a single-rank "world" over the host transport must reproduce the plain run exactly
    with _hip.Context(M, N, K) as ctx:
        ctx.set_hyper(1.2, 1.2)
        ctx.upload(Y, mask=mask)
        ctx.set_factors(W, H)
        ctx.comm_init_host(lambda arr: None, 1, 0)
        losses, n_iter = ctx.run(400, 1e-4)
    assert n_iter == n1
    np.testing.assert_array_equal(losses, np.array(l1))


def test_rccl_single_rank():
    """RCCL plumbing (dlopen, unique id, communicator, in-place all-reduce on the library's stream)."""
    from nbmf_mm_amd import _hip, _dist, nbmf_mm_solver
    M, N, K, Y, mask = _problem()
    W, H = _dist.global_init(M, N, K, random_state=5)
    _, _, l1, _, _ = nbmf_mm_solver(Y, K, max_iter=10, tol=0, mask=mask, random_state=5)
    uid = _hip.comm_unique_id()
    assert len(uid) == 128
    with _hip.Context(M, N, K) as ctx:
        ctx.set_hyper(1.2, 1.2)
        ctx.upload(Y, mask=mask)
        ctx.set_factors(W, H)
        ctx.comm_init(uid, 1, 0)
        assert ctx.n_obs() == np.count_nonzero(mask)
        losses, n_iter = ctx.run(10, 0.0)
    np.testing.assert_array_equal(losses, np.array(l1))
    # same through the column-split code path (W-step exchange + scalar exchange), one rank
    with _hip.Context(M, N, K) as ctx:
        ctx.set_hyper(1.2, 1.2)
        ctx.upload(Y, mask=mask)
        ctx.set_factors(W, H)
        ctx.comm_init(_hip.comm_unique_id(), 1, 0, shard_axis=1)
        losses, n_iter = ctx.run(10, 0.0)
        assert ctx.loss() == losses[-1]
    np.testing.assert_array_equal(losses, np.array(l1))


def test_dir_beta_column_shards_and_parallel_restarts():
    import multiprocessing as mp
    from nbmf_mm_amd import nbmf_mm_solver
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dirbeta_restarts, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    M, N, K, Y, mask = _problem()
    V, Vmask = Y[:300, :], mask[:300, :]
    W1, H1, l1, _, _ = nbmf_mm_solver(V, 9, max_iter=25, tol=0, mask=Vmask, random_state=4, orientation="dir-beta")
    H = np.concatenate([r[4] for r in res], axis=1)
    np.testing.assert_allclose(H, H1, rtol=0, atol=1e-12)
    np.testing.assert_allclose(H.sum(axis=0), 1.0, atol=1e-12)          # dir-beta: columns of H on the simplex
    for r in res:
        np.testing.assert_allclose(r[3], W1, rtol=0, atol=1e-12)        # Beta factor replicated
        np.testing.assert_allclose(r[5], l1, rtol=1e-10, atol=0)
    # restarts: both ranks agree on the winner, which equals the best of the three sequential runs
    seq = [nbmf_mm_solver(V, 9, max_iter=15, tol=0, mask=Vmask, random_state=10 + i) for i in range(3)]
    k = int(np.argmin([s_[2][-1] for s_ in seq]))
    for r in res:
        Wb, Hb, lb, nb, ib = r[6]
        assert ib == k and nb == 15
        np.testing.assert_array_equal(Wb, seq[k][0])
        np.testing.assert_array_equal(lb, seq[k][2])


def _worker_axis1(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from nbmf_mm_amd import _dist, _rendezvous
    dist = _rendezvous.Group(rank, world, ("tcp", "127.0.0.1", port), secret=b"tests-%d" % port)
    try:
        M, N, K, Y, mask = _problem()
        V, Vmask = Y[:300, :], mask[:300, :]
        out = {}
        # dir-beta with V split by ROWS (the all-reduce moves to the W-step)
        r0, r1 = _dist.shard_bounds(V.shape[0], world, rank)
        out["db_rows"] = (r0, r1) + _dist.fit_sharded(V[r0:r1], V.shape, r0, 9, dist, orientation="dir-beta", shard="rows",
                                                     max_iter=25, tol=0, mask_local=Vmask[r0:r1], random_state=4,
                                                     device=0, transport="host")
        # beta-dir with V split by COLUMNS, Duchi projection (needs the global per-row observed counts), stop rule on
        c0, c1 = _dist.shard_bounds(V.shape[1], world, rank)
        out["bd_cols"] = (c0, c1) + _dist.fit_sharded(V[:, c0:c1], V.shape, c0, 9, dist, orientation="beta-dir", shard="cols",
                                                     max_iter=200, tol=1e-4, mask_local=Vmask[:, c0:c1], random_state=4,
                                                     projection="duchi", device=0, transport="host")
        q.put((rank, out))
    finally:
        dist.close()


def test_column_split_of_the_internal_matrix():
    import multiprocessing as mp
    from nbmf_mm_amd import nbmf_mm_solver
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_axis1, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [o for _, o in sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    M, N, K, Y, mask = _problem()
    V, Vmask = Y[:300, :], mask[:300, :]
    # dir-beta / rows: W (Beta factor, M x k) comes back in row slices, H (k x N, simplex columns) whole
    W1, H1, l1, _, _ = nbmf_mm_solver(V, 9, max_iter=25, tol=0, mask=Vmask, random_state=4, orientation="dir-beta")
    W = np.concatenate([r["db_rows"][2] for r in res], axis=0)
    np.testing.assert_allclose(W, W1, rtol=0, atol=1e-12)
    for r in res:
        np.testing.assert_allclose(r["db_rows"][3], H1, rtol=0, atol=1e-12)
        np.testing.assert_allclose(r["db_rows"][4], l1, rtol=1e-10, atol=0)
    np.testing.assert_array_equal(res[0]["db_rows"][3], res[1]["db_rows"][3])      # replicated factor bitwise equal
    # beta-dir / cols with Duchi and the stop rule
    W2, H2, l2, _, n2 = nbmf_mm_solver(V, 9, max_iter=200, tol=1e-4, mask=Vmask, random_state=4, projection="duchi")
    assert 3 < n2 < 200
    Hc = np.concatenate([r["bd_cols"][3] for r in res], axis=1)
    np.testing.assert_allclose(Hc, H2, rtol=0, atol=1e-11)
    for r in res:
        assert r["bd_cols"][5] == n2
        np.testing.assert_allclose(r["bd_cols"][2], W2, rtol=0, atol=1e-11)
        np.testing.assert_allclose(r["bd_cols"][4], l2, rtol=1e-10, atol=0)


def test_three_ranks_uneven_shards():
    """world = 3 (uneven 233/233/234 row shards, one GPU shared, host transport) against the single-process run."""
    import multiprocessing as mp
    from nbmf_mm_amd import nbmf_mm_solver
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    M, N, K, Y, mask = _problem()
    assert [r[2] - r[1] for r in res] == [233, 233, 234]
    W1, H1, l1, _, _ = nbmf_mm_solver(Y, K, max_iter=30, tol=0, alpha=1.2, beta=1.3, mask=mask, random_state=5)
    np.testing.assert_allclose(np.concatenate([r[3] for r in res], axis=0), W1, rtol=0, atol=1e-12)
    for r in res:
        np.testing.assert_allclose(r[4], H1, rtol=0, atol=1e-12)
        np.testing.assert_allclose(r[5], l1, rtol=1e-10, atol=0)
        np.testing.assert_array_equal(r[4], res[0][4])


def _wide_problem():
    g = np.random.default_rng(31)
    M, N, K = 300, 2100, 10          # wide enough that the W-pass has several chunks -> two exchange panels
    Y = (g.random((M, N)) < 0.3).astype(np.float64)
    mask = g.random((M, N)) < 0.9
    return M, N, K, Y, mask


def _worker_wide(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from nbmf_mm_amd import _dist, _rendezvous
    dist = _rendezvous.Group(rank, world, ("tcp", "127.0.0.1", port), secret=b"tests-%d" % port)
    try:
        M, N, K, Y, mask = _wide_problem()
        r0, r1 = _dist.shard_bounds(M, world, rank)
        Wl, H, losses, n_iter = _dist.fit_row_sharded(Y[r0:r1], M, r0, K, dist, max_iter=200, tol=1e-4, mask_local=mask[r0:r1],
                                                      random_state=2, device=0, transport="host")
        q.put((rank, Wl, H, losses, n_iter))
    finally:
        dist.close()


def test_two_exchange_panels(capfd, monkeypatch):
    """The K x N exchange cut into two column panels (second panel's all-reduce overlapping the first
    panel's H-update and W-pass share): RCCL with one rank (two streams + events) and two ranks over the
    host transport, against the single-process run, stop rule included."""
    import multiprocessing as mp
    from nbmf_mm_amd import _hip, _dist, nbmf_mm_solver
    M, N, K, Y, mask = _wide_problem()
    W1, H1, l1, _, n1 = nbmf_mm_solver(Y, K, max_iter=200, tol=1e-4, mask=mask, random_state=2)
    assert 5 < n1 < 200
    monkeypatch.setenv("NBMF_DEBUG", "1")
    monkeypatch.setenv("NBMF_OVERLAP", "1")          # opt-in (the spawned ranks below inherit it)
    W0, H0 = _dist.global_init(M, N, K, random_state=2)
    with _hip.Context(M, N, K) as ctx:
        ctx.set_hyper(1.2, 1.2)
        ctx.upload(Y, mask=mask)
        ctx.set_factors(W0, H0)
        ctx.comm_init(_hip.comm_unique_id(), 1, 0)
        losses, n_iter = ctx.run(200, 1e-4)
        Wk, Hk = ctx.get_factors()
    err = capfd.readouterr().err
    assert "2 exchange panel(s)" in err and "RCCL" in err
    monkeypatch.delenv("NBMF_DEBUG")
    assert n_iter == n1
    # (the prior partial sums are grouped per panel here: last-bit differences from the plain run)
    np.testing.assert_allclose(losses, np.array(l1), rtol=1e-13, atol=0)
    np.testing.assert_allclose(Wk.T, W1, rtol=0, atol=1e-13)
    np.testing.assert_allclose(Hk, H1, rtol=0, atol=1e-13)
    ctx_mp = mp.get_context("spawn")
    q = ctx_mp.Queue()
    port = _free_port()
    procs = [ctx_mp.Process(target=_worker_wide, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    np.testing.assert_allclose(np.concatenate([r[1] for r in res], axis=0), W1, rtol=0, atol=1e-12)
    for r in res:
        assert r[4] == n1
        np.testing.assert_allclose(r[2], H1, rtol=0, atol=1e-12)
        np.testing.assert_allclose(r[3], l1, rtol=1e-10, atol=0)
    np.testing.assert_array_equal(res[0][2], res[1][2])
