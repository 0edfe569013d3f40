"""world_size-2 gloo tests (CPU): the sharding arithmetic and the host-side plumbing of the multi-GPU
path.  The device work itself is covered by tests/test_gpu_dist.py on the GPU box."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    from nbmf_mm_amd import _dist
    from oracle import sharded_oracle
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        M, N, K = 301, 120, 7
        g = np.random.default_rng(11)
        Y = (g.random((M, N)) < 0.3).astype(np.float64)
        mask = (g.random((M, N)) < 0.85).astype(np.float64)
        r0, r1 = _dist.shard_bounds(M, world, rank)
        W, H = _dist.global_init(M, N, K, random_state=3)

        def allreduce(arr):
            dist.all_reduce(torch.from_numpy(arr), op=dist.ReduceOp.SUM)

        nobs = np.array([float(np.count_nonzero(mask[r0:r1]))])
        allreduce(nobs)
        Wl, Hn, losses = sharded_oracle.sharded_solve(Y[r0:r1], mask[r0:r1], W[:, r0:r1], H, 1.2, 1.3, nobs[0],
                                                      allreduce, max_iter=25)
        # the other split (columns of Y): local H-step, all-reduced W-step bracket
        c0, c1 = _dist.shard_bounds(N, world, rank)
        nobs2 = np.array([float(np.count_nonzero(mask[:, c0:c1]))])
        allreduce(nobs2)
        Wc, Hc, lc = sharded_oracle.sharded_solve_cols(Y[:, c0:c1], mask[:, c0:c1], W, H[:, c0:c1], 1.2, 1.3, nobs2[0], N,
                                                       allreduce, max_iter=25)
        # rendezvous object broadcast as attach_comm does it
        uid = [bytes(range(128)) if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        q.put((rank, r0, r1, Wl, Hn, losses, uid[0], (c0, c1, Wc, Hc, lc)))
    finally:
        dist.destroy_process_group()


def test_sharded_iteration_equals_unsharded():
    import torch.multiprocessing as mp
    from oracle import nbmf_oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    M, N, K = 301, 120, 7
    g = np.random.default_rng(11)
    Y = (g.random((M, N)) < 0.3).astype(np.float64)
    mask = (g.random((M, N)) < 0.85).astype(np.float64)
    Wr, Hr, lr, _, _ = orc.solve(Y, K, max_iter=25, tol=0, alpha=1.2, beta=1.3, mask=mask, random_state=3)
    assert [r[1:3] for r in res] == [(0, 150), (150, 301)]
    W = np.concatenate([r[3] for r in res], axis=1).T
    np.testing.assert_allclose(W, Wr, rtol=0, atol=1e-12)
    for r in res:
        np.testing.assert_allclose(r[4], Hr, rtol=0, atol=1e-12)       # H replicated and equal to the unsharded run
        np.testing.assert_allclose(r[5], lr, rtol=1e-12, atol=0)
        assert r[6] == bytes(range(128))
    np.testing.assert_array_equal(res[0][4], res[1][4])                # bitwise identical across ranks
    # column split
    assert [r[7][:2] for r in res] == [(0, 60), (60, 120)]
    Hc = np.concatenate([r[7][3] for r in res], axis=1)
    np.testing.assert_allclose(Hc, Hr, rtol=0, atol=1e-12)
    for r in res:
        np.testing.assert_allclose(r[7][2].T, Wr, rtol=0, atol=1e-12)  # W replicated
        np.testing.assert_allclose(r[7][4], lr, rtol=1e-12, atol=0)
    np.testing.assert_array_equal(res[0][7][2], res[1][7][2])


def test_shard_bounds_cover_and_balance():
    from nbmf_mm_amd import _dist
    for M, world in [(65536, 8), (301, 2), (1000, 3), (8, 8), (262144, 8)]:
        b = [_dist.shard_bounds(M, world, r) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == M
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        sizes = [y - x for x, y in b]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
    with pytest.raises(ValueError):
        _dist.shard_bounds(3, 4, 0)


def test_global_init_matches_reference_rule():
    from nbmf_mm_amd import _dist
    W, H = _dist.global_init(50, 30, 4, random_state=9)
    np.random.seed(9)
    W0 = np.random.uniform(0.1, 0.9, (50, 4))
    H0 = np.random.uniform(0.1, 0.9, (4, 30))
    np.testing.assert_array_equal(H, H0)
    np.testing.assert_array_equal(W, W0.T / W0.T.sum(axis=0, keepdims=True))
