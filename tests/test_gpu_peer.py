"""Peer (xGMI) transport on the GPU box: several processes share the one MI355X and map each other's
exchange arenas through HIP IPC, so the library's own exchange kernels (reduce-scatter fused with the
H-update, two-shot all-reduce, bounded waits) run exactly as they would across GPUs -- only the wires are
missing.  Checked against the single-process HIP run, the host-transport run and the CPU oracle."""
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


def _setup(rank, world, port):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.setdefault("NBMF_PEER_TIMEOUT_MS", "20000")
    from nbmf_mm_amd import _rendezvous
    return _rendezvous.Group(rank, world, ("tcp", "127.0.0.1", port), secret=b"tests-%d" % port)


def _worker_rows(rank, world, port, q, transport="peer"):
    dist = _setup(rank, world, port)
    from nbmf_mm_amd import _dist
    try:
        M, N, K, Y, mask = _problem()
        r0, r1 = _dist.shard_bounds(M, world, rank)
        out = {}
        for name, kw in [("tol0", dict(max_iter=30, tol=0)), ("stop", dict(max_iter=400, tol=1e-4))]:
            out[name] = _dist.fit_row_sharded(Y[r0:r1], M, r0, K, dist, alpha=1.2, beta=1.3, mask_local=mask[r0:r1],
                                              random_state=5, device=0, transport=transport, **kw)
        q.put((rank, r0, r1, out))
    finally:
        dist.close()


def _run(target, world, *extra):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,transport", [(2, "peer"), (3, "peer"), (2, "peer2"), (3, "peer2")])
def test_row_shards_peer_transport(world, transport):
    """Fused reduce-scatter + H-update + broadcast: factors and loss curve of the single-process run, the
    replicated factor bitwise equal on all ranks, and the stop rule firing at the same iteration.  "peer2": the
    exchange in two column panels on two streams (two flag slots), the second overlapped with the W-pass."""
    from nbmf_mm_amd import nbmf_mm_solver
    from oracle import nbmf_oracle as orc
    res = _run(_worker_rows, world, transport)
    M, N, K, Y, mask = _problem()
    W1, H1, l1, _, n1 = nbmf_mm_solver(Y, K, max_iter=30, tol=0, alpha=1.2, beta=1.3, mask=mask, random_state=5)
    Wr, Hr, lr, _, _ = orc.solve(Y, K, max_iter=30, tol=0, alpha=1.2, beta=1.3, mask=mask, random_state=5)
    W = np.concatenate([r[3]["tol0"][0] for r in res], axis=0)
    for ref_W, ref_H, ref_l, tol in [(W1, H1, l1, 1e-12), (Wr, Hr, lr, 1e-9)]:
        np.testing.assert_allclose(W, ref_W, rtol=0, atol=tol)
        np.testing.assert_allclose(res[0][3]["tol0"][1], ref_H, rtol=0, atol=tol)
        np.testing.assert_allclose(res[0][3]["tol0"][2], ref_l, rtol=1e-10, atol=0)
    for r in res[1:]:
        np.testing.assert_array_equal(r[3]["tol0"][1], res[0][3]["tol0"][1])
        np.testing.assert_array_equal(r[3]["tol0"][2], res[0][3]["tol0"][2])
    W2, H2, l2, _, n2 = nbmf_mm_solver(Y, K, max_iter=400, tol=1e-4, alpha=1.2, beta=1.3, mask=mask, random_state=5)
    assert 5 < n2 < 400
    for r in res:
        assert r[3]["stop"][3] == n2
        np.testing.assert_allclose(r[3]["stop"][1], H2, rtol=0, atol=1e-11)      # factors of iteration n2, not n2+1
        np.testing.assert_allclose(r[3]["stop"][2], l2, rtol=1e-10, atol=0)
    np.testing.assert_allclose(np.concatenate([r[3]["stop"][0] for r in res], axis=0), W2, rtol=0, atol=1e-11)


def _worker_axis1(rank, world, port, q):
    dist = _setup(rank, world, port)
    from nbmf_mm_amd import _dist, _hip
    try:
        M, N, K, Y, mask = _problem()
        V, Vmask = Y[:300, :], mask[:300, :]
        out = {}
        r0, r1 = _dist.shard_bounds(V.shape[0], world, rank)
        out["db_rows"] = (r0, r1) + _dist.fit_sharded(V[r0:r1], V.shape, r0, 9, dist, orientation="dir-beta", shard="rows",
                                                     max_iter=25, tol=0, mask_local=Vmask[r0:r1], random_state=4,
                                                     device=0, transport="peer")
        c0, c1 = _dist.shard_bounds(V.shape[1], world, rank)
        out["bd_cols"] = (c0, c1) + _dist.fit_sharded(V[:, c0:c1], V.shape, c0, 9, dist, orientation="beta-dir", shard="cols",
                                                     max_iter=200, tol=1e-4, mask_local=Vmask[:, c0:c1], random_state=4,
                                                     projection="duchi", device=0, transport="peer")
        # evaluation sweeps on a sharded context, and detach -> re-attach with another transport
        Wg, Hg = _dist.global_init(V.shape[0], V.shape[1], 9, random_state=4)
        with _hip.Context(V.shape[0], c1 - c0, 9) as ctx:
            ctx.set_hyper(1.2, 1.2, 1e-8, _hip.PROJ_DUCHI)
            ctx.upload(V[:, c0:c1], mask=Vmask[:, c0:c1])
            ctx.set_factors(Wg, np.ascontiguousarray(Hg[:, c0:c1]))
            used = _dist.attach_comm(ctx, dist, "auto", shard_axis=1)
            a = (used, ctx.loss(), ctx.loglik(), ctx.n_obs())
            la, _ = ctx.run(5, 0.0)
            ctx.comm_detach()
            ctx.set_factors(Wg, np.ascontiguousarray(Hg[:, c0:c1]))
            used2 = _dist.attach_comm(ctx, dist, "host", shard_axis=1)
            b = (used2, ctx.loss(), ctx.loglik(), ctx.n_obs())
            lb, _ = ctx.run(5, 0.0)
        out["eval"] = (a, b, la, lb)
        q.put((rank, out))
    finally:
        dist.close()


def test_column_split_peer_transport():
    from nbmf_mm_amd import nbmf_mm_solver
    res = [o for _, o in _run(_worker_axis1, 2)]
    M, N, K, Y, mask = _problem()
    V, Vmask = Y[:300, :], mask[:300, :]
    W1, H1, l1, _, _ = nbmf_mm_solver(V, 9, max_iter=25, tol=0, mask=Vmask, random_state=4, orientation="dir-beta")
    W = np.concatenate([r["db_rows"][2] for r in res], axis=0)
    np.testing.assert_allclose(W, W1, rtol=0, atol=1e-12)
    for r in res:
        np.testing.assert_allclose(r["db_rows"][3], H1, rtol=0, atol=1e-12)
        np.testing.assert_allclose(r["db_rows"][4], l1, rtol=1e-10, atol=0)
    np.testing.assert_array_equal(res[0]["db_rows"][3], res[1]["db_rows"][3])
    W2, H2, l2, _, n2 = nbmf_mm_solver(V, 9, max_iter=200, tol=1e-4, mask=Vmask, random_state=4, projection="duchi")
    Hc = np.concatenate([r["bd_cols"][3] for r in res], axis=1)
    np.testing.assert_allclose(Hc, H2, rtol=0, atol=1e-11)
    for r in res:
        assert r["bd_cols"][5] == n2
        np.testing.assert_allclose(r["bd_cols"][2], W2, rtol=0, atol=1e-11)
        np.testing.assert_allclose(r["bd_cols"][4], l2, rtol=1e-10, atol=0)
    assert sum(r["eval"][0][3] for r in res) == float(np.count_nonzero(Vmask))
    for r in res:
        a, b, la, lb = r["eval"]
        assert a[0] == "peer" and b[0] == "host"             # "auto" picks the peer transport when it works
        assert a[3] == b[3]                                  # this shard's own observed count
        np.testing.assert_allclose(a[1:3], b[1:3], rtol=1e-13)
        np.testing.assert_allclose(la, lb, rtol=1e-12)       # Duchi after detach: row counts were restored, not doubled


def _worker_absent_peer(rank, world, port, q):
    os.environ["NBMF_PEER_TIMEOUT_MS"] = "1500"
    dist = _setup(rank, world, port)
    from nbmf_mm_amd import _dist, _hip
    import time
    try:
        M, N, K, Y, mask = _problem()
        r0, r1 = _dist.shard_bounds(M, world, rank)
        W, H = _dist.global_init(M, N, K, random_state=5)
        with _hip.Context(r1 - r0, N, K) as ctx:
            ctx.set_hyper(1.2, 1.2)
            ctx.upload(Y[r0:r1], mask=mask[r0:r1])
            ctx.set_factors(np.ascontiguousarray(W[:, r0:r1]), H)
            table = dist.all_gather(ctx.peer_export(0))
            t0 = time.perf_counter()
            msg = None
            if rank == 0:                       # rank 1 never attaches: rank 0 must give up, not hang
                try:
                    ctx.comm_init_peer(b"".join(table), world, rank, 0)
                except _hip.NBMFHipError as e:
                    msg = str(e)
            dt = time.perf_counter() - t0
            # the context is unattached again and still works on its own shard
            losses, n_iter = ctx.run(3, 0.0)
            q.put((rank, msg, dt, n_iter))
        dist.barrier()
    finally:
        dist.close()


def test_missing_rank_times_out_instead_of_hanging():
    res = _run(_worker_absent_peer, 2)
    assert res[0][1] is not None and "timed out" in res[0][1]
    assert res[0][2] < 15.0
    assert res[0][3] == res[1][3] == 3


def _worker_k200(rank, world, port, q):
    dist = _setup(rank, world, port)
    from nbmf_mm_amd import _dist
    try:
        M, N, K, Y, mask = _problem()
        V, Vmask = Y[:300, :], mask[:300, :]
        out = {}
        r0, r1 = _dist.shard_bounds(V.shape[0], world, rank)
        c0, c1 = _dist.shard_bounds(V.shape[1], world, rank)
        for tr in ("peer", "host"):
            out["rows_" + tr] = _dist.fit_sharded(V[r0:r1], V.shape, r0, 200, dist, shard="rows", max_iter=12, tol=0,
                                                  mask_local=Vmask[r0:r1], random_state=4, device=0, transport=tr)
            out["cols_" + tr] = _dist.fit_sharded(V[:, c0:c1], V.shape, c0, 200, dist, shard="cols", max_iter=12, tol=0,
                                                  mask_local=Vmask[:, c0:c1], random_state=4, device=0, transport=tr)
        q.put((rank, out))
    finally:
        dist.close()


def test_sharded_more_than_128_components():
    """Slices (n_components > 128) under both splits and both transports, against the single-process run."""
    from nbmf_mm_amd import nbmf_mm_solver
    res = [o for _, o in _run(_worker_k200, 2)]
    M, N, K, Y, mask = _problem()
    V, Vmask = Y[:300, :], mask[:300, :]
    W1, H1, l1, _, _ = nbmf_mm_solver(V, 200, max_iter=12, tol=0, mask=Vmask, random_state=4)
    for tr in ("peer", "host"):
        W = np.concatenate([r["rows_" + tr][0] for r in res], axis=0)
        np.testing.assert_allclose(W, W1, rtol=0, atol=1e-12)
        H = np.concatenate([r["cols_" + tr][1] for r in res], axis=1)
        np.testing.assert_allclose(H, H1, rtol=0, atol=1e-12)
        for r in res:
            np.testing.assert_allclose(r["rows_" + tr][1], H1, rtol=0, atol=1e-12)
            np.testing.assert_allclose(r["cols_" + tr][0], W1, rtol=0, atol=1e-12)
            np.testing.assert_allclose(r["rows_" + tr][2], l1, rtol=1e-10, atol=0)
            np.testing.assert_allclose(r["cols_" + tr][2], l1, rtol=1e-10, atol=0)
        np.testing.assert_array_equal(res[0]["rows_" + tr][1], res[1]["rows_" + tr][1])


_SEQ = [("rows", 24), ("cols", 9), ("rows", 24), ("cols", 200), ("rows", 130), ("cols", 9), ("cols", 24), ("rows", 9)]


def _worker_sequence(rank, world, port, q):
    dist = _setup(rank, world, port)
    from nbmf_mm_amd import _dist
    try:
        M, N, K, Y, mask = _problem()
        V, Vmask = Y[:300, :], mask[:300, :]
        out = []
        for shard, k in _SEQ:
            lo, hi = _dist.shard_bounds(V.shape[0 if shard == "rows" else 1], world, rank)
            sl = (slice(lo, hi), slice(None)) if shard == "rows" else (slice(None), slice(lo, hi))
            out.append(_dist.fit_sharded(V[sl], V.shape, lo, k, dist, shard=shard, max_iter=6, tol=0, mask_local=Vmask[sl],
                                         random_state=4, device=0, transport="peer"))
        q.put((rank, out))
    finally:
        dist.close()


def test_many_sharded_fits_in_one_process():
    """Contexts come and go, arenas of different sizes and axes are exported again and again: every fit must see
    the peers' CURRENT arenas (the IPC-exported memory is pooled per process and never freed, so a handle cannot
    come to mean stale memory)."""
    from nbmf_mm_amd import nbmf_mm_solver
    res = [o for _, o in _run(_worker_sequence, 2)]
    M, N, K, Y, mask = _problem()
    V, Vmask = Y[:300, :], mask[:300, :]
    single = {}
    for i, (shard, k) in enumerate(_SEQ):
        if k not in single:
            single[k] = nbmf_mm_solver(V, k, max_iter=6, tol=0, mask=Vmask, random_state=4)
        W1, H1, l1 = single[k][:3]
        if shard == "rows":
            W = np.concatenate([r[i][0] for r in res], axis=0)
            H = res[0][i][1]
        else:
            W = res[0][i][0]
            H = np.concatenate([r[i][1] for r in res], axis=1)
        np.testing.assert_allclose(W, W1, rtol=0, atol=1e-12, err_msg=f"fit {i}: {shard} k={k}")
        np.testing.assert_allclose(H, H1, rtol=0, atol=1e-12, err_msg=f"fit {i}: {shard} k={k}")
        np.testing.assert_allclose(res[1][i][2], l1, rtol=1e-10, atol=0)


def test_bench_is_bounded_when_the_peer_transport_attaches_nowhere():
    """bench.py's transport selection with a peer transport that cannot exchange (fault injection: the last rank skips
    its part of every generic exchange, so the known-answer epochs at attach time starve): every rank must get the same
    answer within the short probe deadline -- not 30 s per attempt -- and the run goes on over the next transport.  RCCL
    refuses two ranks on one device, so on this box that is the host transport; the line says which."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NBMF_PEER_FAULT="1")
    env.pop("NBMF_PEER_TIMEOUT_MS", None)                 # the default (30 s) for the run itself, the probe deadline for the probes
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--M", "4096", "--N", "2048",
                        "--K", "32", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=300)
    wall = time.time() - t0
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["replicas_identical"] is True and line["loss_monotone"] is True
    assert not line["config"]["transport"].startswith("peer")          # the faulty transport was not chosen
    assert "peer" not in (line["config"]["transport_trials_s_per_5_iterations"] or {})
    assert wall < 90.0, f"transport selection with a dead peer transport took {wall:.0f} s"
    # the self-evidencing fields of an N > 1 line: the bound on the transport phase and what it took, what the transport in
    # use says about the job, and RCCL's own entry -- here "not timed", with the reason (two ranks on one device)
    cfg = line["config"]
    assert cfg["attempt"] == 0 and cfg["setup_bound_s"] == 100.0 and 0.0 < cfg["setup_s"] < cfg["setup_bound_s"]
    assert cfg["transport_sees"] == {"kind": "host", "nranks_seen": [2, 2]}
    assert line["rccl_value"] is None and line["rccl_nranks"] is None and "share a device" in line["rccl"]["error"]
    assert line["library"]["source_hash"] == line["library"]["tree_source_hash"]


def test_bench_line_still_comes_out_when_a_rank_never_leaves_the_transport_phase():
    """A rank that hangs while the transport is being selected (fault injection in bench.py: rank 1 of attempt 0 sleeps for
    good; rank 0 waits for it in a collective): both workers' watchdogs end the attempt at the phase's bound (15 s here, 100 s
    by default), both supervisors -- which never touch the GPU -- start FRESH workers over the host transport, and the line
    comes out with `transport_check` saying what failed on which rank."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NBMF_BENCH_FAULT="hang_in_setup:1", NBMF_BENCH_SETUP_BOUND_S="15")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--M", "4096", "--N", "2048",
                        "--K", "32", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=300)
    wall = time.time() - t0
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line, cfg = lines[0], lines[0]["config"]
    assert line["n_gpus"] == 2 and line["replicas_identical"] is True and line["loss_monotone"] is True and line["value"] > 0
    assert cfg["transport"] == "host" and cfg["attempt"] == 1
    assert "attempt 0 FAILED" in cfg["transport_check"] and "rank 1: exit code 75: set-up exceeded its bound of 15 s" in cfg["transport_check"]
    assert "rank 0: exit code" in cfg["transport_check"]
    assert "host transport" in line["rccl"]["error"]
    assert wall < 120.0, f"{wall:.0f} s"


_N_GPUS_SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from nbmf_mm_amd import NBMF, nbmf_mm_solver, _hip
g = np.random.default_rng(31)
M, N, K = 2100, 640, 64
X = (g.random((M, N)) < 0.3).astype(np.float64)
mask = g.random((M, N)) < 0.9
out = {}
for orientation in ("beta-dir", "dir-beta"):
    for name, kw in (("tol0", dict(max_iter=12, tol=0.0)), ("stop", dict(max_iter=300, tol=2e-4))):
        one = NBMF(n_components=K, random_state=2, orientation=orientation, alpha=1.2, beta=1.3, **kw).fit(X, mask=mask)
        nxt1 = np.random.uniform()
        many = NBMF(n_components=K, random_state=2, orientation=orientation, alpha=1.2, beta=1.3, n_gpus=8, devices=[0] * 8,
                    **kw).fit(X, mask=mask)
        nxt8 = np.random.uniform()
        l1, l8 = np.array(one.loss_curve_), np.array(many.loss_curve_)
        out[orientation + "/" + name] = dict(
            n_iter=(int(one.n_iter_), int(many.n_iter_)), max_iter=kw["max_iter"],
            loss=float(np.max(np.abs(l1 - l8) / np.abs(l1))) if len(l1) == len(l8) else None,
            W=float(np.abs(one.W_ - many.W_).max()), H=float(np.abs(one.components_ - many.components_).max()),
            shapes=[list(many.W_.shape), list(many.components_.shape)], rng=(nxt1 == nxt8))
    # the function entry, uint8 data, normalised alias: the same fit once more
    W, H, losses, t, n_it = nbmf_mm_solver(X.astype(np.uint8), K, max_iter=5, tol=0, random_state=2, mask=mask,
                                           orientation=orientation, alpha=1.2, beta=1.3, n_gpus=8, devices=[0] * 8)
    ref = nbmf_mm_solver(X, K, max_iter=5, tol=0, random_state=2, mask=mask, orientation=orientation, alpha=1.2, beta=1.3)
    out[orientation + "/solver"] = dict(t=t, n_iter=n_it, loss=float(np.max(np.abs(np.array(losses) - np.array(ref[2])) / np.abs(ref[2]))),
                                        W=float(np.abs(W - ref[0]).max()), H=float(np.abs(H - ref[1]).max()))
# other storage paths through the same entry: real-valued data with real weights (16 bytes per entry), with a bool mask
# (8 bytes, mask folded in), and a scipy CSR matrix with a CSR pattern mask (never densified; sliced by rows per rank)
import scipy.sparse as sp
Xr, wts = g.random((900, 300)), g.random((900, 300))
mb = g.random((900, 300)) < 0.85
Xs = sp.csr_matrix((g.random((900, 300)) < 0.1).astype(np.float64))
for name, data, mk in (("weights", Xr, wts), ("folded", Xr, mb), ("csr", Xs, sp.csr_matrix(mb.astype(np.float64)))):
    kw = dict(n_components=24, random_state=5, max_iter=10, tol=0.0, orientation="dir-beta" if name == "folded" else "beta-dir")
    one = NBMF(**kw).fit(data, mask=mk)
    many = NBMF(n_gpus=4, devices=[0] * 4, **kw).fit(data, mask=mk)
    l1, l4 = np.array(one.loss_curve_), np.array(many.loss_curve_)
    out["paths/" + name] = dict(loss=float(np.max(np.abs(l1 - l4) / np.abs(l1))), W=float(np.abs(one.W_ - many.W_).max()),
                                H=float(np.abs(one.components_ - many.components_).max()))
# HETEROGENEOUS shards (round 5): the form in which a rank's H sweep leaves its products depends on that rank's own
# shard -- byte codes or doubles, factors in range or not -- and the ranks' products are summed.  (a) real-valued V whose
# first quarter of rows happens to be all 0 / 1: rank 0 of 4 stores byte codes (mapped products), ranks 1-3 doubles;
# (b) binary V with ONE negative entry in W_init: the rank that holds that row leaves the plain variant, the others stay.
Xh = g.random((900, 300))
Xh[:225] = Xh[:225] < 0.3
for name, data, mk, extra in (("het_storage", Xh, mb, {}),
                              ("het_variant", (Xr < 0.3).astype(np.float64), mb, "w_init")):
    kw = dict(n_components=24, random_state=5, max_iter=10, tol=0.0)
    if extra == "w_init":
        W0 = np.random.default_rng(3).uniform(0.1, 0.9, (900, 24))
        W0[10, 3] = -0.05
        kw.update(W_init=W0, H_init=np.random.default_rng(4).uniform(0.1, 0.9, (24, 300)))
    one = NBMF(**kw).fit(data, mask=mk)
    many = NBMF(n_gpus=4, devices=[0] * 4, **kw).fit(data, mask=mk)
    l1, l4 = np.array(one.loss_curve_), np.array(many.loss_curve_)
    out["paths/" + name] = dict(loss=float(np.max(np.abs(l1 - l4) / np.abs(l1))), W=float(np.abs(one.W_ - many.W_).max()),
                                H=float(np.abs(one.components_ - many.components_).max()))
try:
    NBMF(n_components=K, n_gpus=3, devices=[0, 0]).fit(X)
    out["bad_devices"] = "no error"
except ValueError as e:
    out["bad_devices"] = str(e)
print("RESULT " + json.dumps(out))
"""


def test_n_gpus_behind_the_drop_in_api_eight_ranks_in_one_process():
    """``NBMF(n_components=64, n_gpus=8)``: the sharded fit behind the reference's own estimator API, ONE process -- eight
    rank threads, a context and stream each, the peer transport addressing the other ranks' arenas directly (same
    process: no IPC).  All eight ranks on device 0 here (the box has one GPU; `devices=[0] * 8`), which needs hardware
    queues of their own for the ranks' streams -- hence the child process with GPU_MAX_HW_QUEUES set before HIP starts.
    Against ``n_gpus=1``: loss curves, W_ and components_ to 1e-12, both orientations (rows of V = rows, or columns, of
    the internal matrix), the stop rule firing at the same iteration, and the global generator left in the same state.
    NOT measured across physical devices (no multi-GPU machine in reach)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="32", NBMF_PERSISTENT="0", NBMF_PEER_TIMEOUT_MS="20000")
    r = subprocess.run([sys.executable, "-c", _N_GPUS_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    for orientation in ("beta-dir", "dir-beta"):
        for name in ("tol0", "stop"):
            o = out[f"{orientation}/{name}"]
            assert o["n_iter"][0] == o["n_iter"][1], o
            assert (name == "tol0") == (o["n_iter"][0] == o["max_iter"]), o       # the stop rule fired in the "stop" runs
            assert o["loss"] is not None and o["loss"] <= 1e-12 and o["W"] <= 1e-12 and o["H"] <= 1e-12, o
            assert o["shapes"] == [[2100, 64], [64, 640]] and o["rng"] is True
        o = out[f"{orientation}/solver"]
        assert o["t"] == 0.0 and o["n_iter"] == 5 and o["loss"] <= 1e-12 and o["W"] <= 1e-12 and o["H"] <= 1e-12, o
    for name in ("weights", "folded", "csr", "het_storage", "het_variant"):
        o = out["paths/" + name]
        assert o["loss"] <= 1e-12 and o["W"] <= 1e-12 and o["H"] <= 1e-12, (name, o)
    assert "devices names 2 GPUs" in out["bad_devices"]
    # without the hardware queues the entry refuses instead of risking ranks that wait for each other on one queue
    from nbmf_mm_amd import _dist
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 16:
        with pytest.raises(ValueError, match="GPU_MAX_HW_QUEUES"):
            _dist.fit_in_process(np.zeros((64, 32)), 4, 8, devices=[0] * 8, max_iter=1)


def test_bench_line_still_comes_out_when_an_exchange_fails_in_the_timed_region():
    """The other way an attempt ends: an error out of the timed region on ONE rank (fault injection: what an exchange that
    times out in mid-run raises).  That worker leaves with the retry code, the other one loses its rendezvous (the relay
    notices the closed socket) and leaves too, both supervisors start fresh workers over the host transport, and the ONE
    line says which rank failed with what."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NBMF_BENCH_FAULT="fail_in_run:0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--M", "4096", "--N", "2048",
                        "--K", "32", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    cfg = lines[0]["config"]
    assert cfg["transport"] == "host" and cfg["attempt"] == 1 and lines[0]["replicas_identical"] is True
    assert "rank 0: exit code 75" in cfg["transport_check"] and "injected: the exchange timed out" in cfg["transport_check"]
    assert "rank 1: exit code" in cfg["transport_check"]
