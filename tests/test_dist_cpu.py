"""world_size-2 (gloo and the product's own group) and world_size-8 tests (CPU): the sharding arithmetic and the host-side plumbing of the multi-GPU
path.  The device work itself is covered by tests/test_gpu_dist.py on the GPU box."""
import os
import socket
import time
from struct import error as struct_error

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, kind):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    from gloo_group import make_group
    from nbmf_mm_amd import _dist
    from oracle import sharded_oracle
    dist = make_group(kind, rank, world, port)
    try:
        M, N, K = 301, 120, 7
        g = np.random.default_rng(11)
        Y = (g.random((M, N)) < 0.3).astype(np.float64)
        mask = (g.random((M, N)) < 0.85).astype(np.float64)
        r0, r1 = _dist.shard_bounds(M, world, rank)
        W, H = _dist.global_init(M, N, K, random_state=3)

        def allreduce(arr):
            dist.all_reduce(arr, "sum")

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
        uid = dist.broadcast(bytes(range(128)) if rank == 0 else None, src=0)
        q.put((rank, r0, r1, Wl, Hn, losses, uid, (c0, c1, Wc, Hc, lc)))
    finally:
        dist.close()


@pytest.mark.parametrize("kind,world", [("gloo", 2), ("stdlib", 2), ("stdlib", 8)])
def test_sharded_iteration_equals_unsharded(kind, world):
    """(8 ranks: the largest launch the benchmark driver makes, on the CPU)"""
    import multiprocessing as mp
    from oracle import nbmf_oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind)) for r in range(world)]
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
    from nbmf_mm_amd import _dist
    assert [r[1:3] for r in res] == [_dist.shard_bounds(M, world, k) for k in range(world)]
    if world == 2:
        assert [r[1:3] for r in res] == [(0, 150), (150, 301)]
    W = np.concatenate([r[3] for r in res], axis=1).T
    np.testing.assert_allclose(W, Wr, rtol=0, atol=1e-12)
    for r in res:
        np.testing.assert_allclose(r[4], Hr, rtol=0, atol=1e-12)       # H replicated and equal to the unsharded run
        np.testing.assert_allclose(r[5], lr, rtol=1e-12, atol=0)
        assert r[6] == bytes(range(128))
    for r in res[1:]:
        np.testing.assert_array_equal(res[0][4], r[4])                 # bitwise identical across ranks
    # column split
    assert [r[7][:2] for r in res] == [_dist.shard_bounds(N, world, k) for k in range(world)]
    Hc = np.concatenate([r[7][3] for r in res], axis=1)
    np.testing.assert_allclose(Hc, Hr, rtol=0, atol=1e-12)
    for r in res:
        np.testing.assert_allclose(r[7][2].T, Wr, rtol=0, atol=1e-12)  # W replicated
        np.testing.assert_allclose(r[7][4], lr, rtol=1e-12, atol=0)
    for r in res[1:]:
        np.testing.assert_array_equal(res[0][7][2], r[7][2])


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


class _StubCtx:
    """Stands in for a _hip.Context in the transport-negotiation test: records the calls attach_comm makes and
    fails where told to."""

    def __init__(self, fail):
        self.fail, self.calls = set(fail), []

    def peer_export(self, axis):
        self.calls.append("export")
        if "export" in self.fail:
            from nbmf_mm_amd import _hip
            raise _hip.NBMFHipError("no IPC here")
        return bytes(128)

    def comm_init_peer(self, handles, world, rank, axis):
        self.calls.append("peer")
        assert len(handles) == 128 * world
        if "peer" in self.fail:
            from nbmf_mm_amd import _hip
            raise _hip.NBMFHipError("peer transport self-test failed")
        if "peer_arg" in self.fail:       # a capability limit: NBMF_ERR_ARG -> ValueError (_hip._check)
            raise ValueError("the peer transport supports at most 16 ranks")

    def comm_init(self, uid, world, rank, axis):
        self.calls.append("rccl")
        if "rccl" in self.fail:
            from nbmf_mm_amd import _hip
            raise _hip.NBMFHipError("ncclCommInitRank failed")

    def comm_init_host(self, fn, world, rank, axis):
        self.calls.append("host")

    def comm_detach(self):
        self.calls.append("detach")

    def set_exchange_panels(self, panels=0):     # per-context settings of the real Context: recorded apart from the calls
        self.panels = panels

    def set_peer_timeout_ms(self, ms=0.0):
        self.peer_timeout_ms = ms


def _worker_negotiate(rank, world, port, q, kind):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    from gloo_group import make_group
    from nbmf_mm_amd import _dist, _hip
    dist = make_group(kind, rank, world, port)
    _hip.comm_unique_id = lambda: bytes(128)          # no librccl call on the CPU box
    try:
        out = {}
        # everything works: peer wins, nothing else is touched
        c = _StubCtx([])
        out["ok"] = (_dist.attach_comm(c, dist, "auto"), c.calls)
        # the peer attach fails on rank 1 only: rank 0 must detach and BOTH move on to RCCL
        c = _StubCtx(["peer"] if rank == 1 else [])
        out["peer_fails_on_1"] = (_dist.attach_comm(c, dist, "auto"), c.calls)
        # no IPC on rank 0, RCCL init fails on rank 1: both end on the host transport
        c = _StubCtx((["export"] if rank == 0 else []) + (["rccl"] if rank == 1 else []))
        out["down_to_host"] = (_dist.attach_comm(c, dist, "auto"), c.calls)
        # a capability limit (ValueError) on ONE rank only: still a vote, both fall back to RCCL together
        c = _StubCtx(["peer_arg"] if rank == 0 else [])
        out["capability_limit"] = (_dist.attach_comm(c, dist, "auto"), c.calls)
        # an explicit transport does not fall back: every rank raises
        c = _StubCtx(["peer"] if rank == 0 else [])
        try:
            _dist.attach_comm(c, dist, "peer")
            out["explicit"] = ("no error", c.calls)
        except _hip.NBMFHipError as e:
            out["explicit"] = ("raised", c.calls)
        q.put((rank, out))
    finally:
        dist.close()


@pytest.mark.parametrize("kind", ["gloo", "stdlib"])
def test_transport_negotiation_never_splits_the_job(kind):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_negotiate, args=(r, 2, port, q, kind)) for r in range(2)]
    for p in procs:
        p.start()
    res = [o for _, o in sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert [r["ok"] for r in res] == [("peer", ["export", "peer"])] * 2
    assert res[0]["peer_fails_on_1"] == ("rccl", ["export", "peer", "detach", "rccl"])
    assert res[1]["peer_fails_on_1"] == ("rccl", ["export", "peer", "rccl"])
    assert res[0]["down_to_host"] == ("host", ["export", "rccl", "detach", "host"])
    assert res[1]["down_to_host"] == ("host", ["export", "rccl", "host"])
    assert res[0]["capability_limit"] == ("rccl", ["export", "peer", "rccl"])
    assert res[1]["capability_limit"] == ("rccl", ["export", "peer", "detach", "rccl"])
    assert res[0]["explicit"] == ("raised", ["export", "peer"])
    assert res[1]["explicit"] == ("raised", ["export", "peer", "detach"])


def _worker_fastest(rank, world, port, q, kind):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import time
    sys.path.insert(0, os.path.join(root, "tests"))
    from gloo_group import make_group
    from nbmf_mm_amd import _dist, _hip
    dist = make_group(kind, rank, world, port)
    _hip.comm_unique_id = lambda: bytes(128)

    class Ctx(_StubCtx):
        """Stub whose iterations are slow over the peer transport on rank 1 only."""
        def __init__(self, fail=()):
            super().__init__(fail)
            self.attached = None

        def comm_init_peer(self, *a):
            super().comm_init_peer(*a)
            self.attached = "peer"

        def comm_init(self, *a):
            super().comm_init(*a)
            self.attached = "rccl"

        def comm_detach(self):
            super().comm_detach()
            self.attached = None

        def run(self, n, tol):
            self.calls.append(f"run{n}")
            # fault injection AFTER a clean attach, on one rank only: the peer transport's exchange times out in the
            # warm-up ("run_warmup") or in the timed run ("run_timed") -- NBMFHipError out of ctx.run
            if self.attached == "peer" and (("run_warmup" in self.fail and n == 2) or ("run_timed" in self.fail and n != 2)):
                raise _hip.NBMFHipError("peer exchange timed out")
            if self.attached == "peer" and rank == 1:
                time.sleep(0.05 * n)

        def synchronize(self):
            pass

    try:
        out = {}
        c = Ctx()
        resets = []
        out["slow_peer"] = _dist.attach_fastest(c, dist, lambda: resets.append(1), candidates=("peer", "rccl", "rccl2"),
                                                iters=3) + (c.attached, len(resets))
        c = Ctx(["rccl"] if rank == 0 else [])             # RCCL refuses on one rank: peer is the only candidate left
        out["peer_only"] = _dist.attach_fastest(c, dist, lambda: None, iters=2)[0], c.attached
        c = Ctx(["export", "rccl"])                        # nothing attaches: host transport
        out["none"] = _dist.attach_fastest(c, dist, lambda: None, iters=2)[0], c.calls[-1]
        # a transport that attaches everywhere but cannot exchange on ONE rank -- in the warm-up, or only in the timed
        # run: the healthy rank and the failed one keep walking through the same collectives, the transport (and its
        # two-panel form) is dropped by both, and the next one is timed and kept
        for where, who in (("run_warmup", 1), ("run_timed", 0)):
            c = Ctx([where] if rank == who else [])
            best, timings = _dist.attach_fastest(c, dist, lambda: None, candidates=("peer", "peer2", "rccl"), iters=3)
            out[where] = best, sorted(timings), c.attached, c.calls.count("peer")
        q.put((rank, out))
    finally:
        dist.close()


@pytest.mark.parametrize("kind", ["gloo", "stdlib"])
def test_attach_fastest_picks_by_the_slowest_rank(kind):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fastest, args=(r, 2, port, q, kind)) for r in range(2)]
    for p in procs:
        p.start()
    res = [o for _, o in sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for r in res:
        best, timings, attached, n_reset = r["slow_peer"]
        assert best in ("rccl", "rccl2") and attached == "rccl" and set(timings) == {"peer", "rccl", "rccl2"}
        assert timings["peer"] > max(timings["rccl"], timings["rccl2"])
        assert n_reset == 4                                  # before each trial and after the final attach
        assert r["peer_only"] == ("peer", "peer")
        assert r["none"] == ("host", "host")
        for where in ("run_warmup", "run_timed"):
            assert r[where] == ("rccl", ["rccl"], "rccl", 1)     # peer attached once, was dropped, peer2 never tried
    assert res[0]["slow_peer"][1] == res[1]["slow_peer"][1]  # the max over ranks is what every rank sees


def _worker_env(rank, world, port, q, tcp):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if tcp:
        os.environ["NBMF_RDZV_PORT"] = str(port)
        os.environ["NBMF_RDZV_SECRET"] = "job-%d" % port       # TCP refuses to run without one
    else:
        os.environ.pop("NBMF_RDZV_PORT", None)
        os.environ.pop("NBMF_RDZV_SECRET", None)               # the Unix socket checks the peer's user id instead
    from nbmf_mm_amd import _rendezvous
    with _rendezvous.init_from_env(timeout=60) as g:
        out = {"who": (g.rank, g.world)}
        out["gather"] = g.all_gather({"rank": rank, "blob": bytes([rank]) * 128})
        out["bcast"] = g.broadcast(np.arange(5) if rank == 1 else None, src=1)
        big = np.full(1 << 20, float(rank + 1))                 # an 8 MiB payload (the host transport's size at c3)
        g.all_reduce(big, "sum")
        out["sum"] = (float(big[0]), float(big[-1]))
        small = np.array([0.1 * (rank + 1), 1.0 / 3.0])
        g.all_reduce(small, "sum")
        out["bits"] = small.tobytes()                           # formed in rank order on every rank: same bits
        out["agree"] = (g.agree(True), g.agree(rank == 0))
        out["max"] = g.max_float(1.5 + rank)
        g.barrier()
    q.put((rank, out))


@pytest.mark.parametrize("tcp,world", [(False, 3), (True, 3), (False, 8)])
def test_stdlib_rendezvous_from_env(tcp, world):
    """The product's own group as a launcher's environment describes it: abstract Unix socket named after
    MASTER_PORT (which the launcher itself may be listening on), or TCP with NBMF_RDZV_PORT."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    holder = None
    if not tcp:                      # somebody else (torch.distributed.run's store) owns the TCP port itself
        holder = socket.socket()
        holder.bind(("127.0.0.1", port))
        holder.listen(1)
    procs = [ctx.Process(target=_worker_env, args=(r, world, port, q, tcp)) for r in range(world)]
    for p in procs:
        p.start()
    res = [o for _, o in sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    if holder:
        holder.close()
    for r, o in enumerate(res):
        assert o["who"] == (r, world)
        assert [d["rank"] for d in o["gather"]] == list(range(world)) and o["gather"][2]["blob"] == b"\x02" * 128
        np.testing.assert_array_equal(o["bcast"], np.arange(5))
        total = world * (world + 1) / 2.0
        assert o["sum"] == (total, total)
        assert o["agree"] == (True, False) and o["max"] == 0.5 + world
        assert o["bits"] == res[0]["bits"]


def test_single_group_is_the_identity():
    from nbmf_mm_amd import _rendezvous
    os.environ.pop("WORLD_SIZE", None)
    g = _rendezvous.init_from_env()
    assert isinstance(g, _rendezvous.SingleGroup) and (g.rank, g.world) == (0, 1)
    a = np.array([1.0, 2.0])
    assert g.all_reduce(a) is a and g.all_gather(7) == [7] and g.broadcast("x") == "x" and g.agree(True) and g.max_float(2) == 2.0


def test_rendezvous_wire_format_round_trip_and_refusals():
    """Messages are a tagged binary encoding, not pickle: what the product sends survives, anything else is refused,
    and a corrupted frame is an error rather than an action."""
    from nbmf_mm_amd import _rendezvous as rz
    msg = ("msg", {"rows": [{"K": 8, "alpha": 1.1, "loss": float("inf"), "name": "val"}], "w": np.arange(12.0).reshape(3, 4),
                   "mask": np.array([True, False]), "h": b"\x00\x01" * 64, "big": 1 << 70, "neg": -5, "none": None, "t": (1, (2.5, "x")),
                   "empty": np.zeros((0, 3), dtype=np.int32), "flag": np.bool_(True), "i32": np.int32(7), "f32": np.float32(0.5)})
    back = rz.decode(rz.encode(msg))
    assert back[0] == "msg" and back[1]["rows"] == msg[1]["rows"] and back[1]["big"] == 1 << 70 and back[1]["t"] == (1, (2.5, "x"))
    np.testing.assert_array_equal(back[1]["w"], msg[1]["w"])
    assert back[1]["w"].dtype == np.float64 and back[1]["mask"].dtype == np.bool_ and back[1]["empty"].shape == (0, 3)
    assert back[1]["flag"] is True and back[1]["i32"] == 7 and back[1]["f32"] == 0.5 and back[1]["h"] == msg[1]["h"]
    with pytest.raises(TypeError):
        rz.encode(object())
    with pytest.raises(TypeError):
        rz.encode(np.array([object()], dtype=object))
    data = rz.encode([1.0, "abc", np.arange(4)])
    for bad in (data[:-3], data + b"N", b"l" + (1 << 60).to_bytes(8, "little") + b"N", b"Z", b"c" + data):
        with pytest.raises((ValueError, struct_error)):
            rz.decode(bad)


def test_tcp_rendezvous_needs_a_secret():
    from nbmf_mm_amd import _rendezvous as rz
    old = os.environ.pop("NBMF_RDZV_SECRET", None)
    try:
        with pytest.raises(ValueError, match="secret"):
            rz.Group(0, 2, ("tcp", "127.0.0.1", _free_port()))
    finally:
        if old is not None:
            os.environ["NBMF_RDZV_SECRET"] = old


def _worker_auth(rank, world, port, q, delay):
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from nbmf_mm_amd import _rendezvous as rz
    time.sleep(delay if rank == 1 else 0.0)      # a late rank: the others wait inside the collective, however long
    with rz.Group(rank, world, ("tcp", "127.0.0.1", port), timeout=60, secret=b"right") as g:
        q.put((rank, g.all_gather(rank)))


def test_strangers_cannot_join_and_late_ranks_can():
    """Before the ranks arrive, somebody else talks to rank 0's relay: garbage, a well-formed handshake with the
    wrong secret, a connection that says nothing.  None of it is parsed as a message, none of it takes a rank's
    place, and the job completes -- with one rank arriving seconds late."""
    import multiprocessing as mp
    from nbmf_mm_amd import _rendezvous as rz
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_worker_auth, args=(r, world, port, q, 3.0)) for r in range(world)]
    procs[0].start()
    strangers = []
    deadline = time.time() + 30
    while True:                                   # wait for the listener
        try:
            s = socket.create_connection(("127.0.0.1", port), timeout=1)
            break
        except OSError:
            assert time.time() < deadline
            time.sleep(0.05)
    s.sendall(b"\x80\x04\x95" + os.urandom(64))   # something pickle-shaped
    strangers.append(s)
    s = socket.create_connection(("127.0.0.1", port), timeout=5)
    s.settimeout(5)
    try:
        rz._handshake_client(s, b"wrong")         # fails on the relay's side (our tag does not verify) or on ours
        rz._send(s, ("hello", 1, world))
    except (ConnectionError, OSError):
        pass
    strangers.append(s)
    strangers.append(socket.create_connection(("127.0.0.1", port), timeout=5))   # silent
    for p in procs[1:]:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for s in strangers:
        s.close()
    assert all(res[r] == [0, 1, 2] for r in range(world))


def test_client_refuses_a_relay_that_does_not_know_the_secret():
    """The other direction: somebody bound the address first and plays relay."""
    import threading
    from nbmf_mm_amd import _rendezvous as rz
    port = _free_port()
    srv = socket.socket()
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind(("127.0.0.1", port))
    srv.listen(1)

    def fake():
        c, _ = srv.accept()
        try:
            rz._handshake_server(c, b"not the job's secret")
            rz._send(c, ("welcome", 2))
        except (ConnectionError, OSError):
            pass
        finally:
            c.close()
    t = threading.Thread(target=fake, daemon=True)
    t.start()
    with pytest.raises(ConnectionError, match="authentication"):
        rz.Group(1, 2, ("tcp", "127.0.0.1", port), timeout=10, secret=b"right")
    t.join(5)
    srv.close()


def test_several_ranks_in_one_process_and_private_init_stream():
    """A process may host several ranks, one thread each (bench.py --ranks-per-process; the peer transport addresses
    same-process arenas directly): the group works between threads of one process, and the reference's init rule is
    drawn from a PRIVATE legacy generator that yields the numbers of np.random.seed(s) + global draws, so concurrent
    ranks do not race on the global stream (and do not disturb it)."""
    import threading
    from nbmf_mm_amd import _dist, _rendezvous
    port = _free_port()
    old = {k: os.environ.get(k) for k in ("WORLD_SIZE", "RANK", "MASTER_ADDR", "MASTER_PORT", "NBMF_RDZV_PORT", "NBMF_RDZV_SECRET")}
    os.environ.update(WORLD_SIZE="4", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("NBMF_RDZV_PORT", None)
    out, errors = {}, []

    def body(rank):
        try:
            with _rendezvous.init_from_env(timeout=60, rank=rank, world=4) as g:
                W, H = _dist.global_init(50, 30, 4, random_state=7)
                out[rank] = (g.all_gather(rank), g.max_float(float(rank)), W, H)
        except BaseException as e:      # noqa: BLE001
            errors.append(repr(e))
    try:
        np.random.seed(123)
        before = np.random.get_state()[1].copy()
        threads = [threading.Thread(target=body, args=(r,)) for r in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(120)
        assert not errors, errors
        np.testing.assert_array_equal(np.random.get_state()[1], before)      # the global stream was not touched
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    np.random.seed(7)                                                       # the reference's rule, _solver.py:102-136
    W0 = np.random.uniform(0.1, 0.9, (50, 4))
    H0 = np.random.uniform(0.1, 0.9, (4, 30))
    for r in range(4):
        assert out[r][0] == [0, 1, 2, 3] and out[r][1] == 3.0
        np.testing.assert_array_equal(out[r][2], W0.T / W0.T.sum(axis=0, keepdims=True))
        np.testing.assert_array_equal(out[r][3], H0)


def _oracle_rank_fit(V_local, global_shape, offset, K, group, orientation="beta-dir", shard="rows", max_iter=500, tol=1e-5,
                     alpha=1.2, beta=1.2, W_init=None, H_init=None, mask_local=None, random_state=None, eps=1e-8,
                     projection="normalize", device=0, transport="auto", progress=None):
    """What ``_dist.fit_sharded`` does for ONE rank of ``fit_in_process``, with the sharded oracle in place of the GPU:
    same arguments, same return value ``(W (rows_here, k), H (k, cols_here), losses, n_iter)``."""
    from nbmf_mm_amd import _dist
    from oracle import sharded_oracle
    M, N = global_shape
    transposed = orientation == "dir-beta"
    m_int, n_int = (N, M) if transposed else (M, N)
    if transposed and W_init is not None and H_init is not None:
        W_init, H_init = np.asarray(H_init).T, np.asarray(W_init).T
    Wi, Hi = _dist.global_init(m_int, n_int, K, random_state, W_init, H_init)
    V_local = np.asarray(V_local, dtype=np.float64)
    mk = None if mask_local is None else np.asarray(mask_local, dtype=np.float64)
    sl = slice(offset, offset + V_local.shape[0])

    def allreduce(arr):
        group.all_reduce(arr, "sum")

    nobs = np.array([float(V_local.size if mk is None else np.count_nonzero(mk))])
    allreduce(nobs)
    def reported(losses):   # (the GPU context reports every ten iterations while it runs; the oracle has only the end)
        if progress is not None:
            for first in range(0, len(losses), 10):
                progress(first, losses[first:first + 10])
        return losses

    if not transposed:
        Wl, H, losses = sharded_oracle.sharded_solve(V_local, mk, Wi[:, sl], Hi, alpha, beta, nobs[0], allreduce, max_iter, tol, eps)
        return Wl.T, H, reported(losses), len(losses)
    W, Hl, losses = sharded_oracle.sharded_solve_cols(V_local.T, None if mk is None else mk.T, Wi, Hi[:, sl], alpha, beta, nobs[0],
                                                      n_int, allreduce, max_iter, tol, eps)
    return Hl.T, W, reported(losses), len(losses)        # un-transposed, _solver.py:178-184


@pytest.mark.parametrize("orientation", ["beta-dir", "dir-beta"])
@pytest.mark.parametrize("n_gpus", [2, 5])
def test_fit_in_process_splits_seeds_and_assembles_like_the_single_fit(orientation, n_gpus):
    """``nbmf_mm_solver(..., n_gpus=N)`` on the CPU: the entry's own logic -- ONE seeding and draw of the global generator
    in the reference's order, row views of V and of the mask per rank, the ranks as threads over ``LocalGroup``, the split
    factor's slices put back together, the stop rule -- with the sharded ORACLE standing in for the GPU ranks, against the
    unsharded oracle (= the reference, tests/test_oracle_golden.py)."""
    from nbmf_mm_amd import _dist
    from oracle import nbmf_oracle as orc
    M, N, K = 203, 96, 5
    g = np.random.default_rng(17)
    V = (g.random((M, N)) < 0.3).astype(np.float64)
    mask = (g.random((M, N)) < 0.85)
    for kw in (dict(max_iter=20, tol=0.0), dict(max_iter=200, tol=3e-4)):
        W, H, losses, t, n_iter = _dist.fit_in_process(V, K, n_gpus, devices=[0] * n_gpus, orientation=orientation, alpha=1.2,
                                                       beta=1.3, mask=mask, random_state=4, _rank_fit=_oracle_rank_fit, **kw)
        Wr, Hr, lr, _, nr = orc.solve(V, K, alpha=1.2, beta=1.3, mask=mask.astype(np.float64), random_state=4,
                                      orientation=orientation, **kw)
        assert t == 0.0 and n_iter == nr == len(losses) and (kw["tol"] == 0.0 or n_iter < kw["max_iter"])
        np.testing.assert_allclose(losses, lr, rtol=1e-12, atol=0)
        np.testing.assert_allclose(W, Wr, rtol=0, atol=1e-12)
        np.testing.assert_allclose(H, Hr, rtol=0, atol=1e-12)
    # the global generator was seeded and drawn from exactly as the single fit does: what comes next is the same number
    nxt = np.random.uniform()
    orc.solve(V, K, max_iter=1, tol=0, random_state=4, orientation=orientation)
    assert np.random.uniform() == nxt
    # custom inits (both given: swapped under dir-beta, _solver.py:122-123) and a rank that fails: the error comes back,
    # nobody hangs
    W0, H0 = g.uniform(0.1, 0.9, (M, K)), g.uniform(0.1, 0.9, (K, N))
    W, H, losses, _, _ = _dist.fit_in_process(V, K, n_gpus, devices=[0] * n_gpus, orientation=orientation, max_iter=6, tol=0,
                                              W_init=W0, H_init=H0, _rank_fit=_oracle_rank_fit)
    Wr, Hr, lr, _, _ = orc.solve(V, K, max_iter=6, tol=0, W_init=W0, H_init=H0, orientation=orientation)
    np.testing.assert_allclose(losses, lr, rtol=1e-12, atol=0)
    np.testing.assert_allclose(W, Wr, rtol=0, atol=1e-12)

    def failing(V_local, global_shape, offset, K, group, **kw):
        if group.rank == n_gpus - 1:
            raise RuntimeError("rank down")
        return _oracle_rank_fit(V_local, global_shape, offset, K, group, **kw)
    with pytest.raises(RuntimeError, match="rank down"):
        _dist.fit_in_process(V, K, n_gpus, devices=[0] * n_gpus, orientation=orientation, max_iter=5, tol=0, _rank_fit=failing)
    with pytest.raises(ValueError, match="devices names"):
        _dist.fit_in_process(V, K, n_gpus, devices=[0], _rank_fit=_oracle_rank_fit)
    # fewer rows than ranks: refused in the caller's thread, before any rank starts (found by tests/manual/fuzz_sharded_vs_single.py:
    # the rank threads died one by one and the caller got a TypeError out of their missing results)
    with pytest.raises(ValueError, match=f"cannot shard {n_gpus - 1} rows over {n_gpus} ranks"):
        _dist.fit_in_process(V[:n_gpus - 1], K, n_gpus, devices=[0] * n_gpus, _rank_fit=_oracle_rank_fit)


def test_fit_in_process_verbose_reports_through_rank_zero_only(capsys):
    """``verbose > 0`` under ``n_gpus > 1``: the reference's lines (src/nbmf_mm/_solver.py:165-166,172-173) come from ONE
    rank's progress callback -- every tenth iteration once, in order, with the losses of the global matrix -- and
    "Converged at iteration" follows the single-GPU solver's rule."""
    from nbmf_mm_amd import _dist
    from oracle import nbmf_oracle as orc
    g = np.random.default_rng(8)
    V = (g.random((90, 40)) < 0.3).astype(np.float64)
    W, H, losses, _, n_iter = _dist.fit_in_process(V, 5, 3, devices=[0] * 3, max_iter=25, tol=0, random_state=2, verbose=1,
                                                   _rank_fit=_oracle_rank_fit)
    out = capsys.readouterr().out.splitlines()
    _, _, lr, _, _ = orc.solve(V, 5, max_iter=25, tol=0, random_state=2)
    assert out == [f"Iter {it:4d}: Loss = {lr[it]:.6f}" for it in (0, 10, 20)]
    _dist.fit_in_process(V, 5, 3, devices=[0] * 3, max_iter=400, tol=1e-3, random_state=2, verbose=1, _rank_fit=_oracle_rank_fit)
    out = capsys.readouterr().out.splitlines()
    assert out[-1].startswith("Converged at iteration ") and all(ln.startswith("Iter ") for ln in out[:-1])


def test_local_group_collectives():
    import threading
    from nbmf_mm_amd import _rendezvous
    groups = _rendezvous.LocalGroup.make(4)
    out = [None] * 4

    def body(r):
        g = groups[r]
        a = np.full(3, float(r + 1))
        g.all_reduce(a, "sum")
        out[r] = (g.all_gather(r * 10), g.broadcast("x" if r == 2 else None, src=2), g.agree(r != 1), g.agree(True),
                  g.max_float(r * 0.5), a.tolist())
        g.barrier()
    ts = [threading.Thread(target=body, args=(r,)) for r in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    assert all(o == ([0, 10, 20, 30], "x", False, True, 1.5, [10.0, 10.0, 10.0]) for o in out)


def test_local_group_names_the_rank_that_never_arrives():
    """A rank stuck in a device call never reaches the next collective: the others give up after the group's timeout with
    a ConnectionError that says which rank was missing (the default timeout is finite: NBMF_LOCAL_GROUP_TIMEOUT_S or 1800 s)."""
    import threading
    from nbmf_mm_amd import _rendezvous
    assert _rendezvous.LocalGroup.make(2)[0]._s.timeout == _rendezvous.LocalGroup.DEFAULT_TIMEOUT_S
    groups = _rendezvous.LocalGroup.make(3, timeout=0.5)
    errs, gate = [None] * 3, threading.Event()

    def body(r):
        try:
            groups[r].barrier()                      # everybody: fine
            if r == 2:
                gate.wait(10)                        # "stuck"
                return
            groups[r].all_gather(r)
        except ConnectionError as e:
            errs[r] = str(e)
    ts = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(3)]
    for t in ts:
        t.start()
    for t in ts[:2]:
        t.join(20)
    gate.set()
    assert errs[0] and errs[1] and all("rank(s) [2]" in e and "0.5 s" in e for e in errs[:2]), errs


def test_fit_in_process_interrupted_caller_ends_every_rank():
    """Ctrl-C while the caller joins the rank threads: the ranks' group is aborted, the threads end within the bound and
    the interrupt goes on to the caller -- nothing keeps sweeping behind its back (with real contexts they are cancelled
    as well: tests/test_gpu_lifecycle.py)."""
    import _thread
    import threading
    import time
    from nbmf_mm_amd import _dist
    V = (np.random.default_rng(3).random((40, 30)) < 0.3).astype(np.float64)
    started = threading.Barrier(4)

    def endless(V_local, global_shape, offset, K, group, **kw):
        started.wait(10)
        while True:                                   # "sweeping": a collective per iteration, for ever
            group.barrier()
            time.sleep(0.005)

    def interrupt():
        started.wait(10)
        time.sleep(0.3)
        _thread.interrupt_main()
    threading.Thread(target=interrupt, daemon=True).start()
    t0 = time.monotonic()
    with pytest.raises(KeyboardInterrupt):
        _dist.fit_in_process(V, 3, 3, devices=[0] * 3, max_iter=5, tol=0, _rank_fit=endless)
    assert time.monotonic() - t0 < 10
    assert not [t for t in threading.enumerate() if t.name.startswith("nbmf-rank-")]


def test_local_group_late_rank_gets_the_named_diagnosis_too():
    """The rank everybody waited for arrives AFTER the barrier broke: it reports the same missing ranks and bound as the
    ranks that timed out, and fit_in_process shows that error rather than a generic one."""
    import threading
    from nbmf_mm_amd import _dist, _rendezvous
    groups = _rendezvous.LocalGroup.make(3, timeout=0.4)
    errs, gate = [None] * 3, threading.Event()

    def body(r):
        try:
            if r == 0:
                gate.wait(10)                        # the late one (rank 0: the error fit_in_process used to show)
            groups[r].barrier()
        except ConnectionError as e:
            errs[r] = str(e)
            if r != 0:
                gate.set()
    ts = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(20)
    assert all(e and "rank(s) [0]" in e and "0.4 s" in e for e in errs), errs
    # through fit_in_process: rank 0 is late (stuck), the others time out; whichever error is raised names rank 0
    V = (np.random.default_rng(3).random((40, 30)) < 0.3).astype(np.float64)
    import os
    os.environ["NBMF_LOCAL_GROUP_TIMEOUT_S"] = "0.4"
    try:
        def late(V_local, global_shape, offset, K, group, **kw):
            if group.rank == 0:
                import time
                time.sleep(1.0)
            group.barrier()
            raise AssertionError("unreachable")
        with pytest.raises(ConnectionError, match=r"rank\(s\) \[0\]"):
            _dist.fit_in_process(V, 3, 3, devices=[0] * 3, max_iter=5, tol=0, _rank_fit=late)
    finally:
        del os.environ["NBMF_LOCAL_GROUP_TIMEOUT_S"]
