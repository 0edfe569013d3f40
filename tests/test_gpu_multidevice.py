"""Tests that select a SECOND PHYSICAL DEVICE -- they run only where at least two MI355X are visible and are skipped
(cleanly, collected) on the one-GPU box.  Everything multi-rank in tests/test_gpu_peer.py and tests/test_gpu_dist.py
stacks its ranks on device 0: exact for the arithmetic, blind to what only distinct devices exercise --
``hipDeviceEnablePeerAccess`` between devices (nbmf_comm_init_peer), HIP IPC handles opened on ANOTHER device, RCCL with
more than one rank, the per-device logarithm table and streams of ``NBMF(n_gpus=N)``.

UNMEASURED until a machine with several GPUs has run them: they are written against the same references
and tolerances as their one-GPU counterparts -- the single-device fit to 1e-12, the replicated factor bitwise equal on
all ranks, the stop rule at the same iteration.  No device is emulated.

(``NBMF_MULTIDEVICE_REHEARSAL=1 GPU_MAX_HW_QUEUES=32 pytest tests/test_gpu_multidevice.py`` runs the same test BODIES with
every rank on device 0 -- a rehearsal of the test code itself on a one-GPU box, which proves nothing about devices and is
never on by default; RCCL, which refuses two ranks on one device, is left out of it.)

Reference loop being sharded: src/nbmf_mm/_solver.py:143-175 (the H-step products of :42-43 are what crosses devices)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ndev():
    try:
        from nbmf_mm_amd import _hip
        return _hip.device_count()
    except Exception:
        return 0


_REHEARSE = os.environ.get("NBMF_MULTIDEVICE_REHEARSAL") == "1"
needs_two = pytest.mark.skipif(_ndev() < 2 and not _REHEARSE,
                               reason="needs at least two visible GPUs (runs when >= 2 devices are visible)")


def _ranks(cap):
    return cap if _REHEARSE else min(cap, _ndev())


def _dev(r):
    return 0 if _REHEARSE else r


@pytest.fixture(autouse=True)
def _five_kernel_reference(monkeypatch):
    """Single-device references on the five-kernel path (the sharded runs are compared with them to 1e-12; the
    single-launch engine adds in another order)."""
    monkeypatch.setenv("NBMF_PERSISTENT", "0")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    g = np.random.default_rng(77)
    M, N, K = 1900, 520, 40
    Y = (g.random((M, N)) < 0.3).astype(np.float64)
    mask = g.random((M, N)) < 0.9
    return M, N, K, Y, mask


@needs_two
@pytest.mark.parametrize("orientation", ["beta-dir", "dir-beta"])
@pytest.mark.parametrize("projection", ["normalize", "duchi"])
def test_n_gpus_on_distinct_devices_equals_one_device(orientation, projection):
    """``NBMF(n_gpus=min(8, ndev))``, one rank per DISTINCT device in one process, against ``n_gpus=1``: loss curve, W_
    and components_ to 1e-12, both orientations, both projections, with the stop rule firing at the same iteration."""
    from nbmf_mm_amd import NBMF
    M, N, K, Y, mask = _problem()
    n = _ranks(8)
    for kw in (dict(max_iter=12, tol=0.0), dict(max_iter=300, tol=2e-4)):
        common = dict(n_components=K, random_state=3, orientation=orientation, alpha=1.2, beta=1.3, projection=projection, **kw)
        one = NBMF(**common).fit(Y, mask=mask)
        many = NBMF(n_gpus=n, devices=[_dev(r) for r in range(n)], **common).fit(Y, mask=mask)
        assert one.n_iter_ == many.n_iter_
        assert (kw["tol"] == 0.0) == (one.n_iter_ == kw["max_iter"])
        np.testing.assert_allclose(many.loss_curve_, one.loss_curve_, rtol=1e-12, atol=0)
        np.testing.assert_allclose(many.W_, one.W_, rtol=0, atol=1e-12)
        np.testing.assert_allclose(many.components_, one.components_, rtol=0, atol=1e-12)


@needs_two
def test_n_gpus_heterogeneous_shards_on_distinct_devices():
    """Real-valued V whose first rows happen to be all 0 / 1: the first device stores byte codes, the others doubles, and
    the ranks' H-sweep products are summed across devices (each rank un-maps its own first: reduce_h_kernel)."""
    from nbmf_mm_amd import NBMF
    g = np.random.default_rng(5)
    n = _ranks(4)
    X = g.random((225 * n, 300))
    X[:225] = X[:225] < 0.3
    mb = g.random(X.shape) < 0.85
    kw = dict(n_components=24, random_state=5, max_iter=10, tol=0.0)
    one = NBMF(**kw).fit(X, mask=mb)
    many = NBMF(n_gpus=n, devices=[_dev(r) for r in range(n)], **kw).fit(X, mask=mb)
    np.testing.assert_allclose(many.loss_curve_, one.loss_curve_, rtol=1e-12, atol=0)
    np.testing.assert_allclose(many.W_, one.W_, rtol=0, atol=1e-12)
    np.testing.assert_allclose(many.components_, one.components_, rtol=0, atol=1e-12)


def _worker(rank, world, port, q, transport, device):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.setdefault("NBMF_PEER_TIMEOUT_MS", "20000")
    from nbmf_mm_amd import _dist, _rendezvous
    dist = _rendezvous.Group(rank, world, ("tcp", "127.0.0.1", port), secret=b"tests-%d" % port)
    try:
        M, N, K, Y, mask = _problem()
        r0, r1 = _dist.shard_bounds(M, world, rank)
        out = {}
        for name, kw in (("tol0", dict(max_iter=20, tol=0)), ("stop", dict(max_iter=400, tol=1e-4))):
            out[name] = _dist.fit_row_sharded(Y[r0:r1], M, r0, K, dist, alpha=1.2, beta=1.3, mask_local=mask[r0:r1],
                                              random_state=5, device=device, transport=transport, **kw)
        q.put((rank, r0, r1, out))
    finally:
        dist.close()


def _run(world, transport):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, transport, _dev(r))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@needs_two
@pytest.mark.parametrize("transport", ["peer", "peer2", "rccl", "host"])
def test_one_process_per_device_every_transport(transport):
    """``fit_row_sharded`` from one process per device (rank r on device r, at most four of them: the box admits few
    GPU processes) over each transport: the single-device fit to 1e-12, the replicated factor and the loss curve bitwise
    identical on all ranks (`replicas_identical`), the stop rule at the single-device iteration."""
    from nbmf_mm_amd import nbmf_mm_solver
    if _REHEARSE and transport == "rccl":
        pytest.skip("RCCL refuses two ranks on one device")
    world = _ranks(4)
    res = _run(world, transport)
    M, N, K, Y, mask = _problem()
    W1, H1, l1, _, _ = nbmf_mm_solver(Y, K, max_iter=20, tol=0, alpha=1.2, beta=1.3, mask=mask, random_state=5)
    W = np.concatenate([r[3]["tol0"][0] for r in res], axis=0)
    np.testing.assert_allclose(W, W1, rtol=0, atol=1e-12)
    np.testing.assert_allclose(res[0][3]["tol0"][1], H1, rtol=0, atol=1e-12)
    np.testing.assert_allclose(res[0][3]["tol0"][2], l1, rtol=1e-12, atol=0)
    for r in res[1:]:                                                    # replicas identical, bit for bit
        np.testing.assert_array_equal(r[3]["tol0"][1], res[0][3]["tol0"][1])
        np.testing.assert_array_equal(r[3]["tol0"][2], res[0][3]["tol0"][2])
    W2, H2, l2, _, n2 = nbmf_mm_solver(Y, K, max_iter=400, tol=1e-4, alpha=1.2, beta=1.3, mask=mask, random_state=5)
    assert 5 < n2 < 400
    for r in res:
        assert r[3]["stop"][3] == n2
        np.testing.assert_allclose(r[3]["stop"][1], H2, rtol=0, atol=1e-11)


@needs_two
def test_peer_attach_with_a_silent_partner_is_an_error_not_a_hang():
    """Rank 0 (device 0) attaches the peer transport to the arena of a rank on device 1 that never joins: peer access
    between the two devices is enabled (hipDeviceEnablePeerAccess), the known-answer exchange reads the partner's arena
    across the link, finds no contribution and gives up within its bound -- an error, with the context left unattached
    and usable."""
    import time
    from nbmf_mm_amd import _hip
    M, N, K, Y, mask = _problem()
    with _hip.Context(M, N, K, device=0) as a, _hip.Context(M, N, K, device=_dev(1)) as b:
        for c in (a, b):
            c.set_hyper(1.2, 1.3)
            c.upload(Y, mask=mask)
        handles = a.peer_export(0) + b.peer_export(0)
        a.set_peer_timeout_ms(1500.0)
        t0 = time.perf_counter()
        with pytest.raises(_hip.NBMFHipError):
            a.comm_init_peer(handles, 2, 0, 0)                          # rank 1 never calls it
        assert time.perf_counter() - t0 < 30.0
        W0 = np.random.default_rng(0).uniform(0.1, 0.9, (K, M))
        W0 /= W0.sum(axis=0, keepdims=True)
        a.set_factors(W0, np.random.default_rng(1).uniform(0.1, 0.9, (K, N)))
        losses, n_it = a.run(3, 0.0)                                     # still usable, unsharded
        assert n_it == 3 and np.all(np.isfinite(losses))
