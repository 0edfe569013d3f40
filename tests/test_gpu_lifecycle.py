"""What a context takes from the device and the host it gives back (GPU).

The reference's fit holds NumPy arrays that the interpreter frees (`/root/reference/src/nbmf_mm/_solver.py:64-216`: everything
is a local of `nbmf_mm_solver`); the drop-in holds device images, slabs, workspaces, pinned staging buffers, events and
streams behind an opaque handle (`include/nbmf_hip.h`: `nbmf_create` / `nbmf_destroy`).  A grid search creates and destroys
hundreds of them in one process (`nbmf_mm_amd/experiments.py`), so what `nbmf_destroy` leaves behind decides whether such
a process survives: these tests count it with the runtime's own `hipMemGetInfo`.
"""
import ctypes
import gc
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from nbmf_mm_amd import _hip
    assert _hip.device_count() >= 1, "no MI355X visible: the GPU tests must run on the GPU box"
    return _hip


def _free_bytes(hip):
    """Free device memory.  From the HIP runtime the library itself is using -- the copy of libamdhip64 this process has
    mapped (PyTorch, which tests/conftest.py imports first, bundles one of its own: opening "libamdhip64.so" by name could
    start a second runtime) -- or, if two copies are mapped, from the driver's count of the card's VRAM in sysfs."""
    hip.load()
    hip.device_synchronize(0)
    with open("/proc/self/maps") as f:
        mapped = sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    if len(mapped) == 1:
        rt = ctypes.CDLL(mapped[0])                      # the same path: the handle that is loaded already
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert rt.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value
    import glob
    used = [int(open(p).read()) for p in glob.glob("/sys/class/drm/card*/device/mem_info_vram_used")]
    if not used:
        pytest.skip(f"no way to count device memory here (runtimes mapped: {mapped})")
    return -sum(used)                                    # (only differences are looked at)


def _rss_bytes():
    with open(f"/proc/{os.getpid()}/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


def _one_fit(hip, V, mask, k, iters, storage="auto"):
    m, n = V.shape
    g = np.random.default_rng(3)
    W0 = g.random((k, m))
    W0 /= W0.sum(axis=0, keepdims=True)
    H0 = g.uniform(0.1, 0.9, (k, n))
    with hip.Context(m, n, k) as ctx:
        ctx.set_hyper(1.2, 1.2, 1e-8)
        ctx.set_storage(storage)
        ctx.upload(V, mask)
        ctx.set_factors(W0, H0)
        losses, n_iter = ctx.run(iters, 0.0)
        W, H = ctx.get_factors()
        assert n_iter == iters and np.isfinite(losses).all() and np.isfinite(W).all() and np.isfinite(H).all()
    return losses[-1]


def test_the_probe_sees_a_live_context(hip):
    """(The measurement itself: a live 4096 x 4096 context on the 8-byte path holds two 128 MiB images, and the probe
    says so -- the tests below would see a leak of one.  Blocks of up to 64 MiB stay with the library's pool when a
    context goes -- nbmf_hip.hip, `dmalloc` / `dfree`: up to NBMF_POOL_MB = 1 GiB, handed to the next context that asks
    for the size -- so every test here takes its base line after one round of what it repeats.)"""
    g = np.random.default_rng(5)
    V = g.random((4096, 4096))
    with hip.Context(4096, 4096, 16) as ctx:
        ctx.upload(V, None)
    gc.collect()
    base = _free_bytes(hip)
    with hip.Context(4096, 4096, 16) as ctx:
        ctx.upload(V, None)
        held = base - _free_bytes(hip)
    assert held >= 256 << 20, f"a live context shows as {held / 2**20:.0f} MiB"
    assert abs(base - _free_bytes(hip)) <= 16 << 20


def test_destroyed_contexts_give_the_device_memory_back(hip):
    """Forty-five create / upload / fit / destroy rounds over every storage path and both engines (small problems run in the
    single launch, the 1536-row one in the five kernels): the device's free memory ends where it stood after the first
    round (whose pools -- streams, the logarithm table, the runtime's own -- stay), to 16 MiB."""
    g = np.random.default_rng(0)
    shapes = [(96, 200, 5), (1536, 1024, 32), (300, 260, 16)]
    cases = []
    for (m, n, k) in shapes:
        Vb = (g.random((m, n)) < 0.3).astype(np.float64)
        Vr = g.random((m, n))
        Mb = g.random((m, n)) < 0.9
        Mw = g.random((m, n))
        cases += [(Vb, None, k, "auto"), (Vb, Mb, k, "auto"), (Vb, Mb, k, "f64"), (Vr, Mb, k, "auto"), (Vr, Mw, k, "auto")]
    first = [_one_fit(hip, V, M, k, 6, st) for (V, M, k, st) in cases]      # warm: pools and lazily built tables exist now
    gc.collect()
    base = _free_bytes(hip)
    rss0 = _rss_bytes()
    for rep in range(3):
        again = [_one_fit(hip, V, M, k, 6, st) for (V, M, k, st) in cases]
        assert again == first                                              # and the fits are the same bits every time
    gc.collect()
    after = _free_bytes(hip)
    assert abs(base - after) <= 16 << 20, f"device memory moved by {(base - after) / 2**20:.1f} MiB over 45 contexts"
    assert _rss_bytes() - rss0 <= 256 << 20, f"host RSS grew by {(_rss_bytes() - rss0) / 2**20:.0f} MiB"


def test_a_context_that_fails_midway_gives_everything_back(hip):
    """Errors between create and destroy -- a mask of the wrong shape, factors of the wrong shape, a run without data --
    leave a context that can still be destroyed, and the memory comes back."""
    g = np.random.default_rng(1)
    V = (g.random((640, 512)) < 0.3).astype(np.float64)
    _one_fit(hip, V, None, 16, 3)
    gc.collect()
    base = _free_bytes(hip)
    for rep in range(10):
        ctx = hip.Context(640, 512, 16)
        with pytest.raises(hip.NBMFHipError):
            ctx.run(3, 0.0)                                               # nothing uploaded
        ctx.upload(V, None)
        with pytest.raises((hip.NBMFHipError, ValueError)):
            ctx.upload(V, np.ones((5, 5)))                                # mask of another shape
        with pytest.raises((hip.NBMFHipError, ValueError)):
            ctx.set_factors(np.ones((3, 640)), np.ones((3, 512)))         # another k
        ctx.close()
        ctx.close()                                                       # closing twice is harmless
    gc.collect()
    after = _free_bytes(hip)
    assert abs(base - after) <= 16 << 20, f"device memory moved by {(base - after) / 2**20:.1f} MiB"


def test_estimators_dropped_without_ceremony(hip):
    """The scikit-learn-style estimator owns no device state between calls: twenty fits + transforms of throw-away
    estimators leave the device where it was."""
    from nbmf_mm_amd import NBMF
    g = np.random.default_rng(2)
    X = (g.random((400, 300)) < 0.3).astype(np.float64)
    NBMF(n_components=8, max_iter=20, random_state=0).fit(X)
    gc.collect()
    base = _free_bytes(hip)
    for rep in range(20):
        est = NBMF(n_components=8, max_iter=20, random_state=rep, orientation="beta-dir" if rep % 2 else "dir-beta")
        Wt = est.fit_transform(X)
        est.transform(X[:50])
        assert np.isfinite(Wt).all() and np.isfinite(est.score(X))
        del est
    gc.collect()
    after = _free_bytes(hip)
    assert abs(base - after) <= 16 << 20, f"device memory moved by {(base - after) / 2**20:.1f} MiB over 20 estimators"
    # ... and so a fitted estimator is plain data: it survives pickling and sklearn's clone protocol (joblib ships estimators
    # to workers this way), and the copy evaluates as the original does
    import pickle
    from sklearn.base import clone
    est = NBMF(n_components=8, max_iter=20, random_state=3).fit(X)
    copy = pickle.loads(pickle.dumps(est))
    np.testing.assert_array_equal(copy.components_, est.components_)
    np.random.seed(5)
    a = est.transform(X[:40])
    np.random.seed(5)
    b = copy.transform(X[:40])
    np.testing.assert_array_equal(a, b)
    fresh = clone(est)
    assert not hasattr(fresh, "components_") and fresh.get_params() == est.get_params()


_SHARDED_SCRIPT = r"""
import ctypes, gc, glob, json, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from nbmf_mm_amd import NBMF, _hip

def free_bytes():
    _hip.load()
    _hip.device_synchronize(0)
    mapped = sorted({line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line})
    if len(mapped) == 1:
        rt = ctypes.CDLL(mapped[0])
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert rt.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value
    return -sum(int(open(p).read()) for p in glob.glob("/sys/class/drm/card*/device/mem_info_vram_used"))

g = np.random.default_rng(4)
X = (g.random((2100, 640)) < 0.3).astype(np.float64)
M = g.random((2100, 640)) < 0.9
def fit(rep):
    est = NBMF(n_components=16, max_iter=8, tol=0.0, random_state=7, n_gpus=4, devices=[0] * 4)
    est.fit(X, mask=M)
    return float(est.objective_history_[-1]) if hasattr(est, "objective_history_") else float(est.reconstruction_err_)
first = fit(0)
gc.collect()
base = free_bytes()
same = all(fit(r) == first for r in range(1, 7))
gc.collect()
print("RESULT " + json.dumps({"moved": base - free_bytes(), "same": same}))
"""


def test_sharded_fits_give_their_arenas_back_or_reuse_them():
    """Six more four-rank fits in one process (rank threads on device 0, peer transport: exchange arenas, flags, a
    context and stream per rank) after a first one: the device's free memory does not move (the arenas are pooled per
    device and reused, everything else is freed with its context) and every fit has the first one's bits."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="32", NBMF_PEER_TIMEOUT_MS="20000")
    r = subprocess.run([sys.executable, "-c", _SHARDED_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert out["same"] is True
    assert abs(out["moved"]) <= 16 << 20, f"device memory moved by {out['moved'] / 2**20:.1f} MiB over 6 sharded fits"


def test_a_problem_that_does_not_fit_is_an_error_and_costs_nothing(hip):
    """4 194 304 x 49 152 binary entries generated on the device would need ~410 GB of tile codes: the upload fails with
    the library's error (no crash, no abort), the context can be destroyed, what it had allocated before the failure
    comes back, and the next fit on the device is the usual one."""
    g = np.random.default_rng(6)
    V = (g.random((512, 384)) < 0.3).astype(np.float64)
    want = _one_fit(hip, V, None, 16, 4)
    gc.collect()
    base = _free_bytes(hip)
    ctx = hip.Context(4194304, 49152, 16)
    try:
        with pytest.raises(hip.NBMFHipError, match="(?i)memory|alloc"):
            ctx.generate(seed=1, density=0.25, observed=0.9)
    finally:
        ctx.close()
    gc.collect()
    # (the allocator answers "out of memory" by giving the pool's idle blocks back to the runtime before it tries again:
    #  free memory may well have GROWN; what must not happen is that the failed context's allocations stay)
    assert base - _free_bytes(hip) <= 256 << 20
    assert _one_fit(hip, V, None, 16, 4) == want


def test_a_context_used_again_is_a_fresh_one(hip):
    """One context, six uploads in a row -- byte codes, doubles with a folded mask, doubles with weight tiles, byte codes
    with a mask, the first again, hyper-parameters and projection changed in between, evaluation calls in between: every
    fit has the bits of the same fit in a context of its own (nothing of an earlier upload -- storage kind, lane masks,
    row counts, the factors' range check, stop flags, the loss carried into the next sweep -- survives the next one).
    Both engines: the 200 x 160 problems run in the single launch, the 1408 x 1152 ones in the five kernels."""
    g = np.random.default_rng(8)
    for (m, n, k) in [(200, 160, 6), (1408, 1152, 32)]:
        Vb = (g.random((m, n)) < 0.3).astype(np.float64)
        Vr = g.random((m, n))
        Mb = g.random((m, n)) < 0.85
        Mw = g.random((m, n))
        W0 = g.random((k, m))
        W0 /= W0.sum(axis=0, keepdims=True)
        H0 = g.uniform(0.1, 0.9, (k, n))
        steps = [(Vb, None, 1.2, 1.2, 0), (Vr, Mb, 1.0, 1.5, 1), (Vr, Mw, 2.0, 1.2, 0), (Vb, Mb, 1.2, 1.2, 1),
                 (Vb, None, 1.2, 1.2, 0), (Vr, None, 1.1, 1.1, 0)]

        def fit(ctx, V, M, al, be, proj, iters):
            ctx.set_hyper(al, be, 1e-8, proj)
            ctx.upload(V, M)
            ctx.set_factors(W0, H0)
            losses, n_iter = ctx.run(iters, 1e-7)
            W, H = ctx.get_factors()
            return losses.tobytes(), n_iter, W.tobytes(), H.tobytes(), ctx.loss(), ctx.loglik()

        fresh = []
        for (V, M, al, be, proj) in steps:
            with hip.Context(m, n, k) as ctx:
                fresh.append(fit(ctx, V, M, al, be, proj, 12))
        with hip.Context(m, n, k) as ctx:
            for i, (V, M, al, be, proj) in enumerate(steps):
                got = fit(ctx, V, M, al, be, proj, 12)
                assert got == fresh[i], f"{m} x {n}: upload {i} differs from a context of its own"
                ctx.w_only_steps(3)                                      # an evaluation-time call between two fits
                ctx.loglik_strict()
        assert fresh[0] == fresh[4]


def test_cancel_from_another_thread_ends_a_run(hip):
    """``nbmf_cancel``: a run of 10**6 iterations with tol = 0 is enqueued far ahead of the device; cancelled from another
    thread it comes back within seconds with the library's "cancelled" error (the enqueue loop stops and the run's own stop
    flag cuts short what is queued), the context stays cancelled, closes cleanly, and the next fit on the device is the
    usual one.  (With tol > 0 the loop is left at the next batch boundary; both are covered.)"""
    import threading
    import time
    g = np.random.default_rng(11)
    V = (g.random((2048, 1536)) < 0.3).astype(np.float64)
    want = _one_fit(hip, V, None, 64, 4)
    for tol in (0.0, 1e-300):
        ctx = hip.Context(2048, 1536, 64)
        try:
            ctx.upload(V)
            W0 = g.random((64, 2048))
            ctx.set_factors(W0 / W0.sum(axis=0, keepdims=True), g.uniform(0.1, 0.9, (64, 1536)))
            threading.Timer(0.5, ctx.cancel).start()
            t0 = time.monotonic()
            with pytest.raises(hip.NBMFHipError, match="cancelled"):
                ctx.run(10 ** 6, tol)
            assert time.monotonic() - t0 < 60.0
            with pytest.raises(hip.NBMFHipError, match="cancelled"):
                ctx.run(3, 0.0)
        finally:
            ctx.close()
    assert _one_fit(hip, V, None, 64, 4) == want


_INTERRUPT_SCRIPT = r"""
import json, sys, threading, time, _thread
sys.path.insert(0, sys.argv[1])
import numpy as np
from nbmf_mm_amd import NBMF
g = np.random.default_rng(4)
X = (g.random((4100, 2048)) < 0.3).astype(np.float64)
def interrupt():
    time.sleep(3.0)
    _thread.interrupt_main()
threading.Thread(target=interrupt, daemon=True).start()
t0 = time.monotonic()
try:
    NBMF(n_components=64, max_iter=10 ** 6, tol=0.0, random_state=7, n_gpus=4, devices=[0] * 4).fit(X)
    out = {"interrupted": False}
except KeyboardInterrupt:
    out = {"interrupted": True, "seconds": time.monotonic() - t0,
           "rank_threads_left": [t.name for t in threading.enumerate() if t.name.startswith("nbmf-rank-")]}
# the device is free again: an ordinary fit runs
est = NBMF(n_components=8, max_iter=5, tol=0.0, random_state=1).fit(X[:300, :200])
out["next_fit_ok"] = bool(np.isfinite(est.loss_curve_[-1]))
print("RESULT " + json.dumps(out))
"""


def test_interrupted_caller_of_the_in_process_sharded_fit_stops_the_ranks():
    """Ctrl-C in the thread that called ``NBMF(n_gpus=4).fit`` (a million iterations ahead of it): the four rank threads
    are cancelled, leave their exchanges and close their contexts within the bound; KeyboardInterrupt reaches the caller;
    no rank thread is left; the next fit in the same process runs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="32", NBMF_PEER_TIMEOUT_MS="20000")
    r = subprocess.run([sys.executable, "-c", _INTERRUPT_SCRIPT, root], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert out["interrupted"] is True and out["rank_threads_left"] == [] and out["seconds"] < 60 and out["next_fit_ok"] is True
