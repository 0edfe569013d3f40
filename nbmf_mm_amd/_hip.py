"""ctypes binding of libnbmf_hip.so (see include/nbmf_hip.h for the C ABI).

The library is the product path: if it is missing, or there is no MI355X, every entry point
raises — there is no CPU fallback (the CPU oracle under oracle/ is test infrastructure only).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

NBMF_OK = 0
NBMF_ERR_ARG = -1
NBMF_ERR_HIP = -2
NBMF_ERR_RANGE = -3
NBMF_ERR_STATE = -4
NBMF_ERR_COMM = -5

MASK_NONE, MASK_F64, MASK_U8 = 0, 1, 2
PROJ_NORMALIZE, PROJ_DUCHI = 0, 1
FLAG_BINARY_PATH = 1
STORAGE = {"auto": 0, "f64": 1, "f64w": 2}
MAX_K = 512
PEER_HANDLE_BYTES = 192

#: every symbol include/nbmf_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "nbmf_abi_version", "nbmf_last_error", "nbmf_device_count", "nbmf_create", "nbmf_destroy",
    "nbmf_set_hyper", "nbmf_upload", "nbmf_upload_csr", "nbmf_generate", "nbmf_get_n_obs", "nbmf_set_factors", "nbmf_get_factors",
    "nbmf_run", "nbmf_w_only_steps", "nbmf_loss", "nbmf_loglik", "nbmf_loglik_strict", "nbmf_comm_unique_id", "nbmf_comm_init",
    "nbmf_comm_init_host", "nbmf_peer_export", "nbmf_comm_init_peer", "nbmf_comm_detach",
    "nbmf_timing_enable", "nbmf_timing_get", "nbmf_synchronize", "nbmf_selftest_unary",
    "nbmf_set_progress", "nbmf_device_synchronize", "nbmf_small_stats", "nbmf_generate_slice", "nbmf_set_storage",
    "nbmf_set_exchange_panels", "nbmf_set_peer_timeout_ms", "nbmf_run_batch", "nbmf_batch_stats", "nbmf_sweep_info",
    "nbmf_upload_v", "nbmf_selftest_mfma_peak", "nbmf_engine_stats", "nbmf_source_hash", "nbmf_comm_info", "nbmf_cancel", "nbmf_variant_stats", "nbmf_device_bus_id",
]
DATA_F64, DATA_U8, DATA_F32 = 0, 1, 2


#: int fn(void* user, double* buf, int64 count) -- in-place sum over ranks on a host buffer
HOST_ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, POINTER(c_double), c_int64)
#: void fn(void* user, int first, int count, const double* losses) -- nbmf_set_progress
PROGRESS_FN = ctypes.CFUNCTYPE(None, c_void_p, c_int, c_int, POINTER(c_double))


class NBMFHipError(RuntimeError):
    """A libnbmf_hip call failed (HIP error, missing GPU, bad state)."""


def library_path() -> str:
    return os.environ.get("NBMF_HIP_LIBRARY", os.path.join(_HERE, "libnbmf_hip.so"))


_lib = None


def _autobuild(path):
    """Build the library once if the toolchain is here.  Several ranks may arrive together from a fresh
    checkout: the build is serialised by a file lock, written under a temporary name and renamed into place,
    so nobody ever maps a half-written file; a failed build reports the compiler's own message."""
    import fcntl
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)
    if not (shutil.which("make") and hipcc):
        return
    with open(path + ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if os.path.exists(path):            # another rank built it while this one waited
            return
        tmp = f"{path}.tmp.{os.getpid()}"
        r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), f"OUT={tmp}", f"HIPCC={hipcc}"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode == 0 and os.path.exists(tmp):
            os.replace(tmp, path)
        else:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise NBMFHipError(f"building {path} failed:\n{r.stdout[-2000:]}")


# GPU_MAX_HW_QUEUES is read by the HIP runtime at ITS first call, whoever makes it -- another HIP user of the process
# (PyTorch, say) may do so before this library is ever loaded.  The earliest moment this package can observe the variable
# is its own import: that value is recorded, and it is an UPPER bound on what the runtime has only if nothing in the
# process started HIP before with a smaller one (set it before ANY HIP user of the process starts).
_HW_QUEUES_AT_LOAD = os.environ.get("GPU_MAX_HW_QUEUES")


def hw_queues_at_load():
    """GPU_MAX_HW_QUEUES as it stood when this package was imported (no later than the library's load and hence, unless
    another HIP user of the process was there first, than the runtime's start); the runtime's default (4) if unset."""
    try:
        return int(_HW_QUEUES_AT_LOAD) if _HW_QUEUES_AT_LOAD is not None else 4
    except ValueError:
        return 4


def load():
    """Load libnbmf_hip.so (built by nbmf_mm_amd/csrc/Makefile or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path) and "NBMF_HIP_LIBRARY" not in os.environ and not os.environ.get("NBMF_NO_AUTOBUILD"):
        _autobuild(path)                    # a source checkout without the built artefact
    if not os.path.exists(path):
        raise NBMFHipError(
            f"{path} not found: build it with `make -C nbmf_mm_amd/csrc` (needs hipcc); "
            "nbmf_mm_amd has no CPU fallback")
    lib = ctypes.CDLL(path)
    if "NBMF_HIP_LIBRARY" in os.environ:
        # an explicitly named build (A/B measurements against an older library): entry points it lacks raise when called
        class _Missing:
            def __init__(self, name):
                self.name, self.argtypes, self.restype = name, None, None

            def __call__(self, *a):
                raise NBMFHipError(f"{path} does not export {self.name}")
        for name in SYMBOLS:
            if not hasattr(lib, name):
                setattr(lib, name, _Missing(name))
    dp = POINTER(c_double)
    lib.nbmf_abi_version.restype = c_int
    lib.nbmf_source_hash.restype = c_char_p
    lib.nbmf_cancel.argtypes = [c_void_p]
    lib.nbmf_device_bus_id.argtypes = [c_int, c_char_p, c_int]
    lib.nbmf_variant_stats.argtypes = [POINTER(ctypes.c_longlong), POINTER(ctypes.c_longlong), POINTER(ctypes.c_longlong)]
    lib.nbmf_comm_info.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]
    lib.nbmf_last_error.restype = c_char_p
    lib.nbmf_device_count.argtypes = [POINTER(c_int)]
    lib.nbmf_create.argtypes = [c_int64, c_int64, c_int, c_int, POINTER(c_void_p)]
    lib.nbmf_destroy.argtypes = [c_void_p]
    lib.nbmf_set_hyper.argtypes = [c_void_p, c_double, c_double, c_double, c_int]
    lib.nbmf_upload.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, POINTER(c_int)]
    lib.nbmf_upload_v.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_int64, POINTER(c_int)]
    lib.nbmf_selftest_mfma_peak.argtypes = [c_int, c_double, dp, dp, dp]
    lib.nbmf_engine_stats.argtypes = [POINTER(ctypes.c_longlong), POINTER(ctypes.c_longlong), POINTER(ctypes.c_longlong)]
    lib.nbmf_generate.argtypes = [c_void_p, ctypes.c_uint64, c_double, c_double]
    lib.nbmf_generate_slice.argtypes = [c_void_p, ctypes.c_uint64, c_double, c_double, c_int64, c_int64, c_int64]
    lib.nbmf_set_storage.argtypes = [c_void_p, c_int]
    lib.nbmf_get_n_obs.argtypes = [c_void_p, dp]
    lib.nbmf_set_factors.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.nbmf_get_factors.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.nbmf_run.argtypes = [c_void_p, c_int, c_double, c_void_p, POINTER(c_int)]
    lib.nbmf_run_batch.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p,
                                   c_void_p, c_void_p]
    lib.nbmf_batch_stats.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int)]
    lib.nbmf_sweep_info.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]
    lib.nbmf_w_only_steps.argtypes = [c_void_p, c_int]
    lib.nbmf_loss.argtypes = [c_void_p, dp]
    lib.nbmf_loglik.argtypes = [c_void_p, c_int, dp]
    lib.nbmf_loglik_strict.argtypes = [c_void_p, dp]
    lib.nbmf_comm_unique_id.argtypes = [c_void_p]
    lib.nbmf_comm_init.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int]
    lib.nbmf_comm_init_host.argtypes = [c_void_p, HOST_ALLREDUCE_FN, c_void_p, c_int, c_int, c_int]
    lib.nbmf_upload_csr.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, POINTER(c_int)]
    lib.nbmf_peer_export.argtypes = [c_void_p, c_int, c_void_p]
    lib.nbmf_comm_init_peer.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int]
    lib.nbmf_comm_detach.argtypes = [c_void_p]
    lib.nbmf_set_exchange_panels.argtypes = [c_void_p, c_int]
    lib.nbmf_set_peer_timeout_ms.argtypes = [c_void_p, c_double]
    lib.nbmf_timing_enable.argtypes = [c_void_p, c_int]
    lib.nbmf_timing_get.argtypes = [c_void_p, dp, POINTER(c_int), dp, POINTER(c_int)]
    lib.nbmf_synchronize.argtypes = [c_void_p]
    lib.nbmf_set_progress.argtypes = [c_void_p, PROGRESS_FN, c_void_p, c_int]
    lib.nbmf_device_synchronize.argtypes = [c_int]
    lib.nbmf_small_stats.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int)]
    lib.nbmf_selftest_unary.argtypes = [c_int, c_int, c_int, c_void_p, c_void_p]
    for name in SYMBOLS:
        if name not in ("nbmf_last_error", "nbmf_source_hash"):
            getattr(lib, name).restype = c_int
    _lib = lib
    return lib


def _check(rc):
    if rc == NBMF_OK:
        return
    msg = load().nbmf_last_error().decode("utf-8", "replace")
    if rc == NBMF_ERR_RANGE:
        raise ValueError("X must be binary")          # _base.py:91 wording
    if rc == NBMF_ERR_ARG:
        raise ValueError(msg)
    raise NBMFHipError(msg)


def source_hash() -> str:
    """The content hash of the sources the loaded library was compiled from (``nbmf_source_hash``)."""
    try:
        return load().nbmf_source_hash().decode()
    except NBMFHipError:                    # an older build named by NBMF_HIP_LIBRARY
        return "unknown"


def tree_source_hash():
    """The same hash computed from the sources next to this file (csrc/Makefile's recipe; tools/src_hash.sh), or None
    where the package was installed without them."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    files = ([os.path.join(csrc, "nbmf_hip.hip")] + sorted(glob.glob(os.path.join(csrc, "*.inc"))) +
             [os.path.join(_HERE, "..", "include", "nbmf_hip.h")])
    if not all(os.path.exists(f) for f in files):
        return None
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def device_count() -> int:
    n = c_int(0)
    rc = load().nbmf_device_count(byref(n))
    return n.value if rc == NBMF_OK else 0


def device_bus_id(device=0) -> str:
    """PCI bus id of HIP device ``device`` in this process: the physical card behind the index (``nbmf_device_bus_id``)."""
    buf = ctypes.create_string_buffer(64)
    _check(load().nbmf_device_bus_id(int(device), buf, 64))
    return buf.value.decode()


def _f64c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Context:
    """One m x n x k internal problem on one GPU (internal layout: Y m x n, W k x m, H k x n)."""

    def __init__(self, m, n, k, device=0):
        self._lib = load()
        self._h = c_void_p()
        _check(self._lib.nbmf_create(int(m), int(n), int(k), int(device), byref(self._h)))
        self.m, self.n, self.k = int(m), int(n), int(k)
        self.binary_path = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.nbmf_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_hyper(self, alpha, beta, eps=1e-8, projection=PROJ_NORMALIZE):
        _check(self._lib.nbmf_set_hyper(self._h, float(alpha), float(beta), float(eps), int(projection)))

    def upload(self, x, mask=None, transposed=False):
        """x: the m x n internal matrix, or (transposed=True) the n x m user matrix whose transpose it is.
        A bool / uint8 array goes up as it is, one byte per entry, a float32 array as it is, four (``nbmf_upload_v``: the
        device converts, exactly); anything else as float64."""
        x = np.asarray(x)
        if x.dtype == np.bool_ or x.dtype == np.uint8:
            x, x_kind = np.ascontiguousarray(x).view(np.uint8), DATA_U8
        elif x.dtype == np.float32:
            x, x_kind = np.ascontiguousarray(x), DATA_F32
        else:
            x, x_kind = _f64c(x), DATA_F64
        want = (self.n, self.m) if transposed else (self.m, self.n)
        if x.shape != want:
            raise ValueError(f"data has shape {x.shape}, context expects {want}")
        kind, mptr, ldm = MASK_NONE, None, 0
        if mask is not None:
            mask = np.asarray(mask)
            if mask.shape != x.shape:
                mask = np.broadcast_to(mask, x.shape)      # Y * mask broadcasting, _solver.py:30
            if mask.dtype == np.bool_ or mask.dtype == np.uint8:
                mask = np.ascontiguousarray(mask).view(np.uint8)
                kind = MASK_U8
            else:
                mask = _f64c(mask)
                kind = MASK_F64
            mptr, ldm = mask.ctypes.data_as(c_void_p), mask.shape[1]
        flags = c_int(0)
        _check(self._lib.nbmf_upload_v(self._h, x.ctypes.data_as(c_void_p), x_kind, x.shape[1], int(bool(transposed)),
                                       mptr, kind, ldm, byref(flags)))
        self.binary_path = bool(flags.value & FLAG_BINARY_PATH)
        return self.binary_path

    def upload_csr(self, x_pattern, mask_pattern=None, transposed=False):
        """Binary data from CSR patterns: ``x_pattern`` / ``mask_pattern`` are ``(indptr, indices)`` pairs of the
        user's matrix (rows = its rows) whose stored entries mean 1 / observed; no dense copy is made anywhere."""
        def prep(p):
            ip = np.ascontiguousarray(p[0], dtype=np.int64)
            ix = np.ascontiguousarray(p[1], dtype=np.int32)
            rows = self.n if transposed else self.m
            if ip.shape != (rows + 1,) or ix.shape != (int(ip[-1]),):
                raise ValueError(f"CSR pattern does not describe {rows} rows")
            return ip, ix
        ip, ix = prep(x_pattern)
        mp, mx = prep(mask_pattern) if mask_pattern is not None else (None, None)
        flags = c_int(0)
        _check(self._lib.nbmf_upload_csr(
            self._h, ip.ctypes.data_as(c_void_p), ix.ctypes.data_as(c_void_p), int(ix.size), int(bool(transposed)),
            None if mp is None else mp.ctypes.data_as(c_void_p), None if mx is None else mx.ctypes.data_as(c_void_p),
            0 if mx is None else int(mx.size), byref(flags)))
        self.binary_path = True
        return True

    def generate(self, seed, density=0.25, observed=1.0, row0=0, col0=0, n_global=None):
        """Fill the context with synthetic binary data generated on the device (see synthetic_reference); with
        ``row0`` / ``col0`` / ``n_global`` the context holds that slice of a larger matrix ``n_global`` columns wide
        (every rank of a sharded run generates its own rows of the same global matrix)."""
        _check(self._lib.nbmf_generate_slice(self._h, int(seed), float(density), float(observed), int(row0), int(col0),
                                             int(self.n if n_global is None else n_global)))
        self.binary_path = True

    def set_storage(self, storage="auto"):
        """Hold the next :meth:`upload` to a storage path: "auto" (byte codes when the data allow), "f64" (doubles even
        for binary values: the reference's arithmetic for real-valued V), "f64w" (doubles plus float64 weight tiles)."""
        if storage not in STORAGE:
            raise ValueError(f"storage must be one of {list(STORAGE)}")
        _check(self._lib.nbmf_set_storage(self._h, STORAGE[storage]))

    def n_obs(self):
        v = c_double(0)
        _check(self._lib.nbmf_get_n_obs(self._h, byref(v)))
        return v.value

    def set_factors(self, W_kxm, H_kxn):
        W, H = _f64c(W_kxm), _f64c(H_kxn)
        if W.shape != (self.k, self.m) or H.shape != (self.k, self.n):
            raise ValueError(f"factor shapes {W.shape}, {H.shape} do not match (k,m)=({self.k},{self.m}), "
                             f"(k,n)=({self.k},{self.n})")
        _check(self._lib.nbmf_set_factors(self._h, W.ctypes.data_as(c_void_p), H.ctypes.data_as(c_void_p)))

    def get_factors(self):
        W = np.empty((self.k, self.m), dtype=np.float64)
        H = np.empty((self.k, self.n), dtype=np.float64)
        _check(self._lib.nbmf_get_factors(self._h, W.ctypes.data_as(c_void_p), H.ctypes.data_as(c_void_p)))
        return W, H

    def run(self, max_iter, tol):
        losses = np.empty(int(max_iter), dtype=np.float64)
        n_iter = c_int(0)
        _check(self._lib.nbmf_run(self._h, int(max_iter), float(tol), losses.ctypes.data_as(c_void_p), byref(n_iter)))
        return losses[: n_iter.value].copy(), n_iter.value

    def run_batch(self, alphas, betas, W0, H0, max_iter, tol):
        """Independent fits of this context's data, one per (alpha, beta, W0, H0) -- a prior grid, restarts -- as many at
        a time as the chip holds, in one launch per group (``nbmf_run_batch``).  ``W0`` is (P, k, m) or (k, m) (the same
        start for every problem), ``H0`` likewise.  Returns ``(losses, n_iter, W, H)``: a list of P loss curves, an int
        array, and the final factors (P, k, m) / (P, k, n).  Problem p's results are bitwise those of ``set_hyper`` +
        ``set_factors`` + ``run`` + ``get_factors``."""
        alphas = np.ascontiguousarray(np.atleast_1d(alphas), dtype=np.float64)
        P = alphas.size
        betas = np.ascontiguousarray(np.broadcast_to(np.atleast_1d(np.asarray(betas, dtype=np.float64)), (P,)))
        W0 = np.ascontiguousarray(np.broadcast_to(_f64c(W0), (P, self.k, self.m)))
        H0 = np.ascontiguousarray(np.broadcast_to(_f64c(H0), (P, self.k, self.n)))
        losses = np.zeros((P, int(max_iter)), dtype=np.float64)
        n_iter = np.zeros(P, dtype=np.int32)
        W = np.empty_like(W0)
        H = np.empty_like(H0)
        ptr = lambda a: a.ctypes.data_as(c_void_p)   # noqa: E731
        _check(self._lib.nbmf_run_batch(self._h, int(P), ptr(alphas), ptr(betas), ptr(W0), ptr(H0), int(max_iter), float(tol),
                                        ptr(losses), ptr(n_iter), ptr(W), ptr(H)))
        return [losses[p, :n_iter[p]].copy() for p in range(P)], n_iter, W, H

    def batch_stats(self):
        """(persistent launches made by run_batch on this context, problems they served)."""
        a, b = c_int(0), c_int(0)
        _check(self._lib.nbmf_batch_stats(self._h, byref(a), byref(b)))
        return a.value, b.value

    def sweep_info(self):
        """{"h_chunks", "h_blocks", "w_chunks", "w_blocks"}: how the two sweeps are cut (after an upload / generate)."""
        v = [c_int(0) for _ in range(4)]
        _check(self._lib.nbmf_sweep_info(self._h, *[byref(x) for x in v]))
        return dict(zip(("h_chunks", "h_blocks", "w_chunks", "w_blocks"), (x.value for x in v)))

    def w_only_steps(self, n_steps):
        _check(self._lib.nbmf_w_only_steps(self._h, int(n_steps)))

    def loss(self):
        v = c_double(0)
        _check(self._lib.nbmf_loss(self._h, byref(v)))
        return v.value

    def loglik(self, clip_theta=False):
        """Data log-likelihood sum of the current factors (no prior, not divided by n_obs);
        clip_theta clips W^T H to [0, 1] first (the reference's inverse_transform)."""
        v = c_double(0)
        _check(self._lib.nbmf_loglik(self._h, int(bool(clip_theta)), byref(v)))
        return v.value

    def loglik_strict(self):
        """Log-likelihood summed over observed entries only (held-out perplexity numerator)."""
        v = c_double(0)
        _check(self._lib.nbmf_loglik_strict(self._h, byref(v)))
        return v.value

    def comm_init(self, uid: bytes, nranks: int, rank: int, shard_axis: int = 0):
        """Attach an RCCL communicator; shard_axis 0 = rows of the internal Y are split, 1 = columns."""
        buf = ctypes.create_string_buffer(bytes(uid), 128)
        _check(self._lib.nbmf_comm_init(self._h, buf, int(nranks), int(rank), int(shard_axis)))

    def comm_init_host(self, allreduce, nranks: int, rank: int, shard_axis: int = 0):
        """Attach a host-mediated all-reduce: ``allreduce(arr)`` must sum the float64 NumPy array
        ``arr`` over all ranks IN PLACE (e.g. gloo).  Same device work as the RCCL path."""
        def _cb(_user, ptr, count):
            try:
                arr = np.ctypeslib.as_array(ptr, shape=(int(count),))
                allreduce(arr)
                return 0
            except Exception:      # never unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1
        self._host_cb = HOST_ALLREDUCE_FN(_cb)          # keep alive as long as the context
        _check(self._lib.nbmf_comm_init_host(self._h, self._host_cb, None, int(nranks), int(rank), int(shard_axis)))

    def peer_export(self, shard_axis: int = 0) -> bytes:
        """Allocate this rank's exchange arena for the peer (xGMI) transport and return its opaque
        PEER_HANDLE_BYTES handle block; all-gather the blocks in rank order and pass them to
        :meth:`comm_init_peer`."""
        buf = ctypes.create_string_buffer(PEER_HANDLE_BYTES)
        _check(self._lib.nbmf_peer_export(self._h, int(shard_axis), buf))
        return buf.raw

    def comm_init_peer(self, handles: bytes, nranks: int, rank: int, shard_axis: int = 0):
        """Attach the peer transport (one process per rank); on failure the context stays unattached."""
        handles = bytes(handles)
        if len(handles) != PEER_HANDLE_BYTES * int(nranks):
            raise ValueError(f"expected {PEER_HANDLE_BYTES} handle bytes per rank")
        buf = ctypes.create_string_buffer(handles, len(handles))
        _check(self._lib.nbmf_comm_init_peer(self._h, buf, int(nranks), int(rank), int(shard_axis)))

    def set_exchange_panels(self, panels=0):
        """Row split: exchange in 1 or 2 column panels at the next attach (0 = default); same on every rank."""
        _check(self._lib.nbmf_set_exchange_panels(self._h, int(panels)))

    def set_peer_timeout_ms(self, ms=0.0):
        """Bound on every wait of the peer transport attached next (0 = default: NBMF_PEER_TIMEOUT_MS or 30 s)."""
        _check(self._lib.nbmf_set_peer_timeout_ms(self._h, float(ms)))

    def comm_detach(self):
        """Drop the attached communicator; every rank must do the same."""
        _check(self._lib.nbmf_comm_detach(self._h))
        self._host_cb = None

    def cancel(self):
        """From ANOTHER thread: make the owner's ``run`` return (NBMFHipError "cancelled") at its next iteration boundary and
        cut short what it has enqueued (``nbmf_cancel``).  Sticky: the context can only be closed afterwards."""
        if self._h:
            _check(self._lib.nbmf_cancel(self._h))

    COMM_KINDS = {0: "none", 1: "rccl", 2: "peer", 3: "host"}

    def comm_info(self):
        """What the attached transport itself reports (``nbmf_comm_info``): ``{"kind", "nranks_seen", "remote"}`` --
        RCCL: ncclCommCount / ncclCommCuDevice of the communicator (-1 where librccl lacks the symbol); peer: arenas
        mapped (own included) / how many of them belong to other ranks; host: as given."""
        k, n, r = c_int(0), c_int(0), c_int(0)
        _check(self._lib.nbmf_comm_info(self._h, byref(k), byref(n), byref(r)))
        return {"kind": self.COMM_KINDS.get(k.value, str(k.value)), "nranks_seen": n.value, "remote": r.value}

    def timing_enable(self, on=True):
        """on: False / True, or an int n > 1 for the sweeps of every n-th iteration only."""
        _check(self._lib.nbmf_timing_enable(self._h, int(on) if not isinstance(on, bool) else int(on)))

    def timing(self):
        hm, wm, hn, wn = c_double(0), c_double(0), c_int(0), c_int(0)
        _check(self._lib.nbmf_timing_get(self._h, byref(hm), byref(hn), byref(wm), byref(wn)))
        return {"hpass_ms": hm.value, "hpass_launches": hn.value, "wpass_ms": wm.value, "wpass_launches": wn.value}

    def synchronize(self):
        _check(self._lib.nbmf_synchronize(self._h))

    def small_stats(self):
        """(runs served by the single-launch path for small problems, how many of those fell back)."""
        r, a = c_int(0), c_int(0)
        _check(self._lib.nbmf_small_stats(self._h, byref(r), byref(a)))
        return r.value, a.value

    def set_progress(self, callback=None, every=10):
        """``callback(first, losses)`` is called from inside :meth:`run` with the losses of iterations
        ``first, first+1, ...`` as they become final (every ``every`` iterations); None switches it off."""
        if callback is None:
            self._progress_cb = None
            _check(self._lib.nbmf_set_progress(self._h, PROGRESS_FN(), None, 0))
            return

        def _cb(_user, first, count, ptr):
            try:
                callback(int(first), [ptr[i] for i in range(int(count))])
            except Exception:      # never unwind through the C frame
                import traceback
                traceback.print_exc()
        self._progress_cb = PROGRESS_FN(_cb)             # keep alive as long as the context
        _check(self._lib.nbmf_set_progress(self._h, self._progress_cb, None, int(every)))


def device_synchronize(device=0):
    _check(load().nbmf_device_synchronize(int(device)))


def comm_unique_id() -> bytes:
    buf = ctypes.create_string_buffer(128)
    _check(load().nbmf_comm_unique_id(buf))
    return buf.raw


def engine_stats():
    """Process-wide: (fits the single-launch engine served, persistent launches that gave up, nbmf_run calls the
    launch-per-kernel engine served)."""
    a, b, c = ctypes.c_longlong(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
    _check(load().nbmf_engine_stats(byref(a), byref(b), byref(c)))
    return a.value, b.value, c.value


def variant_stats():
    """Process-wide: (sweeps launched in the two-state W variant, sweeps launched in the ragged-K variant, runs that resumed
    after a sweep could not assemble a loss within its bound)."""
    a, b, c = ctypes.c_longlong(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
    _check(load().nbmf_variant_stats(byref(a), byref(b), byref(c)))
    return a.value, b.value, c.value


def mfma_peak(device=0, target_ms=100.0):
    """The f64 MFMA rate of ``device`` as measured now: ``{"tflops", "cycles_per_mfma_at_2p4GHz", "launch_ms"}``
    (``nbmf_selftest_mfma_peak``: a loop of bare v_mfma_f64_16x16x4_f64, VGPR accumulators, all SIMDs)."""
    t, cy, ms = c_double(0), c_double(0), c_double(0)
    _check(load().nbmf_selftest_mfma_peak(int(device), float(target_ms), byref(t), byref(cy), byref(ms)))
    return {"tflops": t.value, "cycles_per_mfma_at_2p4GHz": cy.value, "launch_ms": ms.value}


def selftest_unary(op, x, device=0):
    """Apply a device scalar routine of the pass kernel (0: Newton reciprocal, 1: log) to `x` (GPU tests)."""
    d = _f64c(x).ravel()
    out = np.empty_like(d)
    _check(load().nbmf_selftest_unary(int(device), int(op), int(d.size), d.ctypes.data_as(c_void_p),
                                      out.ctypes.data_as(c_void_p)))
    return out


def synthetic_reference(m, n, seed, density=0.25, observed=1.0, rows=None, cols=None):
    """NumPy regeneration of ``Context.generate`` (same splitmix64 hash of (seed, i*n + j)): returns
    (Y float64, mask bool) of the m x n matrix, or of its sub-matrix ``[rows][:, cols]`` when index
    arrays are given (entries are hashed independently, so a slice costs only its own size).
    For the tests; O(rows * cols) host memory."""
    def mix(x):
        x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))
    ri = np.arange(m, dtype=np.uint64) if rows is None else np.asarray(rows, dtype=np.uint64)
    ci = np.arange(n, dtype=np.uint64) if cols is None else np.asarray(cols, dtype=np.uint64)
    with np.errstate(over="ignore"):
        idx = ri[:, None] * np.uint64(n) + ci[None, :]
        base = np.uint64(seed) * np.uint64(0x100000001B3) + idx
        u = (mix(base) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        v = (mix(base ^ np.uint64(0xD6E8FEB86659FD93)) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (u < density).astype(np.float64), v < observed
