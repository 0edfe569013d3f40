"""Run under a host-AddressSanitizer build of libnbmf_hip (tests/test_abi.py starts it with LD_PRELOAD=<asan runtime> and
NBMF_HIP_LIBRARY=<the ASan build>): every entry point of the C ABI is called with null pointers and zero sizes -- the
CPU-reachable part of the library: argument validation, error plumbing, the thread-local error string -- and must come
back with an error code, not a fault.  No GPU needed, and none used: the calls fail before any device work -- except the
entries that take a bare DEVICE INDEX (nbmf_device_synchronize, the self-tests), which are handed device -1 here so
that they, too, fail in argument validation on a box that has a GPU instead of initialising the runtime under the
preloaded sanitizer."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbmf_mm_amd import _hip  # noqa: E402

lib = _hip.load()
assert lib.nbmf_abi_version() == 4
results = {}
for name in _hip.SYMBOLS:
    if name in ("nbmf_abi_version", "nbmf_last_error", "nbmf_source_hash"):
        continue
    fn = getattr(lib, name)
    args = []
    bare_device = name in ("nbmf_device_synchronize", "nbmf_selftest_unary", "nbmf_selftest_mfma_peak")   # first argument: a device index
    for i, t in enumerate(fn.argtypes or []):
        if t in (ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_longlong):
            args.append(-1 if (bare_device and i == 0) else 0)
        elif t is ctypes.c_double:
            args.append(0.0)
        elif t in (_hip.HOST_ALLREDUCE_FN, _hip.PROGRESS_FN):
            args.append(t())                 # a NULL callback
        else:
            args.append(None)                # every pointer NULL
    rc = fn(*args)
    msg = lib.nbmf_last_error()
    results[name] = (rc, msg)
    assert isinstance(rc, int)
# with every pointer NULL nothing may succeed except the calls that have nothing to do (pure queries with optional outputs)
ok_allowed = {"nbmf_engine_stats", "nbmf_variant_stats", "nbmf_destroy"}   # (destroying nothing is fine, like free(NULL))
bad = {k: v for k, v in results.items() if v[0] == 0 and k not in ok_allowed}
assert not bad, bad
# the error text is per thread and survives until the next failure on that thread
import threading
seen = []
def other():
    lib.nbmf_run(None, 1, 0.0, None, None)
    seen.append(lib.nbmf_last_error())
lib.nbmf_create(10, 10, 0, 0, None)
mine = lib.nbmf_last_error()
t = threading.Thread(target=other)
t.start()
t.join()
assert lib.nbmf_last_error() == mine and seen and seen[0] != mine
print("ASAN_NULL_CALLS_OK", len(results))
