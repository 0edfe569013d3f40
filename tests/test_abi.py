"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/nbmf_hip.h declares, fails loudly (no CPU fallback), and nothing under nbmf_mm_amd/ imports
the oracle."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "nbmf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nbmf_[a-z0-9_]+)\s*\(", text)) - {"nbmf_host_allreduce_fn"})


def test_library_exports_every_declared_symbol():
    from nbmf_mm_amd import _hip
    lib = _hip.load()
    declared = _declared_symbols()
    assert len(declared) >= 19
    assert sorted(_hip.SYMBOLS) == declared, "python binding list and header disagree"
    for name in declared:
        assert hasattr(lib, name), f"libnbmf_hip.so does not export {name}"
    assert lib.nbmf_abi_version() == 4


def test_library_is_the_build_of_these_sources():
    """The library carries the content hash of the sources it was compiled from (csrc/Makefile -> nbmf_source_hash): it
    must be the hash of the tree's sources -- on the GPU box too, where the .so travels with the snapshot: a stale binary
    cannot stand in for the sources it is shipped with.  tools/src_hash.sh prints the same."""
    import subprocess
    from nbmf_mm_amd import _hip
    if "NBMF_HIP_LIBRARY" in os.environ:
        pytest.skip("an explicitly named build")
    want = _hip.tree_source_hash()
    assert want is not None and re.fullmatch(r"[0-9a-f]{12}", want)
    assert _hip.source_hash() == want, "libnbmf_hip.so was not built from the sources in the tree: run make -C nbmf_mm_amd/csrc"
    sh = subprocess.run(["bash", os.path.join(ROOT, "tools", "src_hash.sh")], capture_output=True, text=True)
    assert sh.stdout.strip() == want


def test_no_gpu_means_loud_failure_not_fallback():
    from nbmf_mm_amd import NBMF, _hip
    if _hip.device_count() > 0:
        pytest.skip("a GPU is visible here")
    X = (np.random.default_rng(0).random((20, 30)) < 0.3).astype(float)
    with pytest.raises(_hip.NBMFHipError, match="no HIP device"):
        NBMF(n_components=3, max_iter=5).fit(X)
    with pytest.raises(_hip.NBMFHipError):
        _hip.Context(20, 30, 3)


def test_missing_library_is_an_error(monkeypatch):
    from nbmf_mm_amd import _hip
    monkeypatch.setenv("NBMF_HIP_LIBRARY", "/nonexistent/libnbmf_hip.so")
    monkeypatch.setattr(_hip, "_lib", None)
    with pytest.raises(_hip.NBMFHipError, match="no CPU fallback"):
        _hip.load()
    monkeypatch.undo()
    _hip._lib = None
    _hip.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "nbmf_mm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports the oracle"


def test_argument_errors_surface_without_a_gpu():
    from nbmf_mm_amd import _hip
    lib = _hip.load()
    import ctypes
    h = ctypes.c_void_p()
    assert lib.nbmf_create(0, 5, 3, 0, ctypes.byref(h)) == _hip.NBMF_ERR_ARG
    assert b"m and n" in lib.nbmf_last_error()
    assert lib.nbmf_create(5, 5, _hip.MAX_K + 1, 0, ctypes.byref(h)) == _hip.NBMF_ERR_ARG
    assert b"n_components" in lib.nbmf_last_error()
    assert lib.nbmf_run(None, 1, 0.0, None, None) == _hip.NBMF_ERR_ARG
    assert lib.nbmf_destroy(None) == 0


def test_header_compiles_as_c99_and_links():
    """include/nbmf_hip.h is a C header: the C consumer builds against it and the library with gcc."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = os.path.join(ROOT, "build", "abi_smoke_cpu")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-L", os.path.join(ROOT, "nbmf_mm_amd"),
                           "-lnbmf_hip", "-Wl,-rpath," + os.path.join(ROOT, "nbmf_mm_amd"), "-lm", "-o", exe])
    from nbmf_mm_amd import _hip
    if _hip.device_count() == 0:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
        assert out.returncode == 3 and "no GPU" in out.stderr      # fails loudly without a device


def test_cpu_reachable_paths_under_host_asan():
    """SURVEY section 5 (sanitizers): the library built with AddressSanitizer on the host side (`make asan`; device code as
    usual) and every entry point of the ABI called with null pointers and zero sizes in a child process that preloads the
    ASan runtime (tests/asan_null_calls.py): error codes, no fault, no ASan report.  What a host without a GPU can reach of
    the C layer: argument validation, the error plumbing and its thread-local message."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not hipcc or not os.path.exists(clang):
        pytest.skip("needs hipcc and its clang")
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("no AddressSanitizer runtime beside the compiler")
    mk = subprocess.run(["make", "-C", os.path.join(ROOT, "nbmf_mm_amd", "csrc"), "asan"], capture_output=True, text=True)
    assert mk.returncode == 0, (mk.stdout + mk.stderr)[-2000:]
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               NBMF_HIP_LIBRARY=os.path.join(ROOT, "build", "libnbmf_hip_asan.so"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_null_calls.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "ASAN_NULL_CALLS_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
    assert "AddressSanitizer" not in r.stderr
