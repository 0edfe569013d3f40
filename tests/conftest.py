import os
import sys

import numpy as np
import pytest

try:
    # Load order matters when PyTorch shares the process: its wheel bundles a private ROCm runtime
    # (libhsa-runtime64.so, libamdhip64.so).  RCCL finds the HSA runtime by plain soname, so if torch is
    # imported AFTER the system HIP runtime has been initialised, RCCL picks torch's uninitialised copy
    # ("no ROCm-capable device").  So the tests import torch first.  (bench.py and the library
    # never import torch.)
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is only needed by the multi-process tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def config1_X():
    return (np.random.default_rng(0).random((100, 500)) < 0.25).astype(np.float64)


def config1_mask():
    return np.random.default_rng(1).random((100, 500)) < 0.9


def midsize_XM():
    g = np.random.default_rng(0)
    X = (g.random((512, 512)) < 0.25).astype(np.float64)
    M = (g.random((512, 512)) < 0.9).astype(np.float64)
    return X, M


class _EnginePath(str):
    """The fixture's value: compares as the engine's name, and knows how many fits each engine has served since the
    test began (``served()`` -> (single-launch fits, persistent launches that gave up, launch-engine runs))."""

    def served(self):
        from nbmf_mm_amd import _hip
        now = _hip.engine_stats()
        return tuple(a - b for a, b in zip(now, self.start))


@pytest.fixture(params=["single-launch", "five-kernels"])
def both_small_paths(request, monkeypatch):
    """Small problems take the single-launch path (one persistent kernel runs the whole loop) unless it is
    switched off; tests that use this fixture run once each way, so both engines face the same oracle.

    The teardown makes sure the engine a case names is the one that ran: under "single-launch" no persistent kernel
    may have given up (the library would silently redo such a fit with the launches, and every parity assertion
    would pass without having tested the persistent kernel), under "five-kernels" the persistent kernel may not have
    served anything.  Problems that do not qualify for the persistent kernel (K > 32, large shapes) run on the
    launches either way; tests whose problems all qualify assert ``served()[0] >= 1`` themselves."""
    from nbmf_mm_amd import _hip
    monkeypatch.setenv("NBMF_PERSISTENT", "1" if request.param == "single-launch" else "0")
    path = _EnginePath(request.param)
    path.start = _hip.engine_stats()
    yield path
    persistent, gave_up, _ = path.served()
    if request.param == "single-launch":
        assert gave_up == 0, f"{gave_up} persistent launches gave up: those fits were served by the launches"
    else:
        assert persistent == 0, f"{persistent} fits were served by the persistent kernel with NBMF_PERSISTENT=0"
