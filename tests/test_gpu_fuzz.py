"""A fixed-seed slice of the randomised parity sweep (tests/manual/fuzz_vs_oracle.py) in the GPU suite: shapes, K, data
and mask kinds, orientations, projections, eps, inits inside and outside the range a fit keeps, both engines -- every
case against the oracle (losses rtol 1e-9, factors atol 1e-8)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_randomised_cases_against_the_oracle(monkeypatch):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "manual", "fuzz_vs_oracle.py")
    spec = importlib.util.spec_from_file_location("fuzz_vs_oracle", path)
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    monkeypatch.setenv("NBMF_PERSISTENT", "1")        # (the sweep sets it per case; restored afterwards)
    bad, _ = fuzz.run(120, 3, max_dim=500)
    assert bad == 0


def test_randomised_evaluation_calls_against_the_oracle():
    """... and of tests/manual/fuzz_eval_vs_oracle.py: `nbmf_loss`, `nbmf_loglik` (clipped), `nbmf_loglik_strict` and
    `nbmf_w_only_steps` on random factors -- on the simplex or off it, H beyond 1 -- against the oracle's `mm_loss`, `score`,
    `heldout_perplexity` and its transform loop (sums rtol 1e-10, W atol 1e-9)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "manual", "fuzz_eval_vs_oracle.py")
    spec = importlib.util.spec_from_file_location("fuzz_eval_vs_oracle", path)
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad, _ = fuzz.run(250, 11, max_dim=700)
    assert bad == 0


_SHARDED_FUZZ = r"""
import importlib.util, os, sys
root = sys.argv[1]
sys.path.insert(0, root)
spec = importlib.util.spec_from_file_location("fuzz_sharded_vs_single", os.path.join(root, "tests", "manual", "fuzz_sharded_vs_single.py"))
fuzz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fuzz)
bad, _ = fuzz.run(120, 5, max_dim=800)
print("RESULT", bad)
"""


def test_randomised_sharded_fits_against_the_single_context_fit():
    """... and of tests/manual/fuzz_sharded_vs_single.py: 2-6 rank threads on device 0 (peer or host transport), shapes down to
    fewer rows than ranks (refused), shards that choose different storage paths, both orientations and projections, stop
    rules -- against the fit of the whole matrix in one context (same iteration count, losses rtol 1e-10, factors atol 1e-9).
    In a child process: the ranks' streams need hardware queues of their own, set before the runtime starts."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="32", NBMF_PEER_TIMEOUT_MS="20000")
    r = subprocess.run([sys.executable, "-c", _SHARDED_FUZZ, root], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1] == "RESULT 0", r.stdout[-3000:]


def test_randomised_batches_against_one_run_per_problem():
    """... and of tests/manual/fuzz_batch_vs_runs.py: `nbmf_run_batch` (P fits of one data set: a grid of priors, restarts)
    against set_hyper + set_factors + run + get_factors per problem, BIT FOR BIT -- loss curves, iteration counts under
    a stop rule that fires at different iterations for different problems, factors."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "manual", "fuzz_batch_vs_runs.py")
    spec = importlib.util.spec_from_file_location("fuzz_batch_vs_runs", path)
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad, _ = fuzz.run(150, 9, max_dim=1200)
    assert bad == 0
