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
