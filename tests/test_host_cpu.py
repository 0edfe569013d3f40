"""Host-side behaviour of the estimator that needs no GPU: validation order, orientation aliases,
error messages, sklearn protocol (mirrors the reference's tests/test_api.py:139-153,
tests/test_nbmf_mm.py:104-111, tests/test_public_api.py)."""
import numpy as np
import pytest
from sklearn.base import clone

from nbmf_mm_amd import NBMF, NBMFMM, nbmf_mm_solver
from nbmf_mm_amd._utils import check_is_fitted, generate_synthetic_binary_data


def test_exports_and_alias():
    import nbmf_mm_amd
    assert nbmf_mm_amd.__all__ == ["NBMFMM", "NBMF", "nbmf_mm_solver"]
    assert NBMF is NBMFMM
    assert callable(nbmf_mm_solver)


def test_constructor_defaults_match_reference():
    p = NBMF().get_params()
    ref = dict(n_components=10, alpha=1.2, beta=1.2, max_iter=2000, tol=1e-5, W_init=None, H_init=None, init=None,
               random_state=None, verbose=0, orientation="beta-dir")
    for k, v in ref.items():
        assert p[k] == v
    assert p["projection"] == "normalize" and p["n_init"] == 1        # extensions
    c = clone(NBMF(n_components=3, orientation="Dir-Beta", projection_method="duchi"))
    assert c.orientation == "Dir-Beta" and c.projection_method == "duchi"
    NBMF(init="custom")                                               # accepted and ignored


def test_out_of_range_is_rejected_before_any_device_work():
    X = np.random.default_rng(0).normal(size=(50, 30))
    with pytest.raises(ValueError, match="must be binary"):
        NBMFMM(n_components=5).fit(X)


def test_bad_orientation_and_alias_table():
    X = (np.random.default_rng(0).random((10, 12)) < 0.3).astype(float)
    with pytest.raises(ValueError, match="Unknown orientation"):
        NBMF(n_components=2, orientation="beta_dir").fit(X)
    with pytest.raises(ValueError, match="Unknown orientation"):
        NBMF(n_components=2, orientation="BETA-DIR").fit(X)          # exact-string table, no case folding
    m = NBMF()
    assert m._normalize_orientation("Aspect Bernoulli") == "dir-beta"
    assert m._normalize_orientation("bICA") == "beta-dir"
    assert m._normalize_orientation("Dir Beta") == "dir-beta"


def test_nan_and_1d_rejected_by_check_array():
    with pytest.raises(ValueError):
        NBMF(n_components=2).fit(np.array([[0.0, np.nan], [1.0, 0.0]]))
    with pytest.raises(ValueError):
        NBMF(n_components=2).fit(np.array([0.0, 1.0, 1.0]))


def test_not_fitted_message():
    m = NBMF(n_components=2)
    for call in (lambda: m.transform(np.zeros((2, 2))), lambda: m.inverse_transform(np.zeros((2, 2))),
                 lambda: m.score(np.zeros((2, 2)))):
        with pytest.raises(ValueError, match="This NBMFMM instance is not fitted yet."):
            call()
    with pytest.raises(ValueError):
        check_is_fitted(m, "components_")


def test_unknown_projection_and_bad_max_iter():
    X = (np.random.default_rng(0).random((10, 12)) < 0.3).astype(float)
    with pytest.raises(ValueError, match="Unknown projection"):
        nbmf_mm_solver(X, 2, projection="softmax")
    with pytest.raises(ValueError, match="max_iter"):
        nbmf_mm_solver(X, 2, max_iter=0)


def test_inverse_transform_is_host_math():
    m = NBMF(n_components=2)
    m.components_ = np.array([[0.2, 0.9, 1.5], [0.4, 0.1, 0.0]])
    out = m.inverse_transform(np.array([[0.5, 0.5], [1.0, 0.0]]))
    np.testing.assert_allclose(out, np.clip(np.array([[0.3, 0.5, 0.75], [0.2, 0.9, 1.5]]), 0, 1))


def test_synthetic_generator_is_seed_compatible():
    X, W, H = generate_synthetic_binary_data(50, 30, 5, random_state=42)
    assert X.shape == (50, 30) and W.shape == (50, 5) and H.shape == (5, 30)
    assert set(np.unique(X)) <= {0.0, 1.0}
    rs = np.random.RandomState(42)
    np.testing.assert_array_equal(W, rs.uniform(0.1, 0.9, size=(50, 5)))
