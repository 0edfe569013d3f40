"""Host helpers mirrored from the reference (src/nbmf_mm/_utils.py)."""
import numpy as np


def check_is_fitted(estimator, attributes):
    """Same contract and message as src/nbmf_mm/_utils.py:3-9."""
    names = [attributes] if isinstance(attributes, str) else list(attributes)
    missing = [a for a in names if not hasattr(estimator, a)]
    if missing:
        raise ValueError(f"This {type(estimator).__name__} instance is not fitted yet.")


def generate_synthetic_binary_data(n_samples=100, n_features=50, n_components=5, sparsity=0.3,
                                   random_state=None):
    """Logistic-link synthetic generator with the reference's draw order
    (src/nbmf_mm/_utils.py:38-46) so that seeds produce the same matrices."""
    rs = np.random.RandomState(random_state)
    W_true = rs.uniform(0.1, 0.9, size=(n_samples, n_components))
    H_true = (rs.random((n_components, n_features)) < sparsity).astype(float)
    prob = 1 / (1 + np.exp(-W_true @ H_true))
    X = (rs.random((n_samples, n_features)) < prob).astype(float)
    return X, W_true, H_true
