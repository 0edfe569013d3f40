"""Small host-side helpers with the behaviour of the reference's ``_utils`` module
(src/nbmf_mm/_utils.py): the not-fitted check and the logistic-link test-data generator."""
import numpy as np

_NOT_FITTED = "This {name} instance is not fitted yet."


def check_is_fitted(estimator, attributes):
    """Raise ``ValueError("This <Class> instance is not fitted yet.")`` unless every attribute named in
    ``attributes`` (a name or a list of names) exists on ``estimator`` (src/nbmf_mm/_utils.py:3-9)."""
    wanted = (attributes,) if isinstance(attributes, str) else tuple(attributes)
    if not all(hasattr(estimator, a) for a in wanted):
        raise ValueError(_NOT_FITTED.format(name=type(estimator).__name__))


def generate_synthetic_binary_data(n_samples=100, n_features=50, n_components=5, sparsity=0.3,
                                   random_state=None):
    """Binary matrix drawn from Bernoulli(sigmoid(W_true @ H_true)).

    Seed-compatible with the reference generator (src/nbmf_mm/_utils.py:38-46): one legacy
    ``RandomState`` consumed in the order uniform W_true, binary H_true, uniform thresholds for X, so the
    same ``random_state`` yields the same three arrays.  Returns ``(X, W_true, H_true)`` as float arrays.
    """
    gen = np.random.RandomState(random_state)
    shape_w, shape_h, shape_x = (n_samples, n_components), (n_components, n_features), (n_samples, n_features)
    W_true = gen.uniform(0.1, 0.9, size=shape_w)
    H_true = np.where(gen.random(shape_h) < sparsity, 1.0, 0.0)
    logits = W_true @ H_true
    p_one = 1 / (1 + np.exp(-logits))
    X = np.where(gen.random(shape_x) < p_one, 1.0, 0.0)
    return X, W_true, H_true
