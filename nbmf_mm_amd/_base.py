"""scikit-learn style estimator with the reference's public surface
(src/nbmf_mm/_base.py:7-269): same constructor arguments, methods, attributes, orientation
aliases and error messages; ``fit`` and ``transform`` run on an MI355X through libnbmf_hip.
"""
import inspect

import numpy as np
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.utils import check_array

from ._solver import device_score, nbmf_mm_restarts, nbmf_mm_solver, w_only_transform
from ._utils import check_is_fitted

# Accepted spellings of the two orientations: the exact strings of src/nbmf_mm/_base.py:127-137
# (matching is exact, not case-folding; the order is the reference's, it shows in the error message).
_SPELLINGS = (
    ("beta-dir", "beta-dir"), ("dir-beta", "dir-beta"), ("Beta-Dir", "beta-dir"), ("Dir-Beta", "dir-beta"),
    ("Dir Beta", "dir-beta"), ("binary ICA", "beta-dir"), ("Binary ICA", "beta-dir"), ("bICA", "beta-dir"),
    ("Aspect Bernoulli", "dir-beta"),
)
_ORIENTATION_ALIASES = dict(_SPELLINGS)
# (the name of check_array's switch for its NaN / inf pass changed with scikit-learn 1.6)
_FINITE_KW = "ensure_all_finite" if "ensure_all_finite" in inspect.signature(check_array).parameters else "force_all_finite"


class NBMFMM(BaseEstimator, TransformerMixin):
    """Mean-parameterised Bernoulli matrix factorisation by majorisation-minimisation
    (Magron & Fevotte 2022), GPU-resident.

    Parameters follow src/nbmf_mm/_base.py:63-66.  Extensions: ``projection`` (alias
    ``projection_method``) in {"normalize", "duchi"} (README.md:27-35 of the reference), ``n_init``
    (README.md:144: keep the best of several random restarts), ``device``, and ``n_gpus`` / ``devices``: ``fit`` with the
    rows of X sharded over several GPUs of this machine, driven from this one process (SURVEY section 5's config row;
    one exchange of the K x N H-step products per iteration, BASELINE.json's north star) -- same results to 1e-12.

    Attributes after ``fit``: ``W_`` (n_samples, k), ``components_`` (k, n_features),
    ``loss_curve_`` / ``objective_history_``, ``loss_`` / ``reconstruction_err_``, ``n_iter_``
    (_base.py:114-120).
    """

    def __init__(self, n_components=10, alpha=1.2, beta=1.2, max_iter=2000, tol=1e-5, W_init=None,
                 H_init=None, init=None, random_state=None, verbose=0, orientation="beta-dir",
                 projection="normalize", projection_method=None, n_init=1, device=0, n_gpus=1, devices=None):
        self.n_components = n_components
        self.alpha = alpha
        self.beta = beta
        self.max_iter = max_iter
        self.tol = tol
        self.W_init = W_init
        self.H_init = H_init
        self.init = init            # accepted and ignored, as in the reference (_base.py:74)
        self.random_state = random_state
        self.verbose = verbose
        self.orientation = orientation
        self.projection = projection
        self.projection_method = projection_method
        self.n_init = n_init
        self.device = device
        self.n_gpus = n_gpus            # fit() shards the rows of X over this many GPUs from this one process
        self.devices = devices          # ... these ones (default 0 .. n_gpus-1); transform / score use `device`

    # -- helpers -----------------------------------------------------------------------------
    def _normalize_orientation(self, orientation):
        try:
            return _ORIENTATION_ALIASES[orientation]
        except (KeyError, TypeError):
            raise ValueError(f"Unknown orientation: {orientation}. "
                             f"Must be one of {list(_ORIENTATION_ALIASES.keys())}") from None

    def _projection(self):
        return self.projection_method if self.projection_method is not None else self.projection

    @staticmethod
    def _validated(X, keep_sparse=False):
        # _base.py:83 converts every input to float64.  A dense bool / uint8 matrix stays as it is here -- one byte per
        # entry up to the device, where the pack kernel reads the bytes (nbmf_upload_v): the same fit bit for bit, without
        # the 8-byte-per-entry host copy (BASELINE configs[4]: 6.1 GB instead of 49 GB).  Shape and dimension checks are
        # check_array's either way; a uint8 value above 1 is "X must be binary" (:90-91), raised by the device pack.
        dt = getattr(X, "dtype", None)
        if dt is not None and not hasattr(X, "toarray") and (dt == np.bool_ or dt == np.uint8):
            X = check_array(X, dtype=None)
            return X
        if isinstance(X, np.ndarray) and dt == np.float32 and X.ndim == 2 and X.size > (1 << 24):
            # a big float32 matrix likewise: four bytes per entry up to the device, which converts (exactly) and checks
            return check_array(X, dtype=None, **{_FINITE_KW: False})
        # A big dense float array: check_array's pass over every entry for NaN / inf (0.16 s on the 4.3 GB of BASELINE
        # configs[2], a third of a 50-iteration fit) is left to the device pack, which reads every entry anyway and counts
        # the ones that are not finite or out of range; if it finds any, `_finite_or_binary_error` runs sklearn's own check
        # then, so the error raised -- and its wording -- is the reference's, in the reference's order (:83 before :90-91).
        big_dense = (isinstance(X, np.ndarray) and X.dtype == np.float64 and X.ndim == 2 and X.size > (1 << 24))
        kw = {_FINITE_KW: False} if big_dense else {}
        X = check_array(X, accept_sparse="csr", dtype=np.float64, **kw)     # _base.py:83
        if hasattr(X, "toarray") and not keep_sparse:
            X = X.toarray()                                            # :86-87
        return X

    @staticmethod
    def _finite_or_binary_error(X, err):
        """The device pack refused X ("X must be binary": something outside [0, 1] or not finite).  For an input whose
        finite check was left to the device this is where sklearn's check runs: NaN / inf raise ITS error (_base.py:83),
        anything else the reference's ValueError (:90-91)."""
        if isinstance(X, np.ndarray) and X.dtype in (np.float64, np.float32) and "must be binary" in str(err):
            check_array(X, dtype=None)
        raise err

    # -- estimator API ---------------------------------------------------------------------------
    def fit(self, X, y=None, mask=None):
        """Fit the factorisation to X (entries in [0, 1]); ``mask`` marks observed entries."""
        X = self._validated(X, keep_sparse=True)       # sparse V stays sparse up to the device (the solver decides)
        if hasattr(X, "toarray"):
            if X.nnz and (X.data.min() < 0 or X.data.max() > 1):
                raise ValueError("X must be binary")                  # :90-91 on the stored values (zeros are fine)
            return self._fit_validated(X, mask, sparse=True)
        return self._fit_validated(X, mask, sparse=False)

    def _fit_validated(self, X, mask, sparse):
        # "X must be binary" (:90-91).  For big inputs the device pack, which sees every entry anyway, raises
        # it (a host-side pass costs 0.3 s on the 4 GB of BASELINE configs[2], more than the upload itself);
        # small inputs -- and any input whose orientation is bad too, to keep the reference's order of
        # errors -- are checked here as the reference does.
        big = sparse or X.size > (1 << 24)
        try:
            orientation = self._normalize_orientation(self.orientation)
        except ValueError:
            big = False
            raise
        finally:
            if not big and not sparse and X.dtype != np.bool_:
                if X.dtype in (np.float64, np.float32) and X.size > (1 << 24):
                    check_array(X, dtype=None)                  # the NaN / inf pass _validated left to the device (:83 comes first)
                if not np.all((X >= 0) & (X <= 1)):
                    raise ValueError("X must be binary") from None
        self.orientation = orientation                                # written back, :95
        n_init = int(self.n_init)
        if n_init < 1:
            raise ValueError("n_init must be >= 1")
        multi = dict(n_gpus=self.n_gpus, devices=self.devices) if (int(self.n_gpus) != 1 or self.devices is not None) else {}
        try:
            return self._run_fit(X, mask, orientation, n_init, multi)
        except ValueError as e:
            self._finite_or_binary_error(X, e)

    def _run_fit(self, X, mask, orientation, n_init, multi):
        if n_init > 1 and not self.verbose and not multi:
            # restarts share one upload and one library call; small problems run several at a time in one launch
            best, _ = nbmf_mm_restarts(
                X, self.n_components, n_init, max_iter=self.max_iter, tol=self.tol, alpha=self.alpha, beta=self.beta,
                W_init=self.W_init, H_init=self.H_init, mask=mask, random_state=self.random_state, orientation=orientation,
                projection=self._projection(), device=self.device)
            n_init = 0
        else:
            best = None
        for r in range(n_init):
            seed = self.random_state
            if r > 0 and seed is not None:
                seed = seed + r                                       # restarts: consecutive seeds
            result = nbmf_mm_solver(
                Y=X, n_components=self.n_components, max_iter=self.max_iter, tol=self.tol,
                alpha=self.alpha, beta=self.beta, W_init=self.W_init, H_init=self.H_init, mask=mask,
                random_state=seed, verbose=self.verbose, orientation=orientation,
                projection=self._projection(), device=self.device, **multi)
            if best is None or result[2][-1] < best[2][-1]:
                best = result
        W, H, losses, _, n_iter = best
        self.W_ = W
        self.components_ = H
        self.loss_curve_ = losses
        self.objective_history_ = losses
        self.loss_ = losses[-1] if losses else np.inf
        self.n_iter_ = n_iter
        self.reconstruction_err_ = losses[-1] if losses else np.inf
        return self

    def fit_transform(self, X, y=None):
        """Fit and return ``W_`` (no mask argument, as _base.py:145-160)."""
        self.fit(X)
        return self.W_

    def transform(self, X, mask=None):
        """Find W for new rows X with ``components_`` frozen: 50 simplex-factor updates from a
        draw of the global NumPy RNG (_base.py:162-199)."""
        check_is_fitted(self, ["components_"])
        X = self._validated(X, keep_sparse=True)
        try:
            return w_only_transform(X, self.components_, mask=mask, n_iter=50, device=self.device)
        except ValueError as e:
            self._finite_or_binary_error(X, e)

    def inverse_transform(self, W):
        """clip(W @ components_, 0, 1) (_base.py:201-210)."""
        check_is_fitted(self, ["components_"])
        W = check_array(W, dtype=np.float64)
        return np.clip(W @ self.components_, 0.0, 1.0)

    def score(self, X, mask=None):
        """Mean log-likelihood per observed entry of the reconstruction (_base.py:212-247).
        As in the reference the inner ``transform`` is called WITHOUT the mask (:235)."""
        check_is_fitted(self, ["components_"])
        X = self._validated(X, keep_sparse=True)
        H = np.asarray(self.components_, dtype=np.float64)
        try:
            return device_score(X, H, mask=mask, n_iter=50, device=self.device)
        except ValueError as e:
            self._finite_or_binary_error(X, e)

    def perplexity(self, X, mask=None):
        """exp(-score) (_base.py:249-265)."""
        return np.exp(-self.score(X, mask))


# alias kept for backwards compatibility (_base.py:269)
NBMF = NBMFMM
