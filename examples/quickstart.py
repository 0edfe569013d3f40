#!/usr/bin/env python3
"""The reference README's quick-start (README.md:60-91 of siddC/nbmf_mm), unchanged except for the import.
Needs an MI355X and the built library (`make -C nbmf_mm_amd/csrc`)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbmf_mm_amd import NBMF  # was: from nbmf_mm import NBMF

rng = np.random.default_rng(0)
X = (rng.random((100, 500)) < 0.25).astype(float)   # binary {0,1} or probabilities in [0,1]

model = NBMF(n_components=6, orientation="beta-dir", alpha=1.2, beta=1.2, random_state=0).fit(X)
W = model.W_                       # (n_samples, n_components), rows on the simplex
H = model.components_              # (n_components, n_features), entries in (0, 1)
Xhat = model.inverse_transform(W)  # probabilities in (0, 1)
print(f"fit: {model.n_iter_} iterations, loss {model.loss_:.12f}  (reference: 286 iterations, 0.537433210001)")

Y_new = (rng.random((10, 500)) < 0.25).astype(float)
W_new = model.transform(Y_new)
print("transform:", W_new.shape, "row sums", W_new.sum(axis=1).round(12)[:3])

mask = (rng.random(X.shape) < 0.9).astype(float)    # observe 90 % of the entries
model = NBMF(n_components=20).fit(X, mask=mask)
print("score (mean log-likelihood per observed entry):", model.score(X, mask=mask))
print("perplexity:", model.perplexity(X, mask=mask))

model = NBMF(n_components=6, orientation="beta-dir", projection_method="duchi", random_state=0).fit(X)
print(f"duchi projection: loss {model.loss_:.12f}")
