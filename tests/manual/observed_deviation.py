#!/usr/bin/env python3
"""Largest deviations from the golden vectors actually observed (for the tolerance table of DESIGN.md 2): config-1
curve and factors, the masked curve, the 512 x 512 curves, by both engines.  Hand-run on a GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from nbmf_mm_amd import NBMF
G = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))
X = (np.random.default_rng(0).random((100, 500)) < 0.25).astype(np.float64)
for engine in ("1", "0"):
    os.environ["NBMF_PERSISTENT"] = engine
    g = G("config1")
    m = NBMF(6, orientation="beta-dir", alpha=1.2, beta=1.2, random_state=0, max_iter=200, tol=0).fit(X)
    loss, W, H = g["losses"], g["W"], g["H"]
    print(f"engine {'single launch' if engine == '1' else 'five kernels'}: config-1 loss rel {np.max(np.abs(np.array(m.loss_curve_) - loss) / np.abs(loss)):.2e}  "
          f"W abs {np.max(np.abs(m.W_ - W)):.2e}  H abs {np.max(np.abs(m.components_ - H)):.2e}")
g = G("midsize")
gg = np.random.default_rng(0)
Xm = (gg.random((512, 512)) < 0.25).astype(np.float64)
Mk = gg.random((512, 512)) < 0.9
m = NBMF(32, random_state=0, max_iter=500, tol=0).fit(Xm)
print(f"512 x 512 K=32 unmasked, 500 its: loss rel {np.max(np.abs(np.array(m.loss_curve_) - g['unmasked']) / np.abs(g['unmasked'])):.2e}")
m = NBMF(32, random_state=0, max_iter=300, tol=0).fit(Xm, mask=Mk)
print(f"512 x 512 K=32 masked, 300 its:   loss rel {np.max(np.abs(np.array(m.loss_curve_) - g['masked']) / np.abs(g['masked'])):.2e}")
