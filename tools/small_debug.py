#!/usr/bin/env python3
"""Where an iteration of the single-launch path spends its time (workgroup 0): NBMF_SMALL_DEBUG=1 makes the
library print per-part means of the wall clock for iterations 8..62."""
import os, sys
os.environ["NBMF_SMALL_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbmf_mm_amd import _hip, _dist
for name, m, n, k in [("c1 100x500 K=6", 100, 500, 6), ("animals 50x85 K=4", 50, 85, 4), ("lastfm 1226x285 K=8", 1226, 285, 8)]:
    X = (np.random.default_rng(0).random((m, n)) < 0.25).astype(np.float64)
    W, H = _dist.global_init(m, n, k, random_state=0)
    with _hip.Context(m, n, k) as ctx:
        ctx.set_hyper(1.2, 1.2)
        ctx.upload(X)
        ctx.set_factors(W, H)
        print(name, flush=True)
        ctx.run(200, 0.0)
