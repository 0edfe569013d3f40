"""Diagnosis (round 6): the loss assembly's give-up path while other workgroups of the sweep still run (NBMF_PASSFIN_FAULT=2:
the assembling workgroup does not wait at all), against the same run without the fault."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nbmf_mm_amd import _hip
os.environ["NBMF_PERSISTENT"] = "0"
m, n, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 8192, 64)))
g = np.random.default_rng(1)
with _hip.Context(m, n, k) as ctx:
    ctx.set_hyper(1.2, 1.2, 1e-8)
    ctx.generate(seed=5, density=0.05, observed=0.9)
    W = g.uniform(0.1, 0.9, (k, m)); W /= W.sum(axis=0, keepdims=True)
    H = g.uniform(0.1, 0.9, (k, n))
    os.environ.pop("NBMF_PASSFIN_FAULT", None)
    ctx.set_factors(W, H); good, _ = ctx.run(6, 0.0); Hg = ctx.get_factors()[1].copy()
    for rep in range(4):
        os.environ["NBMF_PASSFIN_FAULT"] = "2"
        before = _hip.variant_stats()[2]
        ctx.set_factors(W, H); got, _ = ctx.run(6, 0.0); Hh = ctx.get_factors()[1]
        print("resumed", _hip.variant_stats()[2] - before, "losses equal", bool((got == good).all()), "H equal", bool((Hh == Hg).all()))
        if not (got == good).all():
            print(" good", good); print(" got ", got)
