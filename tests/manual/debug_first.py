import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nbmf_mm_amd import _hip
from oracle import nbmf_oracle as orc
print("rcp err", _hip.selftest_rcp(1 << 16))
for (m, n, k, masked, real) in [(20, 30, 4, False, False), (20, 30, 4, True, False), (24, 40, 3, True, True), (130, 257, 17, False, False), (64, 300, 100, True, False), (300, 200, 64, False, False)]:
    r = np.random.default_rng(1)
    Y = r.random((m, n)) if real else (r.random((m, n)) < 0.3).astype(float)
    mask = None
    if masked:
        mask = r.random((m, n)) if real else (r.random((m, n)) < 0.8).astype(float)
    W = r.uniform(0.1, 0.9, (k, m)); W /= W.sum(axis=0, keepdims=True)
    H = r.uniform(0.1, 0.9, (k, n))
    with _hip.Context(m, n, k) as ctx:
        ctx.set_hyper(1.2, 1.3)
        ctx.upload(Y, mask=mask)
        ctx.set_factors(W, H)
        l0 = ctx.loss()
        losses, nit = ctx.run(3, 0.0)
        Wn, Hn = ctx.get_factors()
        binp = ctx.binary_path
    Wr, Hr = W, H
    lr = []
    for _ in range(3):
        Wr, Hr = orc.mm_step(Y, Wr, Hr, mask, 1.2, 1.3)
        lr.append(orc.mm_loss(Y, Wr, Hr, mask, 1.2, 1.3))
    print(f"{m}x{n} k={k} masked={masked} real={real} bin={binp}: loss0 {l0:.15f} ref {orc.mm_loss(Y, W, H, mask, 1.2, 1.3):.15f} | "
          f"dH {np.abs(Hn - Hr).max():.2e} dW {np.abs(Wn - Wr).max():.2e} dloss {np.abs(np.array(lr) - losses).max():.2e} nit {nit}")
