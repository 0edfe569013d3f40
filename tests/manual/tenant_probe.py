"""Diagnosis (round 6): a fit at configs[4]'s shape while a host thread keeps launching nbmf_selftest_mfma_peak on the null
stream -- the first form of the second-tenant test hung for 400 s.  Dumps every thread's Python stack after 50 s."""
import faulthandler, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nbmf_mm_amd import _hip
faulthandler.dump_traceback_later(110, repeat=False, file=sys.stderr, exit=True)
m, n, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (17000, 360000, 128)))
mode = sys.argv[4] if len(sys.argv) > 4 else "thread"
with _hip.Context(m, n, k) as ctx:
    ctx.set_hyper(1.2, 1.2, 1e-8)
    ctx.generate(seed=5, density=0.05, observed=0.9)
    g = np.random.default_rng(2)
    W = g.uniform(0.1, 0.9, (k, m)); W /= W.sum(axis=0, keepdims=True)
    H = g.uniform(0.1, 0.9, (k, n))
    def fit(tag):
        ctx.set_factors(W, H)
        t0 = time.perf_counter(); l, _ = ctx.run(6, 0.0); dt = time.perf_counter() - t0
        print(f"{tag}: {dt:.3f} s, last loss {l[-1]:.12f}", flush=True)
        return l
    fit("warm-up"); a = fit("alone")
    stop, rows = threading.Event(), []
    def tenant():
        while not stop.is_set():
            t0 = time.perf_counter(); p = _hip.mfma_peak(0, 300.0)
            rows.append((round(t0, 2), round(time.perf_counter() - t0, 2), round(p["cycles_per_mfma_at_2p4GHz"], 1)))
            print("tenant call", rows[-1], flush=True)
    print("tenant alone:", _hip.mfma_peak(0, 300.0), flush=True)
    th = threading.Thread(target=tenant, daemon=True); th.start()
    time.sleep(1.0)
    b = fit("beside the tenant")
    stop.set(); th.join(30)
    print("same bits:", bool((np.array(a) == np.array(b)).all()), flush=True)
