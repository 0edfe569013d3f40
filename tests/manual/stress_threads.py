#!/usr/bin/env python3
"""Many host threads, one device (hand-run on a GPU box: `python tests/manual/stress_threads.py [threads] [fits per thread] [seed]`).

A grid search parallelised by the caller (`joblib` with the threading backend around `NBMF(...).fit`, as scikit-learn's
`GridSearchCV(n_jobs=...)` would do with the reference's estimator) means contexts created, used and destroyed from several
host threads at once: the library's block pool, stream pool, per-device tables and per-thread error strings are shared
state.  Every thread runs its own list of random fits (both engines, every storage path, evaluation calls, an upload that
fails in between); the same lists are then run one after the other on one thread: every result must be the same BITS, and
every failure the same message."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from nbmf_mm_amd import _hip


def one_list(seed, count):
    r = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        m, n = int(r.integers(1, 600)), int(r.integers(1, 600))
        k = int(r.choice([1, 4, 6, 10, 16, 32, 40, 64, 128]))
        real = r.random() < 0.3
        Y = r.random((m, n)) if real else (r.random((m, n)) < 0.3).astype(np.float64)
        mk = r.choice(["none", "bool", "weights"], p=[0.4, 0.4, 0.2])
        mask = None if mk == "none" else ((r.random((m, n)) < 0.85) if mk == "bool" else r.uniform(0.1, 1.0, (m, n)))
        W0 = r.uniform(0.05, 0.95, (k, m))
        W0 /= W0.sum(axis=0, keepdims=True)
        H0 = r.uniform(0.05, 0.95, (k, n))
        iters = int(r.integers(1, 30))
        bad_upload = r.random() < 0.15
        try:
            with _hip.Context(m, n, k) as ctx:
                ctx.set_hyper(float(r.uniform(1, 2)), float(r.uniform(1, 2)), 1e-8, int(r.random() < 0.25))
                if bad_upload:
                    try:
                        ctx.upload(np.full((m, n), 2.0), None)          # values outside [0, 1]: refused
                        out.append("no refusal")
                    except ValueError as e:
                        out.append(("refused", str(e)))
                ctx.upload(Y, mask)
                ctx.set_factors(W0, H0)
                losses, n_iter = ctx.run(iters, 1e-6)
                W, H = ctx.get_factors()
                out.append((losses.tobytes(), n_iter, W.tobytes(), H.tobytes(), repr(ctx.loss()), repr(ctx.loglik(True))))   # (repr: a NaN equals itself)
        except Exception as e:   # noqa: BLE001
            out.append(("EXCEPTION", repr(e)))
    return out


def main(threads, count, seed):
    t0 = time.time()
    got = [None] * threads

    def body(i):
        got[i] = one_list(seed + i, count)

    ts = [threading.Thread(target=body, args=(i,)) for i in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    t1 = time.time()
    want = [one_list(seed + i, count) for i in range(threads)]
    bad = soft = 0
    for i in range(threads):
        for j, (a, b) in enumerate(zip(got[i], want[i])):
            if a != b:
                bad += 1
                what = a[1] if isinstance(a, tuple) and a[0] == "EXCEPTION" else "different bits"
                if len(a) == 6 and len(b) == 6 and a[1] == b[1]:
                    d = [float(np.max(np.abs(np.frombuffer(a[q]) - np.frombuffer(b[q])))) if len(a[q]) == len(b[q]) and len(a[q]) else -1.0 for q in (0, 2, 3)]
                    what += f": max |difference| losses {d[0]:.2e} W {d[1]:.2e} H {d[2]:.2e} over {a[1]} iterations (loss {a[4]!r} / {b[4]!r})"
                    soft += all(0 <= x <= 1e-12 for x in d)
                print(f"thread {i} fit {j}: {what}", flush=True)
    exc = sum(1 for lst in want for a in lst if isinstance(a, tuple) and a[0] == "EXCEPTION")
    print(f"{threads} threads x {count} fits: {bad} differences from the sequential run, {soft} of them <= 1e-12 (the other engine: a "
          f"single-launch fit that gave up under the other threads' load and was redone by the five kernels); {exc} exceptions in the "
          f"sequential run; engine statistics (single-launch fits, launches that gave up, declined, five-kernel runs) {_hip.engine_stats()}; "
          f"threaded {t1 - t0:.1f} s, sequential {time.time() - t1:.1f} s")
    return (bad - soft) + exc


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    sys.exit(1 if main(a[0] if a else 8, a[1] if len(a) > 1 else 60, a[2] if len(a) > 2 else 0) else 0)
