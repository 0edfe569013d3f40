#!/usr/bin/env python3
"""Randomised parity sweep against the oracle (hand-run on a GPU box: `python tests/manual/fuzz_vs_oracle.py [cases] [seed] [max_dim]`).

Every case draws a shape, K, data kind (binary / real), mask kind (none / bool / real weights), orientation,
hyper-parameters, eps, the projection (the reference's or the Duchi extension), an init that is in or out of the range a
fit keeps (the two variants of the sweeps), the engine (single-launch / five kernels) and the iteration
count, runs nbmf_mm_amd and the oracle on the same inputs and compares the loss curve (rtol 1e-9) and the factors (atol 1e-8).  Prints one line per failure and a summary."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from nbmf_mm_amd import nbmf_mm_solver
from oracle import nbmf_oracle as orc

def run(cases, seed, max_dim=700):
    """Returns the number of failing cases (each printed)."""
    r = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for case in range(cases):
        m = int(r.integers(1, max_dim))
        n = int(r.integers(1, max_dim))
        # (round 6: the two-state W sweep needs the swept dimension free of pad rows -- a multiple of 128: the columns of V
        #  under beta-dir, its rows under dir-beta -- which a uniform draw almost never hits)
        if r.random() < 0.15:
            n = 128 * int(r.integers(1, max(2, max_dim // 128 + 1)))
        if r.random() < 0.15:
            m = 128 * int(r.integers(1, max(2, max_dim // 128 + 1)))
        k = int(r.choice([1, 2, 3, 5, 8, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 129, 140]))
        real = r.random() < 0.25
        Y = r.random((m, n)) if real else (r.random((m, n)) < r.uniform(0.05, 0.9)).astype(np.float64)
        as_f32 = real and r.random() < 0.33
        if as_f32:
            Y = Y.astype(np.float32).astype(np.float64)   # (the values the float32 array will hold, for the oracle)
        mk = r.choice(["none", "bool", "weights"], p=[0.4, 0.45, 0.15])
        mask = None if mk == "none" else ((r.random((m, n)) < r.uniform(0.3, 0.99)) if mk == "bool" else r.random((m, n)))
        kw = dict(max_iter=int(r.integers(1, 25)), tol=0, mask=mask, alpha=float(r.uniform(1.0, 2.0)), beta=float(r.uniform(1.0, 2.0)),
                  orientation=str(r.choice(["beta-dir", "dir-beta"])))
        eps_kind = r.choice(["default", "small", "tiny"], p=[0.8, 0.1, 0.1])
        if eps_kind == "small":
            kw["eps"] = 1e-20
        elif eps_kind == "tiny":
            kw["eps"] = 1e-80
        init = r.choice(["seed", "in_range", "H_high", "W_free"], p=[0.4, 0.3, 0.15, 0.15])
        if init == "seed":
            kw["random_state"] = int(r.integers(0, 1000))
        else:
            # (custom inits are given in the orientation's own shapes: W (m, k), H (k, n))
            W0 = r.uniform(0.05, 0.95, (m, k))
            H0 = r.uniform(0.05, 0.95, (k, n))
            if init == "H_high":
                (H0 if kw["orientation"] == "beta-dir" else W0)[...] = r.uniform(0.05, 1.5, (k, n) if kw["orientation"] == "beta-dir" else (m, k))
            if init == "W_free":
                W0 *= 3.0
            kw["W_init"], kw["H_init"] = W0, H0
        os.environ["NBMF_PERSISTENT"] = str(r.choice(["0", "1"]))
        duchi = bool(r.random() < 0.25) and init in ("seed", "in_range")   # (the extension: Euclidean projection, README.md:27-35)
        with np.errstate(all="ignore"):
            Wr, Hr, lr, _, _ = orc.solve(Y, k, step=orc.mm_step_duchi if duchi else None, **kw)
        # (round 4: binary data is handed over as bool / uint8 in a third of the cases -- one byte per entry, nbmf_upload_v;
        #  round 5: as a scipy CSR matrix, Fortran-ordered or as a strided view in some; real data as float32 in a third --
        #  the oracle then works on the float32 values, exactly representable -- and the mask as uint8 / float32 / CSR)
        Y_in = Y
        form = "f64"
        if not real:
            form = str(r.choice(["f64", "u8", "bool", "csr", "fortran", "strided"]))
            if form == "u8":
                Y_in = Y.astype(np.uint8)
            elif form == "bool":
                Y_in = Y.astype(bool)
            elif form == "csr":
                import scipy.sparse as sp
                Y_in = sp.csr_matrix(Y)
        else:
            form = "f32" if as_f32 else str(r.choice(["f64", "fortran", "strided"]))
            if form == "f32":
                Y_in = Y.astype(np.float32)
        if form == "fortran":
            Y_in = np.asfortranarray(Y)
        elif form == "strided":
            big = np.zeros((2 * m, 2 * n), dtype=Y.dtype)
            big[::2, ::2] = Y
            Y_in = big[::2, ::2]
        mask_in = mask
        mform = "as-is"
        if mask is not None:
            mform = str(r.choice(["as-is", "u8", "f32", "csr"] if mk == "bool" else ["as-is", "fortran"]))
            if mform == "u8":
                mask_in = mask.astype(np.uint8)
            elif mform == "f32":
                mask_in = mask.astype(np.float32)
            elif mform == "csr":
                import scipy.sparse as sp
                mask_in = sp.csr_matrix(mask.astype(np.float64))
            elif mform == "fortran":
                mask_in = np.asfortranarray(mask)
        kw_in = dict(kw, mask=mask_in)
        try:
            W, H, l, _, _ = nbmf_mm_solver(Y_in, k, projection="duchi" if duchi else "normalize", **kw_in)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"case {case}: EXCEPTION {e!r}  m={m} n={n} k={k} real={real} mask={mk} init={init} eps={eps_kind} {kw['orientation']} data as {form} mask as {mform}")
            continue
        lr = np.asarray(lr)
        l = np.asarray(l)
        ok = np.array_equal(np.isnan(l), np.isnan(lr))
        fin = ~np.isnan(lr)
        if ok and fin.any():
            ok = np.allclose(l[fin], lr[fin], rtol=1e-9, atol=0)
        if ok and np.all(np.isfinite(Wr)) and np.all(np.isfinite(Hr)) and fin.all():
            ok = np.allclose(W, Wr, rtol=0, atol=1e-8) and np.allclose(H, Hr, rtol=0, atol=1e-8)
        if not ok:
            bad += 1
            dl = np.max(np.abs(l[fin] / lr[fin] - 1)) if fin.any() else float("nan")
            dW = np.max(np.abs(W - Wr)) if np.all(np.isfinite(Wr)) else float("nan")
            print(f"case {case}: MISMATCH rel loss {dl:.2e} max|dW| {dW:.2e}  m={m} n={n} k={k} real={real} mask={mk} init={init} eps={eps_kind} "
                  f"{kw['orientation']} its={kw['max_iter']} engine={os.environ['NBMF_PERSISTENT']} duchi={duchi} data as {form} mask as {mform}", flush=True)
        if case % 50 == 49:
            print(f"  ... {case + 1} cases, {bad} failures, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    return bad, time.time() - t0


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n_bad, secs = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 700)
    print(f"{n_cases} cases, {n_bad} failures, {secs:.0f} s")
    sys.exit(1 if n_bad else 0)
