#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE package (TEST INFRASTRUCTURE).

Run in the build container only:  python oracle/make_golden.py
It imports ``nbmf_mm`` from /root/reference/src (never copied into this repo) and stores
inputs (when not regenerable from a seed) and the reference's outputs as small fixtures.
The case list is SURVEY.md §8c items 1-9, plus (item 10, round 4) storage paths and input kinds beyond them and (item 11, round 5) real-valued data at K = 16 / 32 / 64 and columns without a one under a flat prior, and (item 12) twenty random fits through the estimator, and (item 13, round 6) the held-out perplexity of the reference's experiment driver.  The GPU box never runs this script.
`python oracle/make_golden.py heldout` writes item 13's file only.
"""
import os
import sys

import numpy as np

REF_SRC = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def reference_compute_perplexity():
    """`compute_perplexity` of the reference's experiment script (examples/reproduce_magron2022.py:40-47) WITHOUT importing
    the script (its module level needs pyreadr / pandas and creates directories): the one function definition is taken
    out of the parsed source and compiled here, in the build container, with NumPy as its only global.  Nothing of the
    text is stored anywhere."""
    import ast
    path = "/root/reference/examples/reproduce_magron2022.py"
    tree = ast.parse(open(path).read(), filename=path)
    fn = [node for node in tree.body if isinstance(node, ast.FunctionDef) and node.name == "compute_perplexity"]
    if len(fn) != 1:
        sys.exit("compute_perplexity not found in " + path)
    ns = {"np": np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    return ns["compute_perplexity"]


def heldout():
    """13. held-out perplexity (round 6): the reference's estimator fitted on the training entries of a seeded 40 x 70
    problem, then the reference driver's own compute_perplexity of W_ @ components_ on the validation and the test
    entries (strict masks), on a real-valued weight mask, and with no mask at all."""
    sys.path.insert(0, REF_SRC)
    from nbmf_mm import NBMF
    ref_perp = reference_compute_perplexity()
    g = np.random.default_rng(1313)
    Y = (g.random((40, 70)) < 0.3).astype(np.float64)
    u = g.random((40, 70))
    train, val, test = u < 0.7, (u >= 0.7) & (u < 0.85), u >= 0.85
    weights = g.uniform(0.0, 1.0, (40, 70)) * (u >= 0.7)
    mdl = NBMF(n_components=5, orientation="beta-dir", alpha=1.2, beta=1.2, max_iter=60, tol=1e-5, random_state=12345).fit(Y, mask=train)
    Y_hat = mdl.W_ @ mdl.components_
    out = dict(Y=Y.astype(np.uint8), train=train, val=val, test=test, weights=weights, W=mdl.W_, H=mdl.components_,
               losses=np.array(mdl.loss_curve_), n_iter=np.array(mdl.n_iter_),
               perp_val=np.array(ref_perp(Y, Y_hat, val)), perp_test=np.array(ref_perp(Y, Y_hat, test)),
               perp_val_float=np.array(ref_perp(Y, Y_hat, val.astype(np.float64))),
               perp_weights=np.array(ref_perp(Y, Y_hat, weights)), perp_nomask=np.array(ref_perp(Y, Y_hat)),
               perp_eps=np.array(ref_perp(Y, Y_hat, test, eps=1e-6)))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "heldout.npz"), **out)
    print("heldout.npz:", {k: float(v) for k, v in out.items() if k.startswith("perp")}, "n_iter", int(out["n_iter"]))


def main():
    if not os.path.isdir(REF_SRC):
        sys.exit("reference not present: goldens can only be regenerated in the build container")
    sys.path.insert(0, REF_SRC)
    import nbmf_mm  # noqa: F401  (the reference)
    from nbmf_mm import NBMF, nbmf_mm_solver
    from nbmf_mm._solver import nbmf_mm_update_beta_dir
    from nbmf_mm._utils import generate_synthetic_binary_data

    os.makedirs(OUT, exist_ok=True)

    # 1. one-step vectors ------------------------------------------------------------
    one = {}
    idx = 0
    for (m, n, k) in [(20, 30, 4), (64, 48, 5)]:
        for mkind in ["none", "float", "bool"]:
            for (al, be) in [(1.2, 1.2), (1.1, 1.3), (0.5, 2.0)]:
                g = np.random.default_rng(1000 + idx)
                Y = (g.random((m, n)) < 0.3).astype(np.float64)
                W = g.uniform(0.1, 0.9, (k, m))
                W = W / W.sum(axis=0, keepdims=True)
                H = g.uniform(0.1, 0.9, (k, n))
                if mkind == "none":
                    mask = None
                elif mkind == "float":
                    mask = (g.random((m, n)) < 0.8).astype(np.float64)
                else:
                    mask = g.random((m, n)) < 0.8
                Wn, Hn = nbmf_mm_update_beta_dir(Y, W, H, mask, al, be, 1e-8)
                p = f"c{idx}_"
                one[p + "Y"], one[p + "W"], one[p + "H"] = Y, W, H
                one[p + "mask"] = np.zeros((0,)) if mask is None else mask
                one[p + "ab"] = np.array([al, be])
                one[p + "W_new"], one[p + "H_new"] = Wn, Hn
                idx += 1
    # real-valued Y and a weight (non-binary) mask
    g = np.random.default_rng(77)
    Y = g.random((24, 40))
    W = g.uniform(0.1, 0.9, (3, 24)); W /= W.sum(axis=0, keepdims=True)
    H = g.uniform(0.1, 0.9, (3, 40))
    mask = g.random((24, 40))
    Wn, Hn = nbmf_mm_update_beta_dir(Y, W, H, mask, 1.2, 1.2, 1e-8)
    p = f"c{idx}_"
    one[p + "Y"], one[p + "W"], one[p + "H"], one[p + "mask"] = Y, W, H, mask
    one[p + "ab"] = np.array([1.2, 1.2]); one[p + "W_new"], one[p + "H_new"] = Wn, Hn
    idx += 1
    one["n_cases"] = np.array(idx)
    np.savez_compressed(os.path.join(OUT, "one_step.npz"), **one)

    # 2. config-1 curve ----------------------------------------------------------------
    X = (np.random.default_rng(0).random((100, 500)) < 0.25).astype(np.float64)
    mdl = NBMF(n_components=6, orientation="beta-dir", alpha=1.2, beta=1.2, random_state=0,
               max_iter=200, tol=0).fit(X)
    mdl2 = NBMF(n_components=6, orientation="beta-dir", alpha=1.2, beta=1.2, random_state=0).fit(X)
    np.savez_compressed(os.path.join(OUT, "config1.npz"), losses=np.array(mdl.loss_curve_), W=mdl.W_,
                        H=mdl.components_, n_iter=np.array(mdl.n_iter_),
                        default_n_iter=np.array(mdl2.n_iter_), default_loss=np.array(mdl2.loss_))

    # 3. dir-beta --------------------------------------------------------------------
    mdl = NBMF(n_components=6, orientation="dir-beta", alpha=1.2, beta=1.2, random_state=0,
               max_iter=50, tol=0).fit(X)
    mdlT = NBMF(n_components=6, orientation="beta-dir", alpha=1.2, beta=1.2, random_state=0,
                max_iter=50, tol=0).fit(X.T)
    np.savez_compressed(os.path.join(OUT, "dir_beta.npz"), losses=np.array(mdl.loss_curve_), W=mdl.W_,
                        H=mdl.components_, WT_of_transposed=mdlT.components_.T)

    # 4. masked (float and bool masks) --------------------------------------------------
    mask = np.random.default_rng(1).random(X.shape) < 0.9
    mf = NBMF(n_components=6, alpha=1.2, beta=1.2, random_state=0, max_iter=100, tol=0).fit(
        X, mask=mask.astype(np.float64))
    mb = NBMF(n_components=6, alpha=1.2, beta=1.2, random_state=0, max_iter=100, tol=0).fit(X, mask=mask)
    np.savez_compressed(os.path.join(OUT, "masked.npz"), losses_float=np.array(mf.loss_curve_),
                        losses_bool=np.array(mb.loss_curve_), W=mf.W_, H=mf.components_)

    # 5. real-valued V ---------------------------------------------------------------
    Xr = np.random.default_rng(3).random((50, 30))
    mr = NBMF(n_components=5, random_state=1, max_iter=30, tol=0).fit(Xr)
    np.savez_compressed(os.path.join(OUT, "real_valued.npz"), losses=np.array(mr.loss_curve_), W=mr.W_,
                        H=mr.components_)

    # 6. custom init (tests/test_strict_parity_optional.py:11-29 recipe) ---------------------
    r = np.random.default_rng(123)
    M, N, K = 20, 25, 4
    Yc = (r.random((M, N)) < 0.3).astype(float)
    W0 = r.gamma(shape=1.0, scale=1.0, size=(M, K)); W0 /= W0.sum(axis=1, keepdims=True)
    H0 = np.clip(r.random((K, N)), 1e-6, 1 - 1e-6)
    mc = NBMF(n_components=K, orientation="beta-dir", alpha=1.2, beta=1.2, random_state=123, max_iter=50,
              tol=1e-8, W_init=W0, H_init=H0).fit(Yc)
    md = NBMF(n_components=K, orientation="dir-beta", alpha=1.2, beta=1.2, random_state=123, max_iter=20,
              tol=0, W_init=np.clip(r.random((M, K)), 1e-3, 1 - 1e-3),
              H_init=r.uniform(0.1, 0.9, (K, N)))
    Wd0, Hd0 = md.W_init.copy(), md.H_init.copy()
    md.fit(Yc)
    np.savez_compressed(os.path.join(OUT, "custom_init.npz"), Y=Yc, W0=W0, H0=H0, losses=np.array(mc.loss_curve_),
                        W=mc.W_, H=mc.components_, n_iter=np.array(mc.n_iter_), Wd0=Wd0, Hd0=Hd0,
                        d_losses=np.array(md.loss_curve_), dW=md.W_, dH=md.components_)

    # 7. stop rule -------------------------------------------------------------------
    Xs, _, _ = generate_synthetic_binary_data(50, 30, 5, random_state=42)
    hi = NBMF(n_components=5, tol=0.1, max_iter=1000, random_state=42).fit(Xs)
    lo = NBMF(n_components=5, tol=1e-8, max_iter=1000, random_state=42).fit(Xs)
    np.savez_compressed(os.path.join(OUT, "stop_rule.npz"), X=Xs, n_iter_hi=np.array(hi.n_iter_),
                        n_iter_lo=np.array(lo.n_iter_), loss_hi=np.array(hi.loss_), loss_lo=np.array(lo.loss_),
                        losses_lo=np.array(lo.loss_curve_))

    # 8. transform / score / perplexity -----------------------------------------------------
    mt = NBMF(n_components=6, alpha=1.2, beta=1.2, random_state=0, max_iter=60, tol=0).fit(X, mask=mask)
    Xn = (np.random.default_rng(9).random((10, 500)) < 0.25).astype(np.float64)
    np.random.seed(5)
    Wt = mt.transform(Xn)
    np.random.seed(5)
    Wtm = mt.transform(X, mask=mask.astype(np.float64))
    np.random.seed(6)
    sc = mt.score(X, mask=mask.astype(np.float64))
    np.random.seed(6)
    sc_nomask = mt.score(X)
    np.random.seed(6)
    pp = mt.perplexity(X, mask=mask.astype(np.float64))
    np.savez_compressed(os.path.join(OUT, "transform.npz"), H=mt.components_, Wfit=mt.W_, W_new=Wt, W_masked=Wtm,
                        score=np.array(sc), score_nomask=np.array(sc_nomask), perplexity=np.array(pp))

    # 9. mid-size curves for GPU parity (loss curves only) ----------------------------------
    g = np.random.default_rng(0)
    Xm = (g.random((512, 512)) < 0.25).astype(np.float64)
    Mm = (g.random((512, 512)) < 0.9).astype(np.float64)
    _, _, l_un, _, _ = nbmf_mm_solver(Xm, 32, max_iter=500, tol=0, random_state=0)
    _, _, l_mk, _, _ = nbmf_mm_solver(Xm, 32, max_iter=300, tol=0, random_state=0, mask=Mm)
    _, _, l_db, _, _ = nbmf_mm_solver(Xm[:, :384], 64, max_iter=100, tol=0, random_state=0, orientation="dir-beta",
                                      mask=Mm[:, :384])
    g2 = np.random.default_rng(4)
    Xrv = g2.random((300, 200))
    Wts = g2.random((300, 200))
    _, _, l_rv, _, _ = nbmf_mm_solver(Xrv, 16, max_iter=60, tol=0, random_state=2, mask=Wts)
    np.savez_compressed(os.path.join(OUT, "midsize.npz"), unmasked=np.array(l_un), masked=np.array(l_mk),
                        dir_beta_masked=np.array(l_db), real_weighted=np.array(l_rv))
    # 10. round-4 additions: the reference's outputs for the storage paths and input kinds the round-4 work touched --
    #     real-valued data with real weights / with a bool mask, both orientations, FACTORS included (item 9 keeps curves
    #     only); scipy CSR data with a CSR mask, bool data with an integer mask, float32 data (tests/test_api.py:111-123,
    #     tests/test_public_api.py:125-134: the reference converts them all at _base.py:83); a row nobody observes (0 / 0 in
    #     the simplex factor, NaN from then on: _solver.py:57); transform after a dir-beta fit (always the simplex-W form).
    import scipy.sparse as sp
    g3 = np.random.default_rng(41)
    ex = {}
    Xq = g3.random((90, 130))
    Wq = g3.random((90, 130))
    Bq = g3.random((90, 130)) < 0.8
    for name, orient, mk in (("rw_bd", "beta-dir", Wq), ("rw_db", "dir-beta", Wq), ("rb_bd", "beta-dir", Bq), ("rb_db", "dir-beta", Bq)):
        m10 = NBMF(n_components=7, alpha=1.3, beta=1.1, random_state=3, max_iter=25, tol=0, orientation=orient).fit(Xq, mask=mk)
        ex[name + "_losses"], ex[name + "_W"], ex[name + "_H"] = np.array(m10.loss_curve_), m10.W_, m10.components_
    Xb = (g3.random((70, 110)) < 0.2)
    Mb = (g3.random((70, 110)) < 0.85)
    base = NBMF(n_components=5, random_state=4, max_iter=20, tol=0).fit(Xb.astype(np.float64), mask=Mb.astype(np.float64))
    ex["kinds_losses"], ex["kinds_W"], ex["kinds_H"] = np.array(base.loss_curve_), base.W_, base.components_
    for kind, Xv, mv in (("csr", sp.csr_matrix(Xb.astype(np.float64)), sp.csr_matrix(Mb.astype(np.float64))),
                         ("bool_int", Xb, Mb.astype(np.int32)), ("f32", Xb.astype(np.float32), Mb.astype(np.float32))):
        mk10 = NBMF(n_components=5, random_state=4, max_iter=20, tol=0).fit(Xv, mask=mv)
        ex["kinds_same_" + kind] = np.array(np.array_equal(mk10.loss_curve_, base.loss_curve_) and np.array_equal(mk10.W_, base.W_))
    Mn = Mb.astype(np.float64).copy()
    Mn[9, :] = 0.0
    with np.errstate(all="ignore"):
        mn = NBMF(n_components=5, random_state=4, max_iter=6, tol=0).fit(Xb.astype(np.float64), mask=Mn)
    ex["nanrow_losses"], ex["nanrow_W"] = np.array(mn.loss_curve_), mn.W_
    md = NBMF(n_components=5, random_state=4, max_iter=30, tol=0, orientation="dir-beta").fit(Xb.astype(np.float64))
    np.random.seed(8)
    ex["dirbeta_transform"] = md.transform(Xb[:12].astype(np.float64))
    ex["dirbeta_H"] = md.components_
    np.savez_compressed(os.path.join(OUT, "round4.npz"), **ex)

    # 11. round-5 additions: real-valued data at the component counts whose sweeps round 5 touched -- K = 16 (one 16-block
    #     per strip: the shared reciprocal there is new), K = 32, K = 64 -- plain, with a bool mask under dir-beta, with real
    #     weights; factors for the two small K, curves for all.  And the one-step update of a matrix with columns nobody has
    #     a one in under a flat prior (alpha = 1: the numerator H * P1 + a is an exact 0 there, the update lands on the clip).
    g5 = np.random.default_rng(55)
    r5 = {}
    Xr5 = g5.random((150, 170))
    Br5 = g5.random((150, 170)) < 0.85
    Wt5 = g5.random((150, 170))
    for name, K5, orient, mk, its in (("k16_plain", 16, "beta-dir", None, 40), ("k16_mask_db", 16, "dir-beta", Br5, 40),
                                      ("k32_weights", 32, "beta-dir", Wt5, 30), ("k64_mask", 64, "beta-dir", Br5, 20)):
        m11 = NBMF(n_components=K5, alpha=1.2, beta=1.4, random_state=9, max_iter=its, tol=0, orientation=orient).fit(Xr5, mask=mk)
        r5[name + "_losses"] = np.array(m11.loss_curve_)
        if K5 <= 32:
            r5[name + "_W"], r5[name + "_H"] = m11.W_, m11.components_
    Yz = (g5.random((96, 70)) < 0.3).astype(np.float64)
    Yz[:, 5] = 0.0
    Yz[:, 11] = 0.0
    Wz = g5.uniform(0.1, 0.9, (7, 96))
    Wz /= Wz.sum(axis=0, keepdims=True)
    Hz = g5.uniform(0.1, 0.9, (7, 70))
    Mz = (g5.random((96, 70)) < 0.8).astype(np.float64)
    r5["zc_Y"], r5["zc_W"], r5["zc_H"], r5["zc_mask"] = Yz, Wz, Hz, Mz
    for tag, mk in (("plain", None), ("masked", Mz)):
        Wn, Hn = nbmf_mm_update_beta_dir(Yz, Wz, Hz, mk, 1.0, 1.3, 1e-8)
        r5["zc_W_new_" + tag], r5["zc_H_new_" + tag] = Wn, Hn
    np.savez_compressed(os.path.join(OUT, "round5.npz"), **r5)

    # 12. twenty random fits THROUGH THE ESTIMATOR (round 5): shapes 3..60, K 1..12, binary / real-valued data handed over as
    #     float64 / int / bool / float32 / CSR, no mask / bool / 0-1 float / real weights, every orientation alias of
    #     _base.py:126-136, priors in [1, 3], seeded or custom inits (one or both), a stop rule that fires or not.  Inputs
    #     and parameters are stored with the reference's W_, components_, loss_curve_, n_iter_.
    import json
    import scipy.sparse as sp
    g12 = np.random.default_rng(1212)
    aliases = ["beta-dir", "dir-beta", "Beta-Dir", "Dir-Beta", "Dir Beta", "binary ICA", "Binary ICA", "bICA", "Aspect Bernoulli"]
    est = {"n_cases": np.array(20)}
    for i in range(20):
        m12, n12, k12 = int(g12.integers(3, 61)), int(g12.integers(3, 61)), int(g12.integers(1, 13))
        real = bool(g12.random() < 0.35)
        X12 = g12.random((m12, n12)) if real else (g12.random((m12, n12)) < g12.uniform(0.1, 0.8)).astype(np.float64)
        form = str(g12.choice(["f64", "f32"] if real else ["f64", "int", "bool", "f32", "csr"]))
        if form == "f32":
            X12 = X12.astype(np.float32).astype(np.float64)
        mk = str(g12.choice(["none", "bool", "float01", "weights"], p=[0.35, 0.3, 0.15, 0.2]))
        M12 = {"none": None, "bool": g12.random((m12, n12)) < 0.8, "float01": (g12.random((m12, n12)) < 0.7).astype(np.float64),
               "weights": g12.uniform(0.1, 1.0, (m12, n12))}[mk]
        par = dict(n_components=k12, alpha=float(g12.uniform(1.0, 3.0)), beta=float(g12.uniform(1.0, 3.0)),
                   max_iter=int(g12.integers(3, 80)), tol=float(g12.choice([0.0, 1e-4, 1e-3])), random_state=int(g12.integers(0, 10000)),
                   orientation=str(g12.choice(aliases)))
        init = str(g12.choice(["seed", "both", "W", "H"], p=[0.5, 0.25, 0.125, 0.125]))
        W0 = g12.uniform(0.05, 0.95, (m12, k12)) if init in ("both", "W") else None
        H0 = g12.uniform(0.05, 0.95, (k12, n12)) if init in ("both", "H") else None
        Xin = {"f64": X12, "int": X12.astype(np.int64), "bool": X12.astype(bool), "f32": X12.astype(np.float32),
               "csr": sp.csr_matrix(X12)}[form]
        mdl12 = NBMF(W_init=W0, H_init=H0, **par).fit(Xin, mask=M12)
        pre = f"e{i}_"
        est[pre + "params"] = np.array(json.dumps(dict(par, form=form, mask=mk, init=init)))
        est[pre + "X"] = X12 if real else X12.astype(np.uint8)
        est[pre + "mask"] = np.zeros(0) if M12 is None else (M12 if mk != "float01" else M12.astype(np.uint8))
        est[pre + "W0"] = np.zeros(0) if W0 is None else W0
        est[pre + "H0"] = np.zeros(0) if H0 is None else H0
        est[pre + "W"], est[pre + "H"] = mdl12.W_, mdl12.components_
        est[pre + "losses"], est[pre + "n_iter"] = np.array(mdl12.loss_curve_), np.array(mdl12.n_iter_)
        est[pre + "orientation_after"] = np.array(mdl12.orientation)
    np.savez_compressed(os.path.join(OUT, "estimator_random.npz"), **est)

    heldout()

    print("golden fixtures written to", os.path.normpath(OUT))
    for f in sorted(os.listdir(OUT)):
        print("  %-20s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    if sys.argv[1:] == ["heldout"]:
        if not os.path.isdir(REF_SRC):
            sys.exit("reference not present: goldens can only be regenerated in the build container")
        heldout()
    else:
        main()
