"""Developer tool: randomised parity sweep of predict (all four branches of predictDiag.m / predictCov.m), getPHI (all
outputs) and getPrior against the oracle.  usage: fuzz_predict.py [cases] [seed] [wide]
wide: GC/VC draw d up to 12 (beyond the register-resident kernels' d <= 10), the diagonal kinds d up to 30, k up to 10."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpz_amd
from oracle import gpz_oracle as O
from helpers import make_problem, rel

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"
bad = 0
t0 = time.time()
for c in range(cases):
    method = str(rng.choice(["GL", "VL", "GD", "VD", "GC", "VC"]))
    cov = method[1] == "C"
    d = int(rng.integers(2, 6)) if cov else int(rng.integers(1, 8))
    m = int(rng.choice([1, 2, 3, 5, 8])) if cov else int(rng.choice([1, 2, 5, 16, 17, 40]))
    k = int(rng.choice([1, 1, 2]))
    if WIDE and rng.random() < 0.5:
        d = int(rng.integers(6, 13)) if cov else int(rng.integers(8, 31))
        m = min(m, 5) if cov else m
        if rng.random() < 0.4:
            k = int(rng.integers(9, 11))
    hetero = bool(rng.random() < 0.7)
    seed = int(rng.integers(1 << 30))
    model, theta, X, Y, _, r2 = make_problem(120, d, m, k, method, hetero, seed=seed)
    method = model.method; cov = method[1] == "C"
    model.muX = r2.standard_normal(d) * 0.1; model.sdX = 1.0 + r2.random(d); model.muY = r2.standard_normal(k)
    r4 = O.GPz(theta, model, X, Y, nargout=4)
    pri = r2.random(m) + 0.2
    model.sets["best"] = {"theta": theta, "w": r4.w, "iSigma_w": r4.iSigma_w, "priors": pri / pri.sum()}
    ns = int(rng.choice([1, 2, 7, 30])) if cov else int(rng.choice([1, 5, 64, 130]))
    Xs = r2.standard_normal((ns, d))
    nanfrac = float(rng.choice([0.0, 0.3, 0.5])) if d > 1 else 0.0
    if nanfrac > 0:
        miss = r2.random((ns, d)) < nanfrac
        miss[miss.all(axis=1), int(rng.integers(0, d))] = False
        Xs[miss] = np.nan
    noisy = bool(rng.random() < 0.5)
    Psi = None
    if noisy:
        if cov:
            Psi = np.zeros((d, d, ns))
            for i in range(ns):
                B = 0.3 * r2.standard_normal((d, d)); Psi[:, :, i] = B @ B.T
        else:
            Psi = r2.gamma(1.0, 0.1, (ns, d))
    tag = f"case {c}: {method} d={d} m={m} k={k} het={int(hetero)} ns={ns} nan={nanfrac} noisy={int(noisy)} seed={seed}"
    try:
        P, G, *_ = O.unpack_theta(theta, model)
        tol = 1e-8
        if cov:
            Gm = O.expand_gamma(G, model)
            cg = max(np.linalg.cond(Gm[:, :, j].T @ Gm[:, :, j]) for j in range(Gm.shape[2]))
            tol = max(tol, 200 * cg * 2.2e-16)
            if nanfrac > 0:
                # conditioning on the observed block goes through Sigma = inv(Gamma'Gamma) AND inv(Sigma_oo): both sides lose
                # another factor of the marginal's conditioning (seen at d = 11, 12 with cond 3e5: 4 x the bound above)
                tol = max(tol, 2000 * cg * 2.2e-16)
                # the conditional covariance Sigma_uu - Sigma_uo inv(Sigma_oo) Sigma_ou cancels to ~1/cond of its terms: at cond 2e7
                # oracle and HIP path differ by 7e-5 while the two HIP routes (registers / scratch) agree to every printed digit
                tol = max(tol, min(1e-4, 0.1 * cg * cg * 2.2e-16))
        ref = O.predict_any(Xs, model, Psi=Psi)
        out = gpz_amd.predict(Xs, model, Psi=Psi)
        errs = [rel(out[i], ref[i]) for i in range(6)]
        # getPHI on the normalised inputs (all outputs) and getPrior
        Xn = (Xs - model.muX) / model.sdX
        PsiN = O.fixPsi(Psi, ns, model.sdX, method)
        rp = O.getPHI(Xn, PsiN, theta, model, None, want_N=True)
        gp = gpz_amd.getPHI(Xn, PsiN, theta, model, None, want_N=True)
        errs += [rel(gp[0], rp[0]), rel(gp[2], rp[2]), rel(gp[3], rp[3])]
        if ns >= 5:
            errs.append(rel(gpz_amd.getPrior(Xn, PsiN, theta, model), O.getPrior(Xn, PsiN, theta, model)))
        if max(errs) > tol:
            bad += 1
            print("FAIL", tag, " ".join("%.1e" % e for e in errs), "tol=%.1e" % tol)
    except Exception as e:
        bad += 1
        print("ERROR", tag, repr(e)[:300])
print(f"{cases} cases, {bad} failures, {time.time() - t0:.0f} s")
