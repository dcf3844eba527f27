"""Generate the prediction / pseudo-inverse fixtures under tests/golden/ (p_*.npz, s_*.npz) from the NumPy oracle.

    python tests/golden/make_golden_predict.py

p_*: predict.m with every branch of predictDiag.m / predictCov.m (rows without missing values -> predictFull /
predictNoisy, rows with missing values -> predictMissing / predictNoisyMissing, grouped by NaN pattern), plus getPrior.
s_*: inv_logdet.m on rank-deficient matrices (the truncating branch, inv_logdet.m:7-15).
Same status as make_golden.py: outputs of oracle/gpz_oracle.py frozen at generation time (the MATLAB reference cannot be
run here and ships no recorded outputs)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import gpz_oracle as O  # noqa: E402
from helpers import make_problem  # noqa: E402


def predict_case(name, method, d, m, k, ns, noisy, seed):
    model, theta, X, Y, _, rng = make_problem(200, d, m, k, method, True, seed=seed)
    model.muX = rng.standard_normal(d) * 0.1; model.sdX = 1.0 + rng.random(d); model.muY = rng.standard_normal(k)
    r4 = O.GPz(theta, model, X, Y, nargout=4)
    pri = rng.random(m) + 0.2; pri /= pri.sum()
    model.sets["best"] = {"theta": theta, "w": r4.w, "iSigma_w": r4.iSigma_w, "priors": pri}
    Xs = rng.standard_normal((ns, d))
    miss = rng.random((ns, d)) < 0.3
    miss[miss.all(axis=1), 0] = False
    miss[:3] = False                                    # some complete rows
    if d > 1:
        miss[3] = False; miss[3, 0] = True              # "first dimension missing": the unshuffle quirk of predictCov.m:266-268
    Xs[miss] = np.nan
    Psi = None
    if noisy:
        if model.method[1] == "C":
            Psi = np.zeros((d, d, ns))
            for i in range(ns):
                B = 0.3 * rng.standard_normal((d, d)); Psi[:, :, i] = B @ B.T
        else:
            Psi = rng.gamma(1.0, 0.1, (ns, d))
    mu, sigma, nu, beta_i, gamma, PHI = O.predict_any(Xs, model, Psi=Psi)
    Xn = (Xs - model.muX) / model.sdX
    prior = O.getPrior(Xn, O.fixPsi(Psi, ns, model.sdX, model.method), theta, model)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), d=d, m=m, k=k, method=model.method, theta=theta, w=r4.w,
                        iSigma_w=r4.iSigma_w, priors=pri, muX=model.muX, sdX=model.sdX, muY=model.muY, Xs=Xs,
                        has_psi=int(noisy), Psi=Psi if noisy else np.zeros(0), mu=mu, sigma=sigma, nu=nu, beta_i=beta_i,
                        gamma=gamma, PHI=PHI, prior=prior)


def pinv_case(name, m, rank, seed):
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((m, rank))
    S = B @ B.T
    S = 0.5 * (S + S.T)
    Xi, ld = O.inv_logdet(S)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), S=S, rank=rank, Xi=Xi, logdet=ld)


def main():
    seed = 500
    for method, d, m, k, ns in (("VD", 4, 9, 2, 40), ("GL", 3, 12, 1, 40), ("VC", 3, 4, 1, 10), ("GC", 4, 4, 2, 10)):
        for noisy in (False, True):
            predict_case(f"p_{method}_noisy{int(noisy)}", method, d, m, k, ns, noisy, seed)
            seed += 1
    pinv_case("s_rank5_of_12", 12, 5, 600)
    pinv_case("s_rank60_of_64", 64, 60, 601)
    print("wrote prediction and pseudo-inverse fixtures")


if __name__ == "__main__":
    main()
