"""Generate the golden fixtures under tests/golden/ from the NumPy oracle.

    python tests/golden/make_golden.py

The reference (MATLAB) cannot be executed in the build image and ships no recorded outputs,
so these vectors are outputs of oracle/gpz_oracle.py — itself pinned by tests/test_oracle.py — frozen at
generation time.  Each .npz holds the inputs (X, Y, Psi, omega, masks, theta, model fields) and every output
of GPz() in both modes, getPHI() and predict().  They guard the oracle against regressions and are the
fixed cases the HIP path is compared with through the C ABI (tests/test_gpu_parity.py).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import gpz_oracle as O  # noqa: E402


def make_case(name, n, d, m, k, method, hetero, psi, nan_patterns, seed, masks=True):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    A = rng.standard_normal((d, k)) / np.sqrt(d)
    Y = np.sin(X @ A) + 0.1 * (1.0 + np.abs(X @ A)) * rng.standard_normal((n, k))
    Y -= Y.mean(0)
    model, theta = O.init_theta(X, Y, method, m, hetero, rng)
    theta = theta + 0.05 * rng.standard_normal(theta.size)
    if hetero:
        o = theta.size - 2 * m * k
        theta[o:o + m * k] = 0.05 * rng.standard_normal(m * k)
    Psi = None
    if psi:
        if model.method[1] == "C":
            Psi = np.zeros((d, d, n))
            for i in range(n):
                B = 0.3 * rng.standard_normal((d, d))
                Psi[:, :, i] = B @ B.T
        else:
            Psi = rng.gamma(1.0, 0.2, (n, d))
    if nan_patterns and d > 1:
        rows = rng.random(n) < 0.4
        cols = rng.integers(0, min(d, 2), n)
        X[rows, cols[rows]] = np.nan
    omega = rng.random((n, 1)) + 0.5 if masks else None
    training = rng.random(n) < 0.8 if masks else None
    validation = ~training if masks else None
    r2 = O.GPz(theta, model, X, Y, Psi, omega, training, validation, nargout=2)
    r4 = O.GPz(theta, model, X, Y, Psi, omega, training, validation, nargout=4)
    PHI, Gamma, lnB, N = O.getPHI(X, Psi, theta, model, training, want_N=True)
    out = dict(
        n=n, d=d, m=m, k=k, method=model.method, heteroscedastic=int(hetero),
        X=X, Y=Y, theta=theta,
        has_psi=int(Psi is not None), Psi=Psi if Psi is not None else np.zeros(0),
        has_masks=int(masks), omega=omega if masks else np.zeros(0),
        training=training if masks else np.zeros(0, bool), validation=validation if masks else np.zeros(0, bool),
        nlogML=r2.nlogML, grad=r2.grad, cond=r2.cond,
        trainRMSE=r2.stats["trainRMSE"], trainLL=r2.stats["trainLL"],
        validRMSE=r2.stats.get("validRMSE", np.nan), validLL=r2.stats.get("validLL", np.nan),
        nlogML_partial=r4.nlogML, w=r4.w, iSigma_w=r4.iSigma_w,
        lnBeta_i=lnB,
    )
    if n <= 300:
        out.update(PHI=PHI, N=N)
    if not psi and not nan_patterns:
        model.sets["best"] = {"theta": theta, "w": r4.w, "iSigma_w": r4.iSigma_w}
        Xs = rng.standard_normal((37, d))
        mu, sigma, nu, beta_i, gamma, PHIs, _, _ = O.predict(Xs, model)
        out.update(Xs=Xs, mu=mu, sigma=sigma, nu=nu, beta_i=beta_i, PHIs=PHIs)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    return name


def main():
    names = []
    seed = 100
    for method in ("GL", "VL", "GD", "VD", "GC", "VC"):
        for hetero in (True, False):
            for psi in (False, True):
                for nanp in (False, True):
                    k = 2 if (method in ("VD", "VC") and hetero and not psi and not nanp) else 1
                    n, d, m = (64, 3, 5) if (psi or nanp) else (257, 3, 16)
                    names.append(make_case(f"g_{method}_h{int(hetero)}_p{int(psi)}_n{int(nanp)}", n, d, m, k, method,
                                           hetero, psi, nanp, seed))
                    seed += 1
    # d = 1 (init.m:12-14 forces ?L), d = 10 shapes of the BASELINE configs (scaled down)
    names.append(make_case("g_d1_VL", 257, 1, 16, 1, "VD", True, False, False, 900))
    names.append(make_case("g_c2_shape", 1200, 10, 200, 1, "VD", True, False, False, 901))
    names.append(make_case("g_c4_shape", 1536, 10, 128, 1, "VC", True, False, False, 902, masks=False))
    names.append(make_case("g_m1", 64, 2, 1, 1, "VC", True, False, False, 903))
    print("wrote", len(names), "fixtures")


if __name__ == "__main__":
    main()
