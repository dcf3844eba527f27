"""TEST / BENCH INFRASTRUCTURE — the "vectorised CPU" mode of the reported CPU baseline.

Same mathematics as GPz.m + getPHI.m (k = 1, no input noise, no missing values: the shapes of BASELINE configs 2-4), written
the way a NumPy/OpenBLAS user would write it instead of the way the reference is written:
  * no `for j=1:m` loops (getPHI.m:67, GPz.m:135): the whitened differences Gamma_j (x_i - p_j) of a row chunk are ONE
    n_c x (m d) GEMM, the dP / dGamma sums ONE (m d) x d GEMM;
  * two n m^2 products instead of three (GPz.m:65,69,72: PHI*iSigma_w is formed once and serves nu and dlnPHI);
  * Cholesky inverse / log-determinant instead of the SVD of inv_logdet.m:3-15 (SIGMA is positive definite here).
It is checked against the statement-level oracle (gpz_oracle.GPz) in tests/test_oracle.py and is never imported by the
product package.  Only bench.py's cpu_baseline leg and the tests use it."""
import math

import numpy as np
import scipy.linalg as sla

from . import gpz_oracle as O

LOG2PI = math.log(2.0 * math.pi)


def GPz(theta, model, X, Y, omega=None, chunk=1024):
    """(nlogML, grad) of GPz.m:233-234 for method VD / VC, k = 1, all rows training, X complete, no Psi."""
    if model.k != 1 or model.method not in ("VD", "VC"):
        raise ValueError("vectorised baseline: methods VD / VC with one output")
    X = np.asarray(X, dtype=np.float64)
    if np.isnan(X).any():
        raise ValueError("vectorised baseline: complete inputs only")
    n, d = X.shape
    m, hetero, cov = model.m, model.heteroscedastic, model.method == "VC"
    y = np.asarray(Y, dtype=np.float64).reshape(n)
    om = np.ones(n) if omega is None else np.asarray(omega, dtype=np.float64).reshape(n)
    P, G, lnAlpha, b, v, lnTau = O.unpack_theta(np.asarray(theta, dtype=np.float64).ravel(), model)
    Gamma = O.expand_gamma(G, model)                         # VC: d x d x m, VD: m x d
    if cov:
        Gall = np.ascontiguousarray(Gamma.transpose(1, 0, 2).reshape(d, d * m, order="F"))   # [b, a + d j] = Gamma_j[a, b]
        pG = np.einsum("jb,abj->ja", P, Gamma)               # Gamma_j p_j
    else:
        Gsq = Gamma ** 2                                      # 1 / Sigma_j  (getPHI.m:93)

    def whitened(Xc):
        """U[i, j, a] = (Gamma_j (x_i - p_j))_a  (VC)  /  Gamma_j[a] (x_i - p_j)_a  (VD)."""
        if cov:
            return (Xc @ Gall).reshape(Xc.shape[0], m, d) - pG[None]
        return (Xc[:, None, :] - P[None]) * Gamma[None]

    # pass 1: PHI, beta, SIGMA  (getPHI.m:60-125, GPz.m:43-65)
    PHI = np.empty((n, m))
    for r in range(0, n, chunk):
        Xc = X[r:r + chunk]
        if cov:
            U = whitened(Xc)
            PHI[r:r + chunk] = np.exp(-0.5 * np.einsum("ija,ija->ij", U, U))
        else:   # sum_a Gamma_ja^2 (x_ia - p_ja)^2 expanded into two GEMMs
            PHI[r:r + chunk] = np.exp(-0.5 * ((Xc ** 2) @ Gsq.T - 2.0 * Xc @ (Gsq * P).T + np.sum(Gsq * P ** 2, axis=1)[None]))
    lnBeta_i = b[0] + (PHI @ v[:, 0] if hetero else 0.0)
    beta = np.exp(-lnBeta_i)
    wb = beta * om
    alpha = np.exp(lnAlpha[:, 0])
    BxPHI = PHI * wb[:, None]
    SIGMA = BxPHI.T @ PHI + np.diag(alpha)
    c, low = sla.cho_factor(SIGMA, lower=True)
    iS = sla.cho_solve((c, low), np.eye(m))
    logdet = 2.0 * np.sum(np.log(np.diag(c)))
    T = PHI @ iS                                              # serves GPz.m:69 and :72
    nu = np.einsum("ij,ij->i", PHI, T)
    w = iS @ (BxPHI.T @ y)
    dwda = -iS @ (alpha * w)
    delta = PHI @ w - y
    obd = wb * delta
    nlogML = (-0.5 * obd @ delta - 0.5 * alpha @ w ** 2 + 0.5 * lnAlpha.sum() - 0.5 * logdet - 0.5 * lnBeta_i @ om
              if hetero else
              -0.5 * obd @ delta - 0.5 * alpha @ w ** 2 + 0.5 * lnAlpha.sum() - 0.5 * logdet - 0.5 * b[0] * om.sum())
    dlnAlpha = -0.5 * np.diag(iS) * alpha - (PHI.T @ obd) * dwda - alpha * w * dwda - 0.5 * alpha * w ** 2 + 0.5
    dbeta = 0.5 * (-beta) * (1.0 / beta - (delta ** 2 + nu)) * om
    db = dbeta.sum()
    dlnPHI = -wb[:, None] * T - np.outer(obd, w)
    parts_tail = []
    if hetero:
        tau = np.exp(lnTau[:, 0])
        nlogML += -0.5 * (v[:, 0] ** 2) @ tau + 0.5 * lnTau.sum() - 0.5 * m * LOG2PI
        parts_tail = [PHI.T @ dbeta - v[:, 0] * tau, -0.5 * tau * v[:, 0] ** 2 + 0.5]
        dlnPHI += np.outer(dbeta, v[:, 0])
    nlogML -= 0.5 * LOG2PI * om.sum()
    dPHI = dlnPHI * PHI

    # pass 2: dP, dGamma  (GPz.m:133-213)
    a0 = dPHI.sum(axis=0)
    if cov:
        s1 = np.zeros((m, d))                                 # sum_i dPHI_ij U_ij
        s2 = np.zeros((m * d, d))                             # sum_i dPHI_ij U_ij x_i'
        for r in range(0, n, chunk):
            Xc = X[r:r + chunk]
            W = whitened(Xc) * dPHI[r:r + chunk, :, None]
            s1 += W.sum(axis=0)
            s2 += W.reshape(Xc.shape[0], m * d).T @ Xc
        dP = np.einsum("ja,abj->jb", s1, Gamma)               # (sum dPHI Delta) Gamma'Gamma       GPz.m:152
        dGam = -(s2.reshape(m, d, d) - s1[:, :, None] * P[:, None, :])      # -sum dPHI (Gamma Delta') Delta   :154-158
        dG = dGam.transpose(1, 2, 0).ravel(order="F")
    else:
        s1 = np.zeros((m, d))                                 # sum_i dPHI_ij Delta_ij
        s2 = np.zeros((m, d))                                 # sum_i dPHI_ij Delta_ij^2
        for r in range(0, n, chunk):
            Xc = X[r:r + chunk]
            s1 += dPHI[r:r + chunk].T @ Xc
            s2 += dPHI[r:r + chunk].T @ Xc ** 2
        s2 = s2 - 2.0 * s1 * P + a0[:, None] * P ** 2
        s1 = s1 - a0[:, None] * P
        dP = s1 * Gsq                                         # GPz.m:192
        dG = (-Gamma * s2).ravel(order="F")                   # GPz.m:194
    grad = np.concatenate([dP.ravel(order="F"), dG, dlnAlpha, [db]] + parts_tail)
    return -nlogML / n, -grad / n
