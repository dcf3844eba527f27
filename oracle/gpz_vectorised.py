"""TEST / BENCH INFRASTRUCTURE — the "vectorised CPU" mode of the reported CPU baseline.

Same mathematics as GPz.m + getPHI.m (k = 1, no input noise, no missing values: the shapes of BASELINE configs 2-4), written
the way a NumPy/OpenBLAS user would write it instead of the way the reference is written:
  * no `for j=1:m` loops (getPHI.m:67, GPz.m:135): the quadratic forms (x_i - p_j)' Gamma_j'Gamma_j (x_i - p_j) are expanded over
    the d(d+1)/2 monomials x_a x_b of a row, so ln PHI is ONE n x (d(d+1)/2 + d + 1) by m GEMM and the dP / dGamma sums are ONE
    m x n by n x (d(d+1)/2 + d) GEMM followed by the shift to the centres on m small matrices (`form="gemm"`, the default: every
    n-sized operation is a BLAS-3 call or one elementwise pass; `form="chunked"` keeps the round-3 version, whitened differences
    per row chunk, whose einsum temporaries and not BLAS bounded it);
  * two n m^2 products instead of three (GPz.m:65,69,72: PHI*iSigma_w is formed once and serves nu and dlnPHI);
  * Cholesky inverse / log-determinant instead of the SVD of inv_logdet.m:3-15 (SIGMA is positive definite here).
It is checked against the statement-level oracle (gpz_oracle.GPz) in tests/test_oracle.py and is never imported by the
product package.  Only bench.py's cpu_baseline leg and the tests use it."""
import math

import numpy as np
import scipy.linalg as sla

from . import gpz_oracle as O

LOG2PI = math.log(2.0 * math.pi)


def GPz(theta, model, X, Y, omega=None, chunk=1024, form="gemm"):
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
    gemm = cov and form == "gemm"
    if gemm:
        ia, ib = np.triu_indices(d)
        S = np.einsum("caj,cbj->jab", Gamma, Gamma)          # Gamma_j' Gamma_j
        Sp = np.einsum("jab,jb->ja", S, P)
        # ln PHI = [x x' monomials | x | 1] . C_j,  C_j = -1/2 [ (2 - delta_ab) S_ab ; -2 S p ; p' S p ]
        C = -0.5 * np.concatenate([S[:, ia, ib] * np.where(ia == ib, 1.0, 2.0), -2.0 * Sp, np.sum(Sp * P, axis=1)[:, None]], axis=1)
        Q = np.empty((n, ia.size + d + 1))
        np.multiply(X[:, ia], X[:, ib], out=Q[:, :ia.size])
        Q[:, ia.size:ia.size + d] = X
        Q[:, -1] = 1.0
    elif cov:
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
    if gemm:
        np.matmul(Q, C.T, out=PHI)
        np.exp(PHI, out=PHI)
    for r in range(0, n if not gemm else 0, chunk):
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
    # SIGMA = PHI' diag(wb) PHI + diag(alpha) as a rank-n update of one triangle (dsyrk: half the flops of the general product);
    # the scaled copy of PHI is the buffer T is formed in afterwards
    BxPHI = PHI * np.sqrt(wb)[:, None]
    SIGMA = sla.blas.dsyrk(1.0, BxPHI.T, trans=0, lower=1) if BxPHI.flags.c_contiguous else BxPHI.T @ BxPHI
    SIGMA = np.tril(SIGMA) + np.tril(SIGMA, -1).T + np.diag(alpha)
    c, low = sla.cho_factor(SIGMA, lower=True)
    iS = sla.cho_solve((c, low), np.eye(m))
    logdet = 2.0 * np.sum(np.log(np.diag(c)))
    T = np.matmul(PHI, iS, out=BxPHI)                         # serves GPz.m:69 and :72
    nu = np.einsum("ij,ij->i", PHI, T)
    w = iS @ (PHI.T @ (wb * y))
    dwda = -iS @ (alpha * w)
    delta = PHI @ w - y
    obd = wb * delta
    nlogML = (-0.5 * obd @ delta - 0.5 * alpha @ w ** 2 + 0.5 * lnAlpha.sum() - 0.5 * logdet - 0.5 * lnBeta_i @ om
              if hetero else
              -0.5 * obd @ delta - 0.5 * alpha @ w ** 2 + 0.5 * lnAlpha.sum() - 0.5 * logdet - 0.5 * b[0] * om.sum())
    dlnAlpha = -0.5 * np.diag(iS) * alpha - (PHI.T @ obd) * dwda - alpha * w * dwda - 0.5 * alpha * w ** 2 + 0.5
    dbeta = 0.5 * (-beta) * (1.0 / beta - (delta ** 2 + nu)) * om
    db = dbeta.sum()
    # dlnPHI = -wb T - obd w' (+ dbeta v'), dPHI = dlnPHI .* PHI: in place in T's buffer, the rank-1 / rank-2 part as one thin GEMM
    T *= -wb[:, None]
    parts_tail = []
    if hetero:
        tau = np.exp(lnTau[:, 0])
        nlogML += -0.5 * (v[:, 0] ** 2) @ tau + 0.5 * lnTau.sum() - 0.5 * m * LOG2PI
        parts_tail = [PHI.T @ dbeta - v[:, 0] * tau, -0.5 * tau * v[:, 0] ** 2 + 0.5]
        T += np.stack([-obd, dbeta], axis=1) @ np.stack([w, v[:, 0]], axis=0)
    else:
        T -= np.outer(obd, w)
    nlogML -= 0.5 * LOG2PI * om.sum()
    T *= PHI
    dPHI = T

    # pass 2: dP, dGamma  (GPz.m:133-213)
    a0 = dPHI.sum(axis=0)
    if gemm:
        R = dPHI.T @ Q[:, :-1]                                # raw moments: sum_i dPHI_ij [x_a x_b | x_a]
        m1 = R[:, ia.size:]
        M2 = np.empty((m, d, d))
        M2[:, ia, ib] = R[:, :ia.size]
        M2[:, ib, ia] = R[:, :ia.size]
        c1 = m1 - a0[:, None] * P                             # sum_i dPHI_ij Delta
        Smom = M2 - P[:, :, None] * m1[:, None, :] - m1[:, :, None] * P[:, None, :] + a0[:, None, None] * P[:, :, None] * P[:, None, :]
        dP = np.einsum("jab,jb->ja", S, c1)                   # Gamma'Gamma (sum dPHI Delta)       GPz.m:152
        dGam = -np.einsum("acj,jcb->jab", Gamma, Smom)        # -Gamma_j sum dPHI Delta Delta'     :154-158
        dG = dGam.transpose(1, 2, 0).ravel(order="F")
    elif cov:
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
