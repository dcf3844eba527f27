"""TEST INFRASTRUCTURE — 50-digit (mpmath) evaluation of GPz's negative log marginal likelihood and its gradient,
written from the model's formulas rather than from oracle/gpz_oracle.py (SURVEY.md §8c pin 5).

One general expression covers every method, input noise and missing inputs:

    Sigma_j = inv(Gamma_j' Gamma_j)  (GC/VC)   or   diag(gamma_jc^-2)  (GL/VL/GD/VD)          getPHI.m:73,93
    M_ij    = Sigma_j,oo + Psi_i,oo            (o = observed dimensions of row i, u = missing)    getPHI.m:84,102
    ln PHI_ij = -1/2 D M^-1 D' + 1/2 ln|Sigma_j,oo| - 1/2 ln|M_ij| - 1/2 |u| ln 2,  D = x_i,o - p_j,o   getPHI.m:76,86,97,104

(the oracle and the reference branch into four differently written cases; with Psi = 0 the two log-determinants cancel).
The objective follows GPz.m:43-82,98-110,233 with a plain LU solve and determinant instead of inv_logdet's SVD.  The
gradient is a central difference at 50 digits (h = 1e-18: truncation ~1e-36, rounding ~1e-32), i.e. exact for every
purpose of an fp64 comparison — an independent check of every gradient block of GPz.m:89-106,146-206,215-231.
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 50


def _expand_gamma(G, method, m, d):
    """Sigma_j as an mp matrix per basis function (d x d)."""
    out = []
    for j in range(m):
        if method in ("GC", "VC"):
            off = 0 if method == "GC" else d * d * j
            Gj = mp.matrix(d, d)
            for b in range(d):
                for a in range(d):
                    Gj[a, b] = G[off + a + d * b]                     # column-major d x d (x m)
            out.append(mp.inverse(Gj.T * Gj))
        else:
            S = mp.zeros(d, d)
            for c in range(d):
                gam = {"GL": lambda: G[0], "VL": lambda: G[j], "GD": lambda: G[c], "VD": lambda: G[j + m * c]}[method]()
                S[c, c] = 1 / (gam * gam)
            out.append(S)
    return out


def nlogml(theta, method, m, d, k, hetero, X, Y, Psi=None, omega=None):
    """-L/(n k) at the working precision.  theta: sequence of mp numbers (or floats); X may contain NaN; Psi is n x d
    (diagonal kinds) or d x d x n (GC/VC), NumPy; omega n x 1 or None."""
    theta = [mp.mpf(t) for t in theta]
    n = X.shape[0]
    g_dim = {"GL": 1, "VL": m, "GD": d, "VD": m * d, "GC": d * d, "VC": d * d * m}[method]
    o0 = 0
    P = [[theta[o0 + j + m * c] for c in range(d)] for j in range(m)]; o0 += m * d
    G = theta[o0:o0 + g_dim]; o0 += g_dim
    lnA = [[theta[o0 + j + m * q] for q in range(k)] for j in range(m)]; o0 += m * k
    b = theta[o0:o0 + k]; o0 += k
    if hetero:
        v = [[theta[o0 + j + m * q] for q in range(k)] for j in range(m)]; o0 += m * k
        lnT = [[theta[o0 + j + m * q] for q in range(k)] for j in range(m)]; o0 += m * k
    Sig = _expand_gamma(G, method, m, d)
    om = [mp.mpf(1)] * n if omega is None else [mp.mpf(float(omega[i, 0])) for i in range(n)]
    ln2 = mp.log(2)
    PHI = mp.matrix(n, m)
    cache = {}
    for i in range(n):
        obs = [c for c in range(d) if not np.isnan(X[i, c])]
        nu_ = d - len(obs)
        for j in range(m):
            key = (j, tuple(obs))
            if key not in cache:
                Soo = mp.matrix(len(obs), len(obs))
                for a, ca in enumerate(obs):
                    for bb, cb in enumerate(obs):
                        Soo[a, bb] = Sig[j][ca, cb]
                cache[key] = (Soo, mp.log(mp.det(Soo)) if obs else mp.mpf(0))
            Soo, ldS = cache[key]
            Mx = Soo.copy()
            if Psi is not None:
                for a, ca in enumerate(obs):
                    if Psi.ndim == 3:
                        for bb, cb in enumerate(obs):
                            Mx[a, bb] += mp.mpf(float(Psi[ca, cb, i]))
                    else:
                        Mx[a, a] += mp.mpf(float(Psi[i, ca]))
            if obs:
                D = mp.matrix([mp.mpf(float(X[i, c])) - P[j][c] for c in obs])
                quad = (D.T * mp.lu_solve(Mx, D))[0]
                ldM = mp.log(mp.det(Mx))
            else:
                quad, ldM = mp.mpf(0), mp.mpf(0)
            PHI[i, j] = mp.exp(-quad / 2 + ldS / 2 - ldM / 2 - nu_ * ln2 / 2)
    Ltot = mp.mpf(0)
    ln2pi = mp.log(2 * mp.pi)
    for q in range(k):
        lnb = [b[q] + (sum(PHI[i, j] * v[j][q] for j in range(m)) if hetero else 0) for i in range(n)]      # getPHI.m:117-125
        wb = [mp.exp(-lnb[i]) * om[i] for i in range(n)]                                                    # GPz.m:43,48
        al = [mp.exp(lnA[j][q]) for j in range(m)]
        S = mp.matrix(m, m)
        rhs = mp.matrix(m, 1)
        for a in range(m):
            for c in range(a, m):
                S[a, c] = S[c, a] = sum(PHI[i, a] * wb[i] * PHI[i, c] for i in range(n))
            S[a, a] += al[a]                                                                                 # GPz.m:65
            rhs[a] = sum(PHI[i, a] * wb[i] * mp.mpf(float(Y[i, q])) for i in range(n))
        w = mp.lu_solve(S, rhs)                                                                              # GPz.m:70
        logdet = mp.log(mp.det(S))                                                                           # GPz.m:67
        Lq = mp.mpf(0)
        for i in range(n):
            delta = sum(PHI[i, j] * w[j] for j in range(m)) - mp.mpf(float(Y[i, q]))                         # GPz.m:77
            Lq += -wb[i] * delta * delta / 2 - lnb[i] * om[i] / 2                                            # GPz.m:81-82
        Lq += -sum(al[j] * w[j] * w[j] for j in range(m)) / 2 + sum(lnA[j][q] for j in range(m)) / 2 - logdet / 2
        if hetero:                                                                                           # GPz.m:103
            Lq += -sum(v[j][q] ** 2 * mp.exp(lnT[j][q]) for j in range(m)) / 2 + sum(lnT[j][q] for j in range(m)) / 2 \
                  - mp.mpf(m * k) * ln2pi / 2
        Ltot += Lq
    Ltot -= ln2pi * sum(om) / 2                                                                              # GPz.m:110
    return -Ltot / (n * k)                                                                                   # GPz.m:233


def gradient(theta, *args, h="1e-18", **kw):
    th = [mp.mpf(float(t)) for t in theta]
    hh = mp.mpf(h)
    g = []
    for e in range(len(th)):
        tp = list(th); tm = list(th)
        tp[e] += hh; tm[e] -= hh
        g.append((nlogml(tp, *args, **kw) - nlogml(tm, *args, **kw)) / (2 * hh))
    return g


# ---- the same objective in torch fp64, differentiated by autograd ------------------------------------------------
def torch_nlogml(theta, method, m, d, k, hetero, X, Y, Psi=None, omega=None):
    """The objective above on torch.float64 tensors (vectorised over rows of one NaN pattern and over basis functions);
    torch.autograd of it is a second, independent derivation of the gradient of GPz.m:89-234 at sizes mpmath cannot
    reach.  theta: 1-D float64 tensor with requires_grad; X, Y, Psi, omega: NumPy."""
    import math
    import torch
    n = X.shape[0]
    g_dim = {"GL": 1, "VL": m, "GD": d, "VD": m * d, "GC": d * d, "VC": d * d * m}[method]
    o0 = 0
    P = theta[o0:o0 + m * d].reshape(d, m).T; o0 += m * d                       # column-major m x d
    G = theta[o0:o0 + g_dim]; o0 += g_dim
    lnA = theta[o0:o0 + m * k].reshape(k, m).T; o0 += m * k
    b = theta[o0:o0 + k]; o0 += k
    if hetero:
        v = theta[o0:o0 + m * k].reshape(k, m).T; o0 += m * k
        lnT = theta[o0:o0 + m * k].reshape(k, m).T; o0 += m * k
    if method in ("GC", "VC"):
        Gm = G.reshape(1, d, d).transpose(1, 2).expand(m, d, d) if method == "GC" else G.reshape(m, d, d).transpose(1, 2)
        Sig = torch.linalg.inv(Gm.transpose(1, 2) @ Gm)                         # m x d x d
    else:
        gam = {"GL": lambda: G.expand(m * d).reshape(m, d), "VL": lambda: G.reshape(m, 1).expand(m, d),
               "GD": lambda: G.reshape(1, d).expand(m, d), "VD": lambda: G.reshape(d, m).T}[method]()
        Sig = torch.diag_embed(gam ** -2.0)
    Xt = torch.as_tensor(np.where(np.isnan(X), 0.0, X))
    miss = np.isnan(X)
    PHI = torch.zeros(n, m, dtype=torch.float64)
    for pat in np.unique(miss, axis=0):
        rows = np.flatnonzero((miss == pat).all(axis=1))
        obs = np.flatnonzero(~pat)
        if obs.size == 0:
            PHI[rows] = math.exp(-0.5 * d * math.log(2.0))
            continue
        oi = torch.as_tensor(obs)
        Soo = Sig[:, oi][:, :, oi]                                             # m x do x do
        M = Soo.unsqueeze(0).expand(rows.size, m, obs.size, obs.size)
        if Psi is not None:
            if Psi.ndim == 3:
                Pr = torch.as_tensor(np.ascontiguousarray(Psi[np.ix_(obs, obs, rows)].transpose(2, 0, 1)))
            else:
                Pr = torch.diag_embed(torch.as_tensor(Psi[np.ix_(rows, obs)]))
            M = M + Pr.unsqueeze(1)
        D = (Xt[rows][:, oi].unsqueeze(1) - P[:, oi].unsqueeze(0)).unsqueeze(-1)        # r x m x do x 1
        quad = (D * torch.linalg.solve(M, D)).sum(dim=(-1, -2))
        lnphi = -0.5 * quad + 0.5 * torch.logdet(Soo).unsqueeze(0) - 0.5 * torch.logdet(M) - 0.5 * (d - obs.size) * math.log(2.0)
        PHI[rows] = torch.exp(lnphi)
    Yt = torch.as_tensor(Y)
    om = torch.ones(n, dtype=torch.float64) if omega is None else torch.as_tensor(np.asarray(omega, dtype=np.float64)[:, 0])
    L = torch.zeros((), dtype=torch.float64)
    for q in range(k):
        lnb = b[q] + (PHI @ v[:, q] if hetero else 0.0)
        wb = torch.exp(-lnb) * om
        S = PHI.T @ (PHI * wb[:, None]) + torch.diag(torch.exp(lnA[:, q]))
        w = torch.linalg.solve(S, PHI.T @ (wb * Yt[:, q]))
        delta = PHI @ w - Yt[:, q]
        L = L - 0.5 * (wb * delta ** 2).sum() - 0.5 * (torch.exp(lnA[:, q]) * w ** 2).sum() + 0.5 * lnA[:, q].sum() \
            - 0.5 * torch.logdet(S) - 0.5 * (lnb * om).sum()
        if hetero:
            L = L - 0.5 * (v[:, q] ** 2 * torch.exp(lnT[:, q])).sum() + 0.5 * lnT[:, q].sum() - 0.5 * m * k * math.log(2 * math.pi)
    L = L - 0.5 * math.log(2 * math.pi) * om.sum()
    return -L / (n * k)
