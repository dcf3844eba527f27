"""CPU oracle for the GPz marginal-likelihood objective/gradient path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gpz_amd/`` may import this module;
it is used by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` as the checker, never as the product path.

What this is
------------
A NumPy fp64 restatement of the reference's MATLAB code, written from reading
the sources under ``/root/reference``:

    GPz/GPz.m:1-263        -> :func:`GPz`
    GPz/getPHI.m:1-128     -> :func:`getPHI`
    GPz/inv_logdet.m:1-15  -> :func:`inv_logdet`
    GPz/Dxy.m:1-9          -> :func:`Dxy`
    GPz/predict.m:1-76 + predictDiag.m:58-74 / predictCov.m:53-69 -> :func:`predict`
    predictDiag.m:75-125 / predictCov.m:70-132 (predictNoisy) -> :func:`predict_noisy`
    predict.m:45-69 + predictDiag.m:127-297 / predictCov.m:134-337 (predictMissing, predictNoisyMissing, all
                           branches dispatched by NaN-pattern group) -> :func:`predict_any`
    GPz/getPrior.m:1-22    -> :func:`getPrior`
    GPz/fixPsi.m:1-55      -> :func:`fixPsi`
    GPz/getOmega.m:1-23    -> :func:`getOmega`
    GPz/init.m:54-98       -> :func:`init_theta` (theta layout + heuristics)

Operation order follows the reference statement by statement (per-basis ``for
j=1:m`` loops, three n*m^2 products, SVD pseudo-inverse), so it doubles as the
"reference path (NumPy restatement, as-written)" CPU baseline of BASELINE.md §5.

PARITY: neither MATLAB nor Octave exists in the build image, and the reference ships no
tests, golden vectors or recorded outputs.  Since round 3 the
restatement is pinned by OUTPUTS OF THE REFERENCE'S OWN FILES RUN HERE: ``oracle/mlite.py``
is an interpreter for the subset of MATLAB those files use, ``oracle/run_reference.py``
executes ``GPz.m`` (with ``getPHI.m``, ``inv_logdet.m``), ``predict.m`` (with ``fixPsi.m``,
``predictDiag.m``, ``predictCov.m``), ``getPrior.m``, ``getOmega.m``, ``Dxy.m`` and whole ``init.m`` ->
``train.m`` runs (``minFunc.m`` calling ``GPz.m``, ``callBack.m``) where they lie under
``/root/reference`` and stores inputs + outputs as ``tests/golden/ref_*.npz`` (71 files, one of them demo_sinc.m's own configuration, eight at d = 13 ... 34);
``tests/test_reference_run.py`` compares this module with them (objective to 1e-12, every
output, all six methods, +/- input noise, +/- missing values, k > 1, both heteroscedastic
modes) and re-executes the files whenever the reference tree is present.  The interpreter
is NOT MATLAB: a reader who does not accept it as a run of the reference should read
"parity unpinned" here, as in rounds 1-2.  The pins below do not depend on it
(tests/test_oracle.py): the minFunc
derivative-check protocol (autoDif/autoGrad.m:34-45, derivativeCheck.m:29-40),
the method-nesting identities of getPHI.m:26-40 / GPz.m:215-225, Psi=0 == no
Psi, omega=1 == no omega, mask=all == no mask, an independent dense n x n
Gaussian log-density (Woodbury) check of GPz.m:65-82,110, a hand-computed
m=1,d=1 case, the agreement of the separately written predictDiag.m /
predictCov.m missing-value branches on a diagonal covariance, 50-digit arithmetic and
autograd (tests/mp_reference.py) and Gauss-Hermite quadrature of the prediction
branches (tests/quad_reference.py).

Known numerical limit of the reference, reproduced here: with input noise the
dGamma chain of GPz.m:174-180 goes through Sigma = inv(Gamma'Gamma) twice and
loses cond(Gamma'Gamma)^1.5 * eps; for cond >~ 1e6 central differences of the
objective disagree with this gradient (tools/c5_parity_by_cond.py).

Conventions: arrays are NumPy float64; ``theta`` is a 1-D vector in the
reference's packing order (column-major reshapes, SURVEY.md §8 "theta layout");
boolean masks are ``bool`` arrays; ``None`` plays MATLAB's ``[]``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

LOG2 = math.log(2.0)
LOG2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------
# model struct (init.m:16-20,41-43,86)
# --------------------------------------------------------------------------
@dataclass
class Model:
    m: int
    d: int
    k: int = 1
    method: str = "VD"
    heteroscedastic: bool = True
    g_dim: int = 0
    muX: Optional[np.ndarray] = None
    sdX: Optional[np.ndarray] = None
    muY: Optional[np.ndarray] = None
    sets: dict = field(default_factory=dict)  # 'best'/'last' -> dict(theta,w,iSigma_w)

    def __post_init__(self):
        if self.g_dim == 0:
            self.g_dim = g_dim_of(self.method, self.m, self.d)
        if self.muX is None:
            self.muX = np.zeros(self.d)
        if self.sdX is None:
            self.sdX = np.ones(self.d)
        if self.muY is None:
            self.muY = np.zeros(self.k)


def g_dim_of(method: str, m: int, d: int) -> int:
    """numel(Gamma) per method (init.m:65-86)."""
    return {"GL": 1, "VL": m, "GD": d, "VD": m * d, "GC": d * d, "VC": d * d * m}[method]


def theta_len(model: Model) -> int:
    p = model.m * model.d + model.g_dim + model.m * model.k + model.k
    if model.heteroscedastic:
        p += 2 * model.m * model.k
    return p


def _col(theta, a, b, shape):
    """reshape(theta(a+1:b), shape) with MATLAB column-major order (0-based a,b)."""
    return np.reshape(np.asarray(theta[a:b], dtype=np.float64), shape, order="F")


def unpack_theta(theta, model: Model):
    """Slices of theta: getPHI.m:24-40,117,122; GPz.m:28,32,98,100."""
    m, d, k, g = model.m, model.d, model.k, model.g_dim
    o = 0
    P = _col(theta, o, o + m * d, (m, d)); o += m * d
    G = np.asarray(theta[o:o + g], dtype=np.float64); o += g
    lnAlpha = _col(theta, o, o + m * k, (m, k)); o += m * k
    b = np.asarray(theta[o:o + k], dtype=np.float64); o += k
    v = lnTau = None
    if model.heteroscedastic:
        v = _col(theta, o, o + m * k, (m, k)); o += m * k
        lnTau = _col(theta, o, o + m * k, (m, k)); o += m * k
    return P, G, lnAlpha, b, v, lnTau


def expand_gamma(G, model: Model):
    """getPHI.m:26-40 — repmat the method's parameters to the VD (m x d) or VC (d x d x m) layout."""
    m, d = model.m, model.d
    method = model.method
    if method == "GL":
        return np.full((m, d), G[0])
    if method == "VL":
        return np.tile(np.reshape(G, (m, 1)), (1, d))
    if method == "GD":
        return np.tile(np.reshape(G, (1, d)), (m, 1))
    if method == "VD":
        return np.reshape(G, (m, d), order="F")
    if method == "GC":
        g = np.reshape(G, (d, d), order="F")
        return np.repeat(g[:, :, None], m, axis=2)
    if method == "VC":
        return np.reshape(G, (d, d, m), order="F")
    raise ValueError(method)


# --------------------------------------------------------------------------
# NaN-pattern grouping (getPHI.m:43-54; GPz.m:118-129; predict.m:45-56)
# --------------------------------------------------------------------------
def nan_groups(X):
    """Greedy grouping of rows by NaN pattern, groups ordered by first occurrence.

    Returns (group_id[n] int32, patterns[G, d] bool).  The reference builds a
    logical n x G matrix; the id form is equivalent (column g == (id == g))."""
    missing = np.isnan(X)
    n = X.shape[0]
    gid = np.full(n, -1, dtype=np.int32)
    pats = []
    lst = np.ones(n, dtype=bool)
    while lst.any():
        first = int(np.argmax(lst))
        same = (np.abs(missing[lst].astype(np.int8) - missing[first].astype(np.int8)).sum(axis=1) == 0)
        idx = np.flatnonzero(lst)[same]
        gid[idx] = len(pats)
        pats.append(missing[first].copy())
        lst[idx] = False
    return gid, (np.array(pats, dtype=bool) if pats else np.zeros((0, X.shape[1]), dtype=bool))


def _sum_log_svd(A):
    """sum(log(svd(A))) — getPHI.m:77,86 (0 for the empty matrix)."""
    if A.size == 0:
        return 0.0
    return float(np.sum(np.log(np.linalg.svd(A, compute_uv=False))))


def _mrdivide(B, A):
    """MATLAB B/A = B*inv(A) solved as a linear system (getPHI.m:76,86)."""
    if A.size == 0:
        return np.zeros((B.shape[0], 0))
    return np.linalg.solve(A.T, B.T).T


def _inv(A):
    if A.size == 0:
        return np.zeros_like(A)
    return np.linalg.inv(A)


# --------------------------------------------------------------------------
# getPHI.m
# --------------------------------------------------------------------------
def getPHI(X, Psi, theta, model: Model, selection=None, want_N=False):
    """[PHI,Gamma,lnBeta_i,N] = getPHI(X,Psi,theta,model,selection)  (getPHI.m:1)."""
    X = np.asarray(X, dtype=np.float64)
    if selection is None:
        selection = np.ones(X.shape[0], dtype=bool)          # :3-5
    selection = np.asarray(selection, dtype=bool)
    d, m, k = model.d, model.m, model.k
    method = model.method
    X = X[selection]                                         # :14
    n = X.shape[0]
    if Psi is not None:                                      # :16-22
        Psi = Psi[:, :, selection] if method[1] == "C" else Psi[selection]
    P, G, lnAlpha, b, v, lnTau = unpack_theta(theta, model)  # :24
    Gamma = expand_gamma(G, model)                           # :26-40

    gid, pats = nan_groups(X)                                # :43-54
    lnPHI = np.zeros((n, m))
    lnN = np.zeros((n, m))
    for g in range(pats.shape[0]):                           # :60
        group = gid == g
        u = pats[g]
        o = ~u
        nu_, no_ = int(u.sum()), int(o.sum())
        Xo = X[:, o]
        for j in range(m):                                   # :67
            Delta = Xo - P[j, o]                             # :69
            if method[1] == "C":
                Gj = Gamma[:, :, j]
                Sigma = np.linalg.inv(Gj.T @ Gj)             # :73
                Soo = Sigma[np.ix_(o, o)]
                if Psi is None:
                    Dg = Delta[group]
                    lnPHI[group, j] = -0.5 * np.sum(_mrdivide(Dg, Soo) * Dg, axis=1) - 0.5 * nu_ * LOG2   # :76
                    lnN[group, j] = lnPHI[group, j] - 0.5 * _sum_log_svd(Soo) - 0.5 * no_ * LOG2PI + 0.5 * nu_ * LOG2  # :77
                else:
                    lsS = _sum_log_svd(Soo)
                    for i in np.flatnonzero(group):          # :82
                        PpS = Psi[np.ix_(o, o, [i])][:, :, 0] + Soo          # :84
                        Di = Delta[i:i + 1]
                        lnPHI[i, j] = (-0.5 * np.sum(_mrdivide(Di, PpS) * Di) + 0.5 * lsS
                                       - 0.5 * _sum_log_svd(PpS) - 0.5 * nu_ * LOG2)  # :86
                        lnN[i, j] = lnPHI[i, j] - 0.5 * lsS - 0.5 * no_ * LOG2PI + 0.5 * nu_ * LOG2  # :87
            else:
                Sigma = Gamma[j, o] ** -2.0                  # :93
                if Psi is None:
                    Dg = Delta[group]
                    lnPHI[group, j] = -0.5 * np.sum(Dg ** 2 / Sigma, axis=1) - 0.5 * nu_ * LOG2  # :97
                else:
                    Pg = Psi[np.ix_(group, o)]
                    PpS = Pg + Sigma                         # :102
                    Dg = Delta[group]
                    lnPHI[group, j] = (-0.5 * np.sum(Dg ** 2 / PpS, axis=1)
                                       - 0.5 * np.sum(np.log(1.0 + Pg / Sigma), axis=1) - 0.5 * nu_ * LOG2)  # :104
                lnN[group, j] = lnPHI[group, j] - 0.5 * np.sum(np.log(Sigma)) - 0.5 * no_ * LOG2PI + 0.5 * nu_ * LOG2  # :98,105

    PHI = np.exp(lnPHI)                                      # :113
    lnBeta_i = np.tile(np.reshape(b, (1, k)), (n, 1))        # :117-119
    if model.heteroscedastic:
        lnBeta_i = lnBeta_i + PHI @ v                        # :121-125
    if want_N:
        return PHI, Gamma, lnBeta_i, np.exp(lnN)             # :114
    return PHI, Gamma, lnBeta_i


# --------------------------------------------------------------------------
# inv_logdet.m
# --------------------------------------------------------------------------
def inv_logdet(X):
    """[Xi,logdet] = inv_logdet(X): rank-truncated SVD pseudo-inverse (inv_logdet.m:1-15)."""
    X = np.asarray(X, dtype=np.float64)
    U, s, Vt = np.linalg.svd(X, full_matrices=False)         # :3
    tol = max(X.shape) * np.spacing(np.max(np.abs(s)))       # :7  eps(norm(s,inf))
    r = int(np.sum(s > tol))                                 # :9
    U = U[:, :r]; V = Vt[:r].T; s = s[:r]                    # :10-12
    Xi = (V / s[None, :]) @ U.T                              # :14
    return Xi, float(np.sum(np.log(s)))                      # :15


def cond_of(X):
    s = np.linalg.svd(np.asarray(X, dtype=np.float64), compute_uv=False)
    return float(s[0] / s[-1])


# --------------------------------------------------------------------------
# Dxy.m
# --------------------------------------------------------------------------
def Dxy(X, Y):
    """D = | ||x||^2 + ||y||^2 - 2 x y' |  (Dxy.m:3-7)."""
    X = np.asarray(X, dtype=np.float64); Y = np.asarray(Y, dtype=np.float64)
    xx = np.sum(X ** 2, axis=1)[:, None]
    yy = np.sum(Y ** 2, axis=1)[None, :]
    yb = X @ Y.T
    return np.abs(np.abs(yy + (xx - 2.0 * yb)))


# --------------------------------------------------------------------------
# GPz.m
# --------------------------------------------------------------------------
@dataclass
class GPzResult:
    nlogML: object = 0.0          # scalar (nargout<=2) or 1xk partial (nargout>2)
    grad: object = 0.0
    w: object = 0.0
    iSigma_w: object = 0.0
    PHI: object = None
    stats: dict = field(default_factory=dict)   # trainRMSE, trainLL, validRMSE, validLL (globals, GPz.m:3-7)
    cond: float = float("nan")    # max cond(SIGMA) over outputs (diagnostic, not in the reference)


def GPz(theta, model: Model, X, Y, Psi=None, omega=None, training=None, validation=None, nargout=2):
    """[nlogML,grad,w,iSigma_w,PHI] = GPz(theta,model,X,Y,Psi,omega,training,validation)  (GPz.m:1).

    ``nargout`` selects the reference's two modes: <=2 objective+gradient (+ the four
    global statistics), >2 solve-only (GPz.m:84-87: grad=0, nlogML = unnormalised 1 x k partial,
    statistics untouched)."""
    theta = np.asarray(theta, dtype=np.float64).ravel()
    X = np.asarray(X, dtype=np.float64)
    k, m = model.k, model.m
    method, hetero = model.method, model.heteroscedastic
    n_tot, d = X.shape
    if training is None:
        training = np.ones(n_tot, dtype=bool)                # :16-18
    training = np.asarray(training, dtype=bool)
    if omega is None:
        omega = np.ones((n_tot, 1))                          # :20-22
    omega = np.asarray(omega, dtype=np.float64)
    if omega.ndim == 1:
        omega = omega[:, None]
    n = int(training.sum())                                  # :24
    res = GPzResult()

    P, G, lnAlpha, b, v, lnTau = unpack_theta(theta, model)  # :28,32
    PHI, Gamma, lnBeta_i = getPHI(X, Psi, theta, model, training)   # :30
    res.PHI = PHI
    if Y is None:                                            # :34-40
        return res
    Y = np.asarray(Y, dtype=np.float64)
    if Y.ndim == 1:
        Y = Y[:, None]
    Yt = Y[training]
    om = omega[training]                                     # n x (1 or k)

    beta = np.exp(-lnBeta_i)                                 # :43
    beta_i = beta ** -1.0                                    # :44
    df = -beta                                               # :45
    omega_x_beta = beta * om                                 # :48
    alpha = np.exp(lnAlpha)                                  # :50
    da = alpha

    nu = np.zeros((n, k)); w = np.zeros((m, k)); iSigma_w = np.zeros((m, m, k))
    logdet = np.zeros(k); dwda = np.zeros((m, k)); dlnPHI = np.zeros((n, m)); dlnAlpha = np.zeros((m, k))
    conds = []
    for i in range(k):                                       # :61
        BxPHI = PHI * omega_x_beta[:, i:i + 1]               # :63
        SIGMA = BxPHI.T @ PHI + np.diag(alpha[:, i])         # :65
        iS, logdet[i] = inv_logdet(SIGMA)                    # :67
        conds.append(cond_of(SIGMA))
        iSigma_w[:, :, i] = iS
        nu[:, i] = np.sum(PHI * (PHI @ iS), axis=1)          # :69
        w[:, i] = iS @ (BxPHI.T @ Yt[:, i])                  # :70  (left-to-right: (iS*BxPHI')*y in MATLAB; same value to rounding)
        dwda[:, i] = -iS @ (da[:, i] * w[:, i])              # :71
        dlnPHI = dlnPHI - BxPHI @ iS[:, :m]                  # :72
        dlnAlpha[:, i] = -0.5 * np.diag(iS) * da[:, i]       # :73
    res.cond = float(max(conds))
    res.w, res.iSigma_w = w, iSigma_w

    delta = PHI @ w - Yt                                     # :77
    omega_beta_x_delta = omega_x_beta * delta                # :79
    nlogML = (-0.5 * np.sum(omega_beta_x_delta * delta, axis=0) - 0.5 * np.sum(alpha * w ** 2, axis=0)
              + 0.5 * np.sum(lnAlpha, axis=0) - 0.5 * logdet)                      # :81
    nlogML = nlogML + 0.5 * np.sum(-lnBeta_i * om, axis=0)   # :82

    if nargout > 2:                                          # :84-87
        res.nlogML = nlogML
        res.grad = 0.0
        return res

    dlnAlpha = dlnAlpha - (PHI.T @ omega_beta_x_delta) * dwda - alpha * w * dwda - 0.5 * da * w ** 2 + 0.5  # :89
    dlnPHI = dlnPHI - omega_beta_x_delta @ w[:m].T           # :90
    dbeta = 0.5 * df * (beta_i - (delta ** 2 + nu)) * om     # :93
    db = np.sum(dbeta, axis=0)                               # :94
    if hetero:                                               # :96-108
        tau = np.exp(lnTau)
        nlogML = nlogML - 0.5 * np.sum(v ** 2 * tau, axis=0) + 0.5 * np.sum(lnTau, axis=0) - 0.5 * m * k * LOG2PI  # :103
        dv = PHI[:, :m].T @ dbeta - v * tau                  # :104
        dlnTau = -0.5 * tau * v ** 2 + 0.5                   # :105
        dlnPHI = dlnPHI + dbeta @ v.T                        # :106
    nlogML = float(np.sum(nlogML) - 0.5 * LOG2PI * np.sum(om))   # :110

    dPHI = dlnPHI * PHI[:, :m]                               # :113
    dP = np.zeros_like(P)
    dGamma = np.zeros_like(Gamma)
    Xt = X[training]
    gid, pats = nan_groups(Xt)                               # :118-129
    lst = np.flatnonzero(training)                           # :131
    for g in range(pats.shape[0]):                           # :133
        group = gid == g
        u = pats[g]; o = ~u
        for j in range(m):                                   # :135
            Delta = Xt[:, o] - P[j, o]                       # :142
            if method[1] == "C":
                Gj = Gamma[:, :, j]
                iSigma = Gj.T @ Gj                           # :146
                Sigma = np.linalg.inv(iSigma)                # :147
                Soo = Sigma[np.ix_(o, o)]
                GuuGuo = _inv(iSigma[np.ix_(u, u)]) @ iSigma[np.ix_(u, o)]       # :156,178
                A = Gj[:, o] - Gj[:, u] @ GuuGuo
                if Psi is None:
                    iSoo = np.linalg.inv(Soo)                # :151
                    Dg = Delta[group]; dpg = dPHI[group, j]
                    dP[j, o] = dP[j, o] + (dpg @ Dg) @ iSoo  # :152
                    diSoo = -0.5 * (Dg * dpg[:, None]).T @ Dg                      # :154
                    dGo = 2.0 * A @ diSoo                    # :157
                    dGamma[:, o, j] = dGamma[:, o, j] + dGo  # :158
                    dGamma[:, u, j] = dGamma[:, u, j] - dGo @ GuuGuo.T            # :159
                else:
                    iSoo_ = np.linalg.inv(Soo)
                    for i in np.flatnonzero(group):          # :168
                        iPSoo = np.linalg.inv(Soo + Psi[np.ix_(o, o, [lst[i]])][:, :, 0])   # :170
                        Di = Delta[i:i + 1]
                        dP[j, o] = dP[j, o] + dPHI[i, j] * (Di @ iPSoo)[0]        # :172
                        dSoo = 0.5 * (iSoo_ - iPSoo + iPSoo @ (Di.T @ Di) @ iPSoo)  # :174
                        diSoo = -Soo @ dSoo @ Soo            # :176
                        dGo = 2.0 * A @ diSoo                # :179
                        dGamma[:, o, j] = dGamma[:, o, j] + dPHI[i, j] * dGo       # :180
                        dGamma[:, u, j] = dGamma[:, u, j] - dPHI[i, j] * dGo @ GuuGuo.T  # :181
            else:
                Sigma = Gamma[j, o] ** -2.0                  # :189
                Dg = Delta[group]; dpg = dPHI[group, j]
                if Psi is None:
                    dP[j, o] = dP[j, o] + (dpg @ Dg) / Sigma                       # :192
                    dGamma[j, o] = dGamma[j, o] - Gamma[j, o] * np.sum(Dg ** 2 * dpg[:, None], axis=0)  # :194
                else:
                    Pg = Psi[np.ix_(lst[group], o)]
                    PpS = Pg + Sigma                         # :200
                    dP[j, o] = dP[j, o] + dpg @ (Dg / PpS)   # :202
                    PxiS = (1.0 + Pg / Sigma) ** -1.0        # :204
                    dGamma[j, o] = dGamma[j, o] - Gamma[j, o] * (dpg @ (Dg * PxiS) ** 2
                                                                 - dpg @ (PxiS * Sigma - Sigma))  # :206

    if method == "GL":                                       # :215-225
        dG = np.array([dGamma.sum()])
    elif method == "VL":
        dG = dGamma.sum(axis=1)
    elif method == "GD":
        dG = dGamma.sum(axis=0)
    elif method == "GC":
        dG = dGamma.sum(axis=2).ravel(order="F")
    else:
        dG = dGamma.ravel(order="F")
    parts = [dP.ravel(order="F"), dG, dlnAlpha.ravel(order="F"), db.ravel()]     # :227
    if hetero:
        parts += [dv.ravel(order="F"), dlnTau.ravel(order="F")]                   # :229-231
    grad = np.concatenate(parts)
    res.nlogML = -nlogML / (n * k)                           # :233
    res.grad = -grad / (n * k)                               # :234

    om1 = omega[training, 0:1]                               # omega(training) == first column  (:236)
    res.stats["trainRMSE"] = math.sqrt(np.sum(delta ** 2 * om1) / (n * k))          # :236
    res.stats["trainLL"] = float(np.sum((-0.5 * beta * delta ** 2 + 0.5 * np.log(beta)) * om) / (n * k) - 0.5 * LOG2PI)  # :237

    if validation is not None and np.asarray(validation).size > 0:                  # :239
        validation = np.asarray(validation, dtype=bool)
        nv = int(validation.sum())
        PHIv, _, lnBv = getPHI(X, Psi, theta, model, validation)                    # :243
        betav = np.exp(-lnBv)
        deltav = PHIv @ w - Y[validation]                    # :254-255  (nu :250-252 is computed and unused)
        omv1 = omega[validation, 0:1]
        res.stats["validRMSE"] = math.sqrt(np.sum(deltav ** 2 * omv1) / (nv * k))   # :258
        res.stats["validLL"] = float(np.sum((-0.5 * betav * deltav ** 2 + 0.5 * np.log(betav)) * omega[validation])
                                     / (nv * k) - 0.5 * LOG2PI)                     # :259
    return res


# --------------------------------------------------------------------------
# predict.m (no-Psi / no-missing branch -> predictFull)
# --------------------------------------------------------------------------
def predict(X, model: Model, whichSet="best", selection=None):
    """[mu,sigma,nu,beta_i,gamma,PHI,w,iSigma_w] = predict(X,model,...) restricted to the branch
    predict.m:25-43,60-73 -> predictDiag.m:39-43,58-74 / predictCov.m:34-38,53-69 (no Psi, no NaN)."""
    X = np.asarray(X, dtype=np.float64)
    if selection is None:
        selection = np.ones(X.shape[0], dtype=bool)
    st = model.sets[whichSet]
    X = (X[selection] - model.muX) / model.sdX               # :25,35-36
    if np.isnan(X).any():
        raise NotImplementedError("oracle.predict covers only the no-missing branch (SURVEY.md §8 a27)")
    theta, w, iSigma_w = st["theta"], st["w"], st["iSigma_w"]
    n, k = X.shape[0], model.k
    PHI, _, ElnS = getPHI(X, None, theta, model, None)       # predictDiag.m:63
    mu = PHI @ w                                             # :65
    nu = np.zeros((n, k))
    for out in range(k):
        nu[:, out] = np.sum(PHI * (PHI @ iSigma_w[:, :, out]), axis=1)   # :69-71
    beta_i = np.exp(ElnS)                                    # :73
    gamma = np.zeros((n, k))                                 # :74
    sigma = nu + beta_i + gamma                              # predict.m:72
    mu = mu + model.muY                                      # predict.m:73
    return mu, sigma, nu, beta_i, gamma, PHI, w, iSigma_w


def predict_noisy(X, Psi, model: Model, whichSet="best"):
    """predict() with input noise and no missing values: predict.m:25-43,60-73 -> predictNoisy
    (predictDiag.m:75-125 for GL/VL/GD/VD, predictCov.m:70-132 for GC/VC).  X raw (n x d), Psi raw in any layout
    fixPsi.m accepts.  Returns (mu, sigma, nu, beta_i, gamma, PHI)."""
    X = np.asarray(X, dtype=np.float64)
    n, d = X.shape
    st = model.sets[whichSet]
    Xn = (X - model.muX) / model.sdX                          # predict.m:35-36
    PsiN = fixPsi(Psi, n, model.sdX, model.method)            # predict.m:43
    theta, w, iSigma_w = st["theta"], st["w"], st["iSigma_w"]
    m, k = model.m, model.k
    P, G, lnAlpha, b, v, lnTau = unpack_theta(theta, model)
    if v is None:
        v = np.zeros((m, k))                                  # predictDiag.m:17-21
    Gamma = expand_gamma(G, model)
    PHI, _, ElnS = getPHI(Xn, PsiN, theta, model, None)       # predictDiag.m:80
    mu = PHI @ w
    nu = np.zeros((n, k)); gamma = np.zeros((n, k)); VlnS = np.zeros((n, k))
    if model.method[1] != "C":
        iSigma = Gamma ** 2.0                                 # :89-91
        Sigma = Gamma ** -2.0
        lnz = -0.5 * np.sum(np.log(iSigma), axis=1)
        for i in range(m):
            Z = None
            for j in range(i + 1):                            # :93-94
                iCij = iSigma[i] + iSigma[j]
                Cij = 1.0 / iCij
                cij = (P[i] * iSigma[i] + P[j] * iSigma[j]) / iCij
                lnZij = (lnz[i] + lnz[j] - 0.5 * np.sum((P[i] - P[j]) ** 2 / (Sigma[i] + Sigma[j]))
                         - 0.5 * np.sum(np.log(Sigma[i] + Sigma[j])))          # :101
                Delta = Xn - cij
                CpP = PsiN + Cij
                lnNxc = -0.5 * np.sum(Delta ** 2 / CpP, axis=1) - 0.5 * np.sum(np.log(CpP), axis=1)   # :107
                Z = np.exp(lnZij + lnNxc)[:, None]
                gamma = gamma + 2.0 * Z * (w[i] * w[j])
                VlnS = VlnS + 2.0 * Z * (v[i] * v[j])
                nu = nu + 2.0 * Z * iSigma_w[i, j, :]
            # the loop variable j is still i here (:117-119): the diagonal term is counted once
            gamma = gamma - Z * (w[i] * w[i])
            VlnS = VlnS - Z * (v[i] * v[i])
            nu = nu - Z * iSigma_w[i, i, :]
    else:
        iSigma = np.zeros((d, d, m)); Sigma = np.zeros((d, d, m)); lnz = np.zeros(m)
        for i in range(m):                                    # predictCov.m:86-92
            iSigma[:, :, i] = Gamma[:, :, i].T @ Gamma[:, :, i]
            Sigma[:, :, i] = np.linalg.inv(iSigma[:, :, i])
            lnz[i] = -0.5 * _sum_log_svd(iSigma[:, :, i])
        for i in range(m):
            for j in range(i + 1):
                iCij = iSigma[:, :, i] + iSigma[:, :, j]
                Cij = np.linalg.inv(iCij)
                cij = _mrdivide((P[i] @ iSigma[:, :, i] + P[j] @ iSigma[:, :, j])[None, :], iCij)[0]
                Dl = (P[i] - P[j])[None, :]
                Sij = Sigma[:, :, i] + Sigma[:, :, j]
                lnZij = lnz[i] + lnz[j] - 0.5 * float((_mrdivide(Dl, Sij) @ Dl.T)[0, 0]) - 0.5 * _sum_log_svd(Sij)   # :105
                for r in range(n):                            # the reference loops samples outermost (:94); same sums
                    Dx = (Xn[r] - cij)[None, :]
                    CpP = PsiN[:, :, r] + Cij
                    lnNxc = -0.5 * float((_mrdivide(Dx, CpP) @ Dx.T)[0, 0]) - 0.5 * _sum_log_svd(CpP)                # :111
                    Z = math.exp(lnZij + lnNxc)
                    c2 = 2.0 if j < i else 1.0                # 2x in the loop, minus 1x for j == i (:123-125)
                    gamma[r] += c2 * Z * (w[i] * w[j])
                    VlnS[r] += c2 * Z * (v[i] * v[j])
                    nu[r] += c2 * Z * iSigma_w[i, j, :]
    VlnS = VlnS - (ElnS - b) ** 2                             # predictDiag.m:123
    gamma = gamma - mu ** 2
    beta_i = np.exp(ElnS) * (1.0 + 0.5 * VlnS)
    sigma = nu + beta_i + gamma                               # predict.m:72
    return mu + model.muY, sigma, nu, beta_i, gamma, PHI


def _pm_finish(PHI, w, v, b, gamma, VlnS, nu):
    """Tail shared by predictMissing / predictNoisyMissing (predictDiag.m:198-209, predictCov.m:216-229)."""
    mu = PHI @ w
    ElnS = PHI @ v
    VlnS = VlnS - ElnS ** 2
    ElnS = ElnS + b
    beta_i = np.exp(ElnS) * (1.0 + 0.5 * VlnS)
    gamma = gamma - mu ** 2
    return mu, nu, beta_i, gamma, PHI


def _predict_missing_diag(X, Psi, Gamma, w, v, b, P, iSigma_w, priors):
    """predictMissing (Psi is None, predictDiag.m:127-209) and predictNoisyMissing (predictDiag.m:211-297) for one
    group of rows sharing the NaN pattern of X[0].  Psi: n x d or None."""
    o = ~np.isnan(X[0]); u = ~o
    n = X.shape[0]; m, k = w.shape
    iSigma = Gamma ** 2.0; Sigma = Gamma ** -2.0
    lnz = -0.5 * np.sum(np.log(iSigma), axis=1)                 # :138 / :222 (0.5*sum(log(Sigma)) is the same number)
    No = np.zeros((n, m)); Ex = np.zeros((n, m))
    for i in range(m):                                           # :142-148 / :226-233
        Delta = X[:, o] - P[i, o]
        if Psi is None:
            No[:, i] = np.exp(-0.5 * np.sum(Delta ** 2 / Sigma[i, o], axis=1) - 0.5 * np.sum(np.log(Sigma[i, o])))
        else:
            SpP = Psi[:, o] + Sigma[i, o]
            No[:, i] = np.exp(-0.5 * np.sum(Delta ** 2 / SpP, axis=1) - 0.5 * np.sum(np.log(SpP), axis=1))
        Ex[:, i] = No[:, i] * priors[i]
    Pio = Ex / Ex.sum(axis=1, keepdims=True)                     # :150-152
    # Nij(i,j) over the missing dimensions (:158); PHI(:,i) = No(:,i) * sum_j Pio(:,j) Nij(i,j), times exp(lnz) (:160-161)
    dP = P[:, None, u] - P[None, :, u]
    SS = Sigma[:, None, u] + Sigma[None, :, u]
    Nij = np.exp(-0.5 * np.sum(dP ** 2 / SS, axis=2) - 0.5 * np.sum(np.log(SS), axis=2))
    PHI = No * (Pio @ Nij.T) * np.exp(lnz)[None, :]
    gamma = np.zeros((n, k)); nu = np.zeros((n, k)); VlnS = np.zeros((n, k))
    for i in range(m):                                           # :170-195 / :254-283
        for j in range(i + 1):
            Cij = 1.0 / (iSigma[i] + iSigma[j])
            cij = (P[i] * iSigma[i] + P[j] * iSigma[j]) * Cij
            Delta = X[:, o] - cij[o]
            if Psi is None:
                Nobs = np.exp(-0.5 * np.sum(Delta ** 2 / Cij[o], axis=1) - 0.5 * np.sum(np.log(Cij[o])))
            else:
                CpP = Psi[:, o] + Cij[o]
                Nobs = np.exp(-0.5 * np.sum(Delta ** 2 / CpP, axis=1) - 0.5 * np.sum(np.log(CpP), axis=1))
            Dl = P[:, u] - cij[u]
            CpS = Sigma[:, u] + Cij[u]
            Nu = np.exp(-0.5 * np.sum(Dl ** 2 / CpS, axis=1) - 0.5 * np.sum(np.log(CpS), axis=1))
            EcCij = Nobs * (Pio @ Nu)                            # sum(N.*Pio,2), N = No*Nu'
            Dl = P[i] - P[j]
            Z = (math.exp(lnz[i] + lnz[j] - 0.5 * np.sum(Dl ** 2 / (Sigma[i] + Sigma[j]))
                          - 0.5 * np.sum(np.log(Sigma[i] + Sigma[j]))) * EcCij)[:, None]
            c2 = 2.0 if j < i else 1.0                           # 2x in the loop, minus 1x for the (i,i) term after it
            gamma = gamma + c2 * Z * (w[i] * w[j])
            VlnS = VlnS + c2 * Z * (v[i] * v[j])
            nu = nu + c2 * Z * iSigma_w[i, j, :]
    return _pm_finish(PHI, w, v, b, gamma, VlnS, nu)


def _predict_missing_cov(X, Psi, Gamma, w, v, b, P, iSigma_w, priors):
    """predictMissing (Psi is None, predictCov.m:134-229) and predictNoisyMissing (predictCov.m:231-337) for one
    group of rows sharing the NaN pattern of X[0].  Psi: d x d x n or None.  Kept quirk: in predictNoisyMissing the
    post-loop correction reads `PHI(id,j) = PHI(id,i)-NPio` with j == i (:316), the same as predictMissing's :205."""
    o = ~np.isnan(X[0]); u = ~o
    oi = np.flatnonzero(o); ui = np.flatnonzero(u)
    n, d = X.shape; m, k = w.shape
    iSigma = np.zeros((d, d, m)); Sigma = np.zeros((d, d, m)); lnz = np.zeros(m)
    R = []
    for i in range(m):
        iSigma[:, :, i] = Gamma[:, :, i].T @ Gamma[:, :, i]
        Sigma[:, :, i] = np.linalg.inv(iSigma[:, :, i])
        lnz[i] = -0.5 * _sum_log_svd(iSigma[:, :, i])
        Soo = Sigma[np.ix_(oi, oi, [i])][:, :, 0]
        R.append(np.linalg.solve(Soo, Sigma[np.ix_(oi, ui, [i])][:, :, 0]))       # Sigma(o,o)\Sigma(o,~o)
    PHI = np.zeros((n, m)); gamma = np.zeros((n, k)); nu = np.zeros((n, k)); VlnS = np.zeros((n, k))
    Ex = np.zeros((n, m))
    X_hat = np.zeros((n, d, m))
    Psi_hat = np.zeros((d, d, m, n if Psi is not None else 1))
    for i in range(m):
        Soo = Sigma[np.ix_(oi, oi, [i])][:, :, 0]
        Delta = X[:, oi] - P[i, oi]
        cond_u = Sigma[np.ix_(ui, ui, [i])][:, :, 0] - Sigma[np.ix_(ui, oi, [i])][:, :, 0] @ R[i]
        if Psi is None:
            Ex[:, i] = np.exp(-0.5 * np.sum(_mrdivide(Delta, Soo) * Delta, axis=1) - 0.5 * _sum_log_svd(Soo)) * priors[i]
            Psi_hat[np.ix_(ui, ui, [i], [0])] = cond_u[:, :, None, None]
        else:
            T = np.vstack([np.eye(oi.size), R[i].T])             # [eye(do); R']
            # [~,unshuffle] = sort([find(o) find(~o)]); Psi_hat(unshuffle,unshuffle,i,id) = T*Psi(o,o,id)*T'   (:266-268)
            # As an ASSIGNMENT target this scatters block row a (dimension perm[a]) to row unshuffle[a], which is the
            # intended row perm[a] only when the permutation is its own inverse; kept as written.
            unshuffle = np.argsort(np.concatenate([oi, ui]), kind="stable")
            for r in range(n):
                Moo = Soo + Psi[np.ix_(oi, oi, [r])][:, :, 0]
                Dr = Delta[r][None, :]
                Ex[r, i] = math.exp(-0.5 * float((_mrdivide(Dr, Moo) @ Dr.T)[0, 0]) - 0.5 * _sum_log_svd(Moo)) * priors[i]
                blk = T @ Psi[np.ix_(oi, oi, [r])][:, :, 0] @ T.T
                full = np.zeros((d, d))
                full[np.ix_(unshuffle, unshuffle)] = blk
                full[np.ix_(ui, ui)] += cond_u
                Psi_hat[:, :, i, r] = full
        X_hat[:, ui, i] = Delta @ R[i] + P[i, ui]
        X_hat[:, oi, i] = X[:, oi]
    Pio = Ex / Ex.sum(axis=1, keepdims=True)

    def dens(Dl, S):                                             # exp(-1/2 sum((D/S).*D,2) - 1/2 sum(log(svd(S))))
        return np.exp(-0.5 * np.sum(_mrdivide(Dl, S) * Dl, axis=1) - 0.5 * _sum_log_svd(S))

    for r in range(n if Psi is not None else 1):
        rows = slice(r, r + 1) if Psi is not None else slice(0, n)
        pr = r if Psi is not None else 0
        for i in range(m):
            NPio = None; Z = None
            for j in range(i + 1):
                iCij = iSigma[:, :, i] + iSigma[:, :, j]
                Cij = np.linalg.inv(iCij)
                cij = _mrdivide((P[i] @ iSigma[:, :, i] + P[j] @ iSigma[:, :, j])[None, :], iCij)[0]
                N = dens(X_hat[rows, :, j] - P[i], Sigma[:, :, i] + Psi_hat[:, :, j, pr])
                NPio = N * Pio[rows, j]
                PHI[rows, i] += NPio
                N = dens(X_hat[rows, :, i] - P[j], Sigma[:, :, j] + Psi_hat[:, :, i, pr])
                NPio = N * Pio[rows, i]
                PHI[rows, j] += NPio
                EcCij = 0.0
                for l in range(m):
                    N = dens(X_hat[rows, :, l] - cij, Cij + Psi_hat[:, :, l, pr])
                    EcCij = EcCij + N * Pio[rows, l]
                Dl = (P[i] - P[j])[None, :]
                Sij = Sigma[:, :, i] + Sigma[:, :, j]
                Z = (math.exp(lnz[i] + lnz[j] - 0.5 * float((_mrdivide(Dl, Sij) @ Dl.T)[0, 0]) - 0.5 * _sum_log_svd(Sij))
                     * EcCij)[:, None]
                c2 = 2.0 if j < i else 1.0
                gamma[rows] += c2 * Z * (w[i] * w[j])
                VlnS[rows] += c2 * Z * (v[i] * v[j])
                nu[rows] += c2 * Z * iSigma_w[i, j, :]
            PHI[rows, i] -= NPio                                 # the (i,i) term was added twice (:205 / :316)
    PHI = PHI * np.exp(lnz)[None, :]
    return _pm_finish(PHI, w, v, b, gamma, VlnS, nu)


def predict_any(X, model: Model, Psi=None, whichSet="best", selection=None):
    """[mu,sigma,nu,beta_i,gamma,PHI] = predict(X,model,'Psi',Psi,...) with every branch of predict.m:45-69:
    rows are grouped by NaN pattern; a group without missing values goes to predictFull / predictNoisy, a group
    with missing values to predictMissing / predictNoisyMissing (predictDiag.m:39-55, predictCov.m:34-50)."""
    X = np.asarray(X, dtype=np.float64)
    if selection is not None:
        sel = np.asarray(selection, dtype=bool)
        X = X[sel]
        if Psi is not None:
            Psi = np.asarray(Psi)
            Psi = Psi[:, :, sel] if (model.method[1] == "C" and Psi.ndim == 3) else Psi[sel]
    n, d = X.shape
    st = model.sets[whichSet]
    Xn = (X - model.muX) / model.sdX
    PsiN = fixPsi(Psi, n, model.sdX, model.method)
    theta, w, iSigma_w = st["theta"], st["w"], st["iSigma_w"]
    m, k = model.m, model.k
    P, G, lnAlpha, b, v, lnTau = unpack_theta(theta, model)
    if v is None:
        v = np.zeros((m, k))
    Gamma = expand_gamma(G, model)
    priors = np.asarray(st.get("priors", np.ones(m) / m), dtype=np.float64).ravel()
    cov = model.method[1] == "C"
    mu = np.zeros((n, k)); nu = np.zeros((n, k)); beta_i = np.zeros((n, k)); gamma = np.zeros((n, k)); PHI = np.zeros((n, m))
    gid, pats = nan_groups(Xn)
    G_ = len(pats)
    sub = Model(m=m, d=d, k=k, method=model.method, heteroscedastic=model.heteroscedastic)
    sub.sets = model.sets
    for g in range(G_):
        idx = np.flatnonzero(gid == g)
        Xg = Xn[idx]
        Pg = None if PsiN is None else (PsiN[:, :, idx] if cov else PsiN[idx])
        if not np.isnan(Xg[0]).any():
            if Pg is None:
                out = predict(Xg, sub, whichSet)
                res = (out[0], out[2], out[3], out[4], out[5])
            else:
                out = predict_noisy(Xg, Pg, sub, whichSet)
                res = (out[0], out[2], out[3], out[4], out[5])
        else:
            fn = _predict_missing_cov if cov else _predict_missing_diag
            res = fn(Xg, Pg, Gamma, w, v, b, P, iSigma_w, priors)
        for dst, src in zip((mu, nu, beta_i, gamma, PHI), res):
            dst[idx] = src
    sigma = nu + beta_i + gamma
    return mu + model.muY, sigma, nu, beta_i, gamma, PHI


def getPrior(X, Psi, theta, model: Model, selection=None):
    """prior = getPrior(X,Sx,theta,model,set)   (getPrior.m:1-22): EM-style fixed point on the mixture weights of
    the normalised basis densities N (N is loop-invariant; the reference recomputes it every iteration)."""
    m = model.m
    prior = np.ones(m) / m
    N = getPHI(X, Psi, theta, model, selection, want_N=True)[3]
    for _ in range(100):
        old = prior
        wgt = N * prior
        wgt = wgt / wgt.sum(axis=1, keepdims=True)
        prior = wgt.mean(axis=0)
        if np.linalg.norm(old - prior) / np.linalg.norm(old + prior) < 1e-10:
            break
    return prior


# --------------------------------------------------------------------------
# host-side input helpers the hot path's callers use
# --------------------------------------------------------------------------
def fixPsi(Psi, n, sdX, method):
    """fixPsi.m:1-55 — rescale by sdX and convert to d x d x n cube (*C) or n x d."""
    if Psi is None:
        return None
    sdX = np.asarray(sdX, dtype=np.float64).ravel()
    d = sdX.size
    Psi = np.asarray(Psi, dtype=np.float64)
    cube = Psi.ndim == 3 and Psi.shape == (d, d, n)
    if Psi.ndim == 1:
        Psi = Psi[:, None]
    if method[1] == "C":
        new = np.zeros((d, d, n))
        if not cube:
            if Psi.shape[1] == 1:
                for i in range(n):
                    new[:, :, i] = (np.eye(d) * Psi[i, 0]) / np.outer(sdX, sdX)
            else:
                for i in range(n):
                    new[:, :, i] = np.diag(Psi[i] / sdX ** 2)
        else:
            new = Psi / np.outer(sdX, sdX)[:, :, None]
        return new
    if not cube:
        if Psi.shape[1] == 1:
            return np.tile(Psi, (1, d)) / sdX ** 2
        return Psi / sdX ** 2
    new = np.zeros((n, d))
    for i in range(n):
        new[i] = np.diag(Psi[:, :, i] / np.outer(sdX, sdX))
    return new


def getOmega(Y, method="balanced", binWidth=None):
    """getOmega.m:1-23 (cost-sensitive weights)."""
    Y = np.asarray(Y, dtype=np.float64).ravel()
    n = Y.size
    if method == "balanced":
        minY, maxY = Y.min(), Y.max()
        if binWidth is None:
            binWidth = (maxY - minY) / 100.0
        bins = int(math.ceil((maxY - minY) / binWidth))
        centers = minY + np.arange(1, bins + 1) * binWidth - binWidth / 2.0
        # hist(Y,centers): bin edges midway between centres, outer bins open-ended
        edges = np.concatenate(([-np.inf], 0.5 * (centers[1:] + centers[:-1]), [np.inf]))
        counts = np.histogram(Y, bins=edges)[0].astype(np.float64)
        ind = np.argmin(Dxy(Y[:, None], centers[:, None]), axis=1)
        return (counts.max() / counts[ind])[:, None]
    if method == "normalized":
        return ((1.0 + Y) ** -2.0)[:, None]
    return np.ones((n, 1))


def init_theta(X, Y, method, m, heteroscedastic=True, rng=None, training=None):
    """theta packing + heuristics of init.m:54-98 on already-normalised X, centred Y, no missing
    values (pca/fillLinear reduce to mean/cov in that case).  ``rng`` replaces MATLAB's rand."""
    X = np.asarray(X, dtype=np.float64); Y = np.asarray(Y, dtype=np.float64)
    if Y.ndim == 1:
        Y = Y[:, None]
    n, d = X.shape
    k = Y.shape[1]
    if d == 1:
        method = method[0] + "L"                             # init.m:12-14
    if training is None:
        training = np.ones(n, dtype=bool)
    rng = rng or np.random.default_rng(0)
    Xt, Yt = X[training], Y[training]
    b = np.log(np.var(Yt, axis=0, ddof=1))                   # :54
    lnAlpha = np.tile(-np.log(np.var(Yt, axis=0, ddof=1)), (m, 1))   # :55
    mu = Xt.mean(axis=0)
    C = np.cov(Xt.T, ddof=1).reshape(d, d)
    S, U = np.linalg.eigh(C)
    order = np.argsort(-np.abs(S)); S = np.abs(S[order]); U = U[:, order]
    Vi = np.diag(np.sqrt(S)) @ U.T                           # pca.m:37-45 (Ti)
    P = (rng.random((m, d)) - 0.5) * math.sqrt(12.0)         # :58
    P = P @ Vi + mu                                          # :59
    gamma = np.sqrt(0.5 * m ** (1.0 / d) / np.mean(Dxy(Xt, P), axis=0))   # :62
    if method == "GL":
        Gamma = np.array([gamma.mean()])
    elif method == "VL":
        Gamma = gamma.copy()
    elif method == "GD":
        Gamma = np.ones(d) * gamma.mean()
    elif method == "VD":
        Gamma = np.tile(gamma[:, None], (1, d))
    elif method == "GC":
        Gamma = np.eye(d) * gamma.mean()
    else:
        Gamma = np.zeros((d, d, m))
        for j in range(m):
            Gamma[:, :, j] = np.eye(d) * gamma[j]
    model = Model(m=m, d=d, k=k, method=method, heteroscedastic=heteroscedastic)
    parts = [P.ravel(order="F"), Gamma.ravel(order="F"), lnAlpha.ravel(order="F"), b.ravel()]
    if heteroscedastic:
        parts += [np.zeros(m * k), np.zeros(m * k)]          # :92-97
    return model, np.concatenate(parts)


def fd_gradient(fun, theta):
    """Central-difference gradient with the reference's step (autoDif/autoGrad.m:34-45):
    mu = 2*sqrt(1e-12)*(1+norm(x))."""
    theta = np.asarray(theta, dtype=np.float64)
    p = theta.size
    mu = 2.0 * math.sqrt(1e-12) * (1.0 + np.linalg.norm(theta))
    g = np.zeros(p)
    for j in range(p):
        e = np.zeros(p); e[j] = mu
        g[j] = (fun(theta + e) - fun(theta - e)) / (2.0 * mu)
    return g
