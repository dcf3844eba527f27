"""Host-side callers of the objective/gradient path: ``init``, ``train`` (L-BFGS with a strong-Wolfe line search)
and the small input helpers, mirroring the reference's MATLAB host code so that the package is usable end to end.

This is glue ABOVE the C ABI: all arithmetic on n-sized data still happens in
libgpz_hip.so through :class:`gpz_amd.GPzContext`; what lives here is the p-sized optimiser state and
argument plumbing.  Semantics follow

    GPz/init.m, GPz/train.m, GPz/callBack.m, GPz/fixPsi.m, GPz/getOmega.m, GPz/sample.m, GPz/metrics.m,
    minFunc_2012/minFunc/minFunc.m (LBFGS branch :544-582, step/termination logic :968-1150),
    lbfgsAdd.m, lbfgsProd.m, WolfeLineSearch.m, ArmijoBacktrack.m, polyinterp.m

with minFunc's defaults for 'lbfgs' (corrections 100, c1 1e-4, c2 0.9, cubic interpolation, optTol 1e-5,
progTol 1e-9, 25 line-search iterations).
"""
from __future__ import annotations

import math
import time

import numpy as np

from . import api
from .api import GPzContext, Model


# --------------------------------------------------------------------------------------------------
# minFunc: L-BFGS + strong Wolfe
# --------------------------------------------------------------------------------------------------
def _vdot(a, b):
    """Inner product of two optimiser vectors.  NumPy vectors go through einsum, not BLAS: a threaded ddot on a 10^5-vector
    costs milliseconds where the BLAS pool is wider than the cores the process may use (18 ms against 50 us on the
    8-core build container), and minFunc's iteration is a dozen such products."""
    if isinstance(a, np.ndarray) and isinstance(b, np.ndarray):
        return float(np.einsum("i,i->", a, b))
    return float(a @ b)


def _legal(v):
    """isLegal.m: real, no NaN, no Inf."""
    if isinstance(v, DevVec):
        return v.legal()
    a = np.asarray(v)
    return bool(np.all(np.isfinite(a)))


def _amax(v):
    """max(abs(v)) for a host array or a device vector."""
    return v.amax() if isinstance(v, (DevVec, _Scaled)) else float(np.max(np.abs(v)))


def _asum(v):
    return v.asum() if isinstance(v, DevVec) else float(np.sum(np.abs(v)))


class _Scaled:
    """The product t*d of a scalar and a device vector, kept symbolic."""

    def __init__(self, a, v):
        self.a, self.v = a, v

    def amax(self):
        return abs(self.a) * self.v.amax()


class DevVec:
    """A p-vector that lives on the GPU (a float64 torch tensor used as plain device memory).  It supports exactly the
    operations minFunc's L-BFGS driver and line searches perform on x, g and d — x + t*d, g'd, max|.|, sum|.|, copies —
    so the same driver code runs with host arrays or with device-resident vectors; the reductions are the library's
    kernels (gpz_vec_stats), only scalars come back to the host."""

    __array_priority__ = 1000

    def __init__(self, t):
        self.t = t
        self._own = None        # [., max|v|, sum|v|, .] of this vector, computed once (vectors are never modified in place)
        self._max = None        # max|v| alone, when a product g'v of another vector brought it along (gpz_vec_stats out[3])

    @staticmethod
    def from_host(a, device=0):
        import torch
        return DevVec(torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(f"cuda:{device}"))

    def host(self):
        return self.t.cpu().numpy()

    @property
    def size(self):
        return self.t.numel()

    def _stats(self, other=None):
        from . import _lib
        import ctypes as C
        out = (C.c_double * 4)()
        _lib.check_plain(_lib.load().gpz_vec_stats(self.t.data_ptr(), None if other is None else other.t.data_ptr(),
                                                   self.t.numel(), self.t.device.index or 0, None, out))
        return out

    def __matmul__(self, other):
        st = self._stats(other)                 # one kernel gives g'd and the maxima of both vectors
        if self._own is None:
            self._own = tuple(st)
        if other._max is None:
            other._max = float(st[3])
        return float(st[0])

    def _self_stats(self):
        if self._own is None:
            self._own = tuple(self._stats())
        return self._own

    def amax(self):
        if self._own is None and self._max is not None:
            return self._max
        return float(self._self_stats()[1])

    def asum(self):
        return float(self._self_stats()[2])

    def legal(self):
        return bool(np.isfinite(self.amax()))

    def __rmul__(self, a):
        return _Scaled(float(a), self)          # t*d is only ever added to x or measured: no kernel, no temporary

    def __add__(self, other):
        """x + t*d in one kernel of the library (gpz_vec_axpy)."""
        import torch
        from . import _lib
        a, v = (other.a, other.v) if isinstance(other, _Scaled) else (1.0, other)
        out = torch.empty_like(self.t)
        _lib.check_plain(_lib.load().gpz_vec_axpy(out.data_ptr(), self.t.data_ptr(), a, v.t.data_ptr(), self.t.numel(),
                                                  self.t.device.index or 0, None))
        return DevVec(out)

    def __neg__(self):
        import torch
        from . import _lib
        out = torch.empty_like(self.t)          # 0*x - 1*x written as x + (-2)*x
        _lib.check_plain(_lib.load().gpz_vec_axpy(out.data_ptr(), self.t.data_ptr(), -2.0, self.t.data_ptr(),
                                                  self.t.numel(), self.t.device.index or 0, None))
        return DevVec(out)

    def copy(self):
        return DevVec(self.t.clone())


def _polyinterp(points, xmin_bound=None, xmax_bound=None):
    """Minimiser of the polynomial interpolating the given (x, f, g) rows; ``None`` marks an unknown f or g
    (polyinterp.m uses sqrt(-1)).  Two points with both values known take the closed-form cubic branch."""
    pts = [(float(x), f, g) for x, f, g in points]
    xs = [p[0] for p in pts]
    xmin, xmax = min(xs), max(xs)
    lo = xmin if xmin_bound is None else xmin_bound
    hi = xmax if xmax_bound is None else xmax_bound
    known = sum((p[1] is not None) + (p[2] is not None) for p in pts)
    order = known - 1
    if len(pts) == 2 and order == 3:
        a, b = (0, 1) if pts[0][0] <= pts[1][0] else (1, 0)          # a = point with the smaller x
        xa, fa, ga = pts[a]
        xb, fb, gb = pts[b]
        d1 = ga + gb - 3.0 * (fa - fb) / (xa - xb)
        disc = d1 * d1 - ga * gb
        if disc >= 0.0:
            d2 = math.sqrt(disc)
            t = xb - (xb - xa) * ((gb + d2 - d1) / (gb - ga + 2.0 * d2))
            return min(max(t, lo), hi)
        return 0.5 * (lo + hi)
    # general case: least-squares-free exact interpolation, then the best critical point inside the bounds
    rows, rhs = [], []
    for x, f, g in pts:
        if f is not None:
            rows.append([x ** j for j in range(order, -1, -1)]); rhs.append(f)
    for x, f, g in pts:
        if g is not None:
            rows.append([(order - j) * x ** (order - j - 1) if order - j - 1 >= 0 else 0.0 for j in range(order + 1)])
            rhs.append(g)
    A = np.array(rows, dtype=float)
    params = np.linalg.lstsq(A, np.array(rhs, dtype=float), rcond=None)[0]
    dparams = np.array([params[i] * (order - i) for i in range(order)])
    cps = [lo, hi] + xs
    if np.all(np.isfinite(dparams)) and dparams.size > 0:
        cps += [r.real for r in np.roots(dparams) if abs(r.imag) < 1e-12]
    best, fbest = 0.5 * (lo + hi), np.inf
    for c in cps:
        if lo <= c <= hi:
            fc = float(np.polyval(params, c))
            if fc < fbest:
                best, fbest = c, fc
    return best


def _armijo(fun, x, t, d, f, fr, g, gtd, c1, prog_tol):
    """ArmijoBacktrack.m with LS_interp = 2, LS_multi = 0."""
    f_new, g_new = fun(x + t * d)
    evals = 1
    while f_new > fr + c1 * t * gtd or not _legal(f_new):
        temp = t
        if not _legal(f_new):
            t = 0.5 * t
        elif not _legal(g_new):
            t = _polyinterp([(0.0, f, gtd), (t, f_new, None)], 0.0, t)
        else:
            t = _polyinterp([(0.0, f, gtd), (t, f_new, _vdot(g_new, d))], 0.0, t)
        if t < temp * 1e-3:
            t = temp * 1e-3
        elif t > temp * 0.6:
            t = temp * 0.6
        f_new, g_new = fun(x + t * d)
        evals += 1
        if _amax(t * d) <= prog_tol:
            return 0.0, f, g, evals
    return t, f_new, g_new, evals


def _wolfe(fun, x, t, d, f, g, gtd, c1, c2, max_ls, prog_tol):
    """WolfeLineSearch.m with cubic interpolation (LS_interp = 2): bracketing then zoom.  Returns the step, the
    function value and gradient at x + t*d and the number of evaluations."""
    f_new, g_new = fun(x + t * d)
    evals = 1
    gtd_new = _vdot(g_new, d) if _legal(g_new) else float("nan")
    ls_iter = 0
    t_prev, f_prev, g_prev, gtd_prev = 0.0, f, g, gtd
    nrm_d = _amax(d)
    done = False
    bracket = None
    while ls_iter < max_ls:
        if not _legal(f_new) or not _legal(g_new):
            t = 0.5 * (t + t_prev)
            t, f_new, g_new, ev = _armijo(fun, x, t, d, f, f, g, gtd, c1, prog_tol)
            return t, f_new, g_new, evals + ev
        if f_new > f + c1 * t * gtd or (ls_iter > 1 and f_new >= f_prev):
            bracket = [[t_prev, f_prev, g_prev], [t, f_new, g_new]]
            break
        if abs(gtd_new) <= -c2 * gtd:
            bracket = [[t, f_new, g_new]]
            done = True
            break
        if gtd_new >= 0:
            bracket = [[t_prev, f_prev, g_prev], [t, f_new, g_new]]
            break
        temp = t_prev
        t_prev = t
        min_step = t + 0.01 * (t - temp)
        max_step = t * 10.0
        t = _polyinterp([(temp, f_prev, gtd_prev), (t, f_new, gtd_new)], min_step, max_step)
        f_prev, g_prev, gtd_prev = f_new, g_new, gtd_new
        f_new, g_new = fun(x + t * d)
        evals += 1
        gtd_new = _vdot(g_new, d) if _legal(g_new) else float("nan")
        ls_iter += 1
    if ls_iter == max_ls:
        bracket = [[0.0, f, g], [t, f_new, g_new]]
    insuf = False
    while not done and ls_iter < max_ls:
        lo_pos = 0 if bracket[0][1] <= bracket[1][1] else 1
        hi_pos = 1 - lo_pos
        f_lo = bracket[lo_pos][1]
        b0, b1 = bracket[0][0], bracket[1][0]
        if not (_legal(bracket[0][1]) and _legal(bracket[1][1]) and _legal(bracket[0][2]) and _legal(bracket[1][2])):
            t = 0.5 * (b0 + b1)
        else:
            t = _polyinterp([(b0, bracket[0][1], _vdot(bracket[0][2], d)), (b1, bracket[1][1], _vdot(bracket[1][2], d))])
        bmax, bmin = max(b0, b1), min(b0, b1)
        if min(bmax - t, t - bmin) / (bmax - bmin) < 0.1:
            if insuf or t >= bmax or t <= bmin:
                t = bmax - 0.1 * (bmax - bmin) if abs(t - bmax) < abs(t - bmin) else bmin + 0.1 * (bmax - bmin)
                insuf = False
            else:
                insuf = True
        else:
            insuf = False
        f_new, g_new = fun(x + t * d)
        evals += 1
        gtd_new = _vdot(g_new, d) if _legal(g_new) else float("nan")
        ls_iter += 1
        armijo = f_new < f + c1 * t * gtd
        if not armijo or f_new >= f_lo:
            bracket[hi_pos] = [t, f_new, g_new]
        else:
            if abs(gtd_new) <= -c2 * gtd:
                done = True
            elif gtd_new * (bracket[hi_pos][0] - bracket[lo_pos][0]) >= 0:
                bracket[hi_pos] = list(bracket[lo_pos])
            bracket[lo_pos] = [t, f_new, g_new]
        if not done and abs(bracket[0][0] - bracket[1][0]) * nrm_d < prog_tol:
            break
    best = min(bracket, key=lambda r: r[1])                         # WolfeLineSearch.m:250-253
    return best[0], best[1], best[2], evals


class _LBFGSDevice:
    """The same memory on the GPU: S and Y never leave the device (gpz_lbfgs_* of the C ABI, k_lbfgs.hip)."""

    def __init__(self, p, corrections, device=0):
        from . import _lib
        import ctypes as C
        self._lib = _lib
        self._h = C.c_void_p()
        _lib.check_plain(_lib.load().gpz_lbfgs_create(int(p), int(corrections), int(device), None, C.byref(self._h)))

    def add_step(self, g, g_old, t, d):
        import ctypes as C
        added = C.c_int32()
        self._lib.check_plain(self._lib.load().gpz_lbfgs_add(self._h, g.t.data_ptr(), g_old.t.data_ptr(), float(t),
                                                            d.t.data_ptr(), C.byref(added)))
        return bool(added.value)

    def direction(self, g):
        import torch
        d = torch.empty_like(g.t)
        self._lib.check_plain(self._lib.load().gpz_lbfgs_direction(self._h, g.t.data_ptr(), d.data_ptr()))
        return DevVec(d)

    def close(self):
        if self._h:
            self._lib.load().gpz_lbfgs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _LBFGS:
    """Circular-buffer L-BFGS memory: lbfgsAdd.m (skip when y's <= 1e-10) and the two-loop product lbfgsProd.m."""

    def __init__(self, p, corrections):
        self.S = np.zeros((corrections, p)); self.Y = np.zeros((corrections, p)); self.YS = np.zeros(corrections)   # one pair per row
        self.start, self.end, self.hdiag, self.cap = 0, -1, 1.0, corrections
        self.count = 0

    def add_step(self, g, g_old, t, d):
        return self.add(g - g_old, t * d)

    def add(self, y, s):
        ys = _vdot(y, s)
        if ys <= 1e-10:
            return False
        if self.count < self.cap:
            self.end += 1
            self.count += 1
        else:
            self.start = (self.start + 1) % self.cap
            self.end = (self.end + 1) % self.cap
        self.S[self.end] = s; self.Y[self.end] = y; self.YS[self.end] = ys
        self.hdiag = ys / _vdot(y, y)
        return True

    def direction(self, g):
        idx = [(self.start + q) % self.cap for q in range(self.count)]
        d = -g.copy()
        al = {}
        for i in reversed(idx):
            al[i] = _vdot(self.S[i], d) / self.YS[i]
            d -= al[i] * self.Y[i]
        d *= self.hdiag
        for i in idx:
            be = _vdot(self.Y[i], d) / self.YS[i]
            d += self.S[i] * (al[i] - be)
        return d


def minfunc_lbfgs(fun, x0, max_iter=200, output_fcn=None, corrections=100, opt_tol=1e-5, prog_tol=1e-9, c1=1e-4,
                  c2=0.9, max_ls=25):
    """minFunc(funObj, x0, options) with options.method = 'lbfgs' (minFunc.m:314-1150).

    ``fun(x) -> (f, g)``; ``output_fcn(x, kind, i, fun_evals, f, t, gtd, g, d, opt_cond) -> stop`` is called with
    kind 'init', 'iter' and 'done' like minFunc's outputFcn.  Returns (x, f, exitflag, fun_evals, message)."""
    on_device = isinstance(x0, DevVec)
    x = x0.copy() if on_device else np.asarray(x0, dtype=np.float64).copy()
    f, g = fun(x)
    evals = 1
    if _amax(g) <= opt_tol:
        return x, f, 1, evals, "Optimality Condition below optTol"
    if output_fcn and output_fcn(x, "init", 0, evals, f, None, None, g, None, _amax(g)):
        return x, f, -1, evals, "Stopped by output function"
    mem = _LBFGSDevice(x.size, corrections, x.t.device.index or 0) if on_device else _LBFGS(x.size, corrections)
    exitflag, msg = 0, "Reached Maximum Number of Iterations"
    t = 1.0
    d = None
    g_old = None
    i = 0
    for i in range(1, max_iter + 1):
        if i == 1:
            d = -g
        else:
            mem.add_step(g, g_old, t, d)
            d = mem.direction(g)
        g_old = g.copy()
        gtd = _vdot(g, d)                      # (first: on device vectors the same kernel brings max|d| for the legality test)
        if not _legal(d):
            exitflag, msg = -3, "Step direction is illegal"
            break
        if gtd > -prog_tol:
            exitflag, msg = 2, "Directional Derivative below progTol"
            break
        t = min(1.0, 1.0 / _asum(g)) if i == 1 else 1.0                      # minFunc.m:983,988-990
        f_old = f
        t, f, g, ev = _wolfe(fun, x, t, d, f, g, gtd, c1, c2, max_ls, prog_tol)
        evals += ev
        x = x + t * d
        opt_cond = _amax(g)
        if output_fcn and output_fcn(x, "iter", i, evals, f, t, gtd, g, d, opt_cond):
            exitflag, msg = -1, "Stopped by output function"
            break
        if opt_cond <= opt_tol:
            exitflag, msg = 1, "Optimality Condition below optTol"
            break
        if _amax(t * d) <= prog_tol:
            exitflag, msg = 2, "Step Size below progTol"
            break
        if abs(f - f_old) < prog_tol:
            exitflag, msg = 2, "Function Value changing by less than progTol"
            break
    if output_fcn:
        output_fcn(x, "done", i, evals, f, None, None, g, None, _amax(g))
    return x, f, exitflag, evals, msg


# --------------------------------------------------------------------------------------------------
# small input helpers
# --------------------------------------------------------------------------------------------------
def fixPsi(Psi, n, sdX, method):
    """fixPsi.m: rescale the input-noise variances by sdX and convert to d x d x n (GC/VC) or n x d."""
    if Psi is None:
        return None
    sdX = np.asarray(sdX, dtype=np.float64).ravel()
    d = sdX.size
    Psi = np.asarray(Psi, dtype=np.float64)
    cube = Psi.ndim == 3 and Psi.shape == (d, d, n)
    if Psi.ndim == 1:
        Psi = Psi[:, None]
    outer = np.outer(sdX, sdX)
    if method[1] == "C":
        if cube:
            return Psi / outer[:, :, None]
        new = np.zeros((d, d, n))
        if Psi.shape[1] == 1:
            new[np.arange(d), np.arange(d), :] = (Psi[:, 0][None, :] / np.diag(outer)[:, None])
        else:
            new[np.arange(d), np.arange(d), :] = (Psi / sdX ** 2).T
        return new
    if cube:
        return np.stack([np.diag(Psi[:, :, i] / outer) for i in range(n)])
    if Psi.shape[1] == 1:
        return np.tile(Psi, (1, d)) / sdX ** 2
    return Psi / sdX ** 2


def getOmega(Y, method="balanced", binWidth=None):
    """getOmega.m: cost-sensitive weights ('balanced' inverse bin frequency, 'normalized' (1+Y)^-2, else ones)."""
    Y = np.asarray(Y, dtype=np.float64).ravel()
    if method == "balanced":
        lo, hi = Y.min(), Y.max()
        if binWidth is None:
            binWidth = (hi - lo) / 100.0
        bins = int(math.ceil((hi - lo) / binWidth))
        centers = lo + np.arange(1, bins + 1) * binWidth - binWidth / 2.0
        edges = np.concatenate(([-np.inf], 0.5 * (centers[1:] + centers[:-1]), [np.inf]))
        counts = np.histogram(Y, bins=edges)[0].astype(np.float64)
        ind = np.argmin(api.Dxy(Y[:, None], centers[:, None]), axis=1)
        return (counts.max() / counts[ind])[:, None]
    if method == "normalized":
        return ((1.0 + Y) ** -2.0)[:, None]
    return np.ones((Y.size, 1))


def sample(n, trainSample, validSample, testSample, rng=None):
    """[training,validation,testing] = sample(n,trainSample,validSample,testSample)   (sample.m).
    trainSample < 1: the three arguments are FRACTIONS of n (validation and testing rounded up, training capped by what
    is left, sample.m:3-7 - they are not renormalised); otherwise they are row COUNTS (demo_photoz.m:49).  The masks
    are taken from one random permutation in the order validation, testing, training (sample.m:15-17)."""
    rng = rng or np.random.default_rng()
    if trainSample < 1:
        validSample = int(math.ceil(n * validSample))
        testSample = int(math.ceil(n * testSample))
        trainSample = min(int(math.ceil(n * trainSample)), n - testSample - validSample)
    trainSample, validSample, testSample = int(trainSample), int(validSample), int(testSample)
    r = rng.permutation(n)
    tr = np.zeros(n, bool); va = np.zeros(n, bool); te = np.zeros(n, bool)
    va[r[:validSample]] = True
    te[r[validSample:validSample + testSample]] = True
    tr[r[validSample + testSample:validSample + testSample + trainSample]] = True
    return tr, va, te


def metrics(y, mu, sigma, fun):
    """scores = metrics(y,mu,sigma,fun)   (metrics.m): samples sorted by predictive variance, fun(y,mu,sigma) evaluated
    element-wise on the sorted vectors, running mean cumsum(.)./(1:n)'."""
    y, mu, sigma = (np.asarray(a, dtype=np.float64).ravel() for a in (y, mu, sigma))
    order = np.argsort(sigma, kind="stable")
    vals = np.asarray(fun(y[order], mu[order], sigma[order]), dtype=np.float64).ravel()
    return np.cumsum(vals) / np.arange(1, y.size + 1)


# --------------------------------------------------------------------------------------------------
# init.m / train.m
# --------------------------------------------------------------------------------------------------
def _pca_fill(X):
    """pca.m + fillLinear.m on the training rows: NaN-aware mean / covariance, basis colouring matrix, and X with
    missing entries replaced by their conditional means (pca.m:5-47, fillLinear.m:18-29)."""
    n, d = X.shape
    miss = np.isnan(X)
    Xz = np.where(miss, 0.0, X)
    counts = n - miss.sum(0)
    mu = Xz.sum(0) / counts
    Xc = np.where(miss, 0.0, Xz - mu)
    mm = miss.astype(np.float64)
    sig = n * (Xc.T @ Xc) / (n - mm.T @ mm)
    ev, U = np.linalg.eigh(sig)
    S = np.abs(ev)
    order = np.argsort(-S)
    U, S = U[:, order], S[order]
    Ti = np.diag(np.sqrt(S / (n - 1))) @ U.T
    cov = sig / n
    Xl = X.copy()
    pats = {}
    for i in range(n):
        pats.setdefault(tuple(miss[i]), []).append(i)
    for pat, rows in pats.items():
        u = np.array(pat)
        if u.any():
            o = ~u
            Dl = Xl[np.ix_(rows, o)] - mu[o]
            Xl[np.ix_(rows, u)] = Dl @ np.linalg.solve(cov[np.ix_(o, o)], cov[np.ix_(o, u)]) + mu[u]
    return mu, Ti, Xl


def init(X, Y, method, m, heteroscedastic=True, normalize=True, omega=None, training=None, Psi=None, rng=None,
         device=0):
    """model = init(X,Y,method,m,...)   (init.m): model struct, normalisation, PCA-coloured random centres,
    length-scale heuristic, theta packing and the first solve for w, inv(SIGMA)."""
    X = np.asarray(X, dtype=np.float64); Y = np.asarray(Y, dtype=np.float64)
    if Y.ndim == 1:
        Y = Y[:, None]
    n, d = X.shape
    k = Y.shape[1]
    rng = rng or np.random.default_rng()
    training = np.ones(n, bool) if training is None else np.asarray(training, bool)
    if d == 1:
        method = method[0] + "L"                                                   # init.m:12-14
    if normalize:                                                                  # init.m:22-39
        miss = np.isnan(X)
        Xz = np.where(miss, 0.0, X)
        counts = (~miss).sum(0)
        muX = Xz.sum(0) / counts
        sdX = np.sqrt((Xz ** 2).sum(0) / counts - muX ** 2)
    else:
        muX, sdX = np.zeros(d), np.ones(d)
    muY = Y[training].mean(0)
    model = Model(m=m, d=d, k=k, method=method, heteroscedastic=heteroscedastic, muX=muX, sdX=sdX, muY=muY)
    Yc = Y - muY
    Xn = (X - muX) / sdX
    PsiN = fixPsi(Psi, n, sdX, method) if Psi is not None else None
    var_y = np.var(Yc[training], axis=0, ddof=1)
    b = np.log(var_y)                                                              # init.m:54
    lnAlpha = np.tile(-np.log(var_y), (m, 1))                                      # init.m:55
    mu, Ti, Xl = _pca_fill(Xn[training])                                           # init.m:57,61
    P = (rng.random((m, d)) - 0.5) * math.sqrt(12.0)                               # init.m:58
    P = P @ Ti + mu                                                                # init.m:59
    gamma = np.sqrt(0.5 * m ** (1.0 / d) / np.mean(api.Dxy(Xl, P, device=device), axis=0))   # init.m:62
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
        Gamma = np.einsum("ab,j->abj", np.eye(d), gamma)
    parts = [P.ravel(order="F"), np.ravel(Gamma, order="F"), lnAlpha.ravel(order="F"), b.ravel()]
    if heteroscedastic:
        parts += [np.zeros(m * k), np.zeros(m * k)]                                # init.m:92-97
    theta = np.concatenate(parts)
    ctx = GPzContext(model, Xn, Yc, PsiN, omega, training, None, device=device)   # f = @(params) GPz(...)  init.m:89
    try:
        w, iS, _ = ctx.solve(theta)                                                # init.m:104
    finally:
        ctx.close()
    last = {"theta": theta, "w": w, "iSigma_w": iS, "P": P, "priors": np.ones(m) / m}
    if heteroscedastic:
        last["v"] = np.zeros((m, k))
    model.sets["last"] = last
    model.sets["best"] = dict(last, LL=-np.inf)                                    # init.m:115-117
    return model


def _device_vectors_available():
    """DevVec needs torch (as plain device memory) and a visible GPU."""
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def train(model, X, Y, maxIter=200, maxAttempts=np.inf, omega=None, training=None, validation=None, Psi=None,
          verbose=True, device=0, device_resident=None, dtype="f64", n_gpus=None, reducer="rccl"):
    """model = train(model,X,Y,...)   (train.m + callBack.m): L-BFGS on the negative log marginal likelihood with
    per-iteration statistics, best-on-validation tracking and early stopping after maxAttempts non-improving
    iterations.  device_resident=True keeps theta, the gradient, the search direction and the L-BFGS memory on the GPU
    (gpz_eval_dev + gpz_lbfgs_*): per evaluation only f and the statistics cross PCIe.  The default (None) takes that route
    on one GPU whenever device vectors are available (torch with a visible GPU) - at 100 corrections the NumPy two-loop
    of the host route costs more than a small problem's evaluation (profiles/r06_train_timing.txt: 2.5 x the bare evaluation
    per evaluation at BASELINE config 2 against 1.4 x) - and the host route otherwise; False forces the host route
    (minFunc's own arithmetic order, what the CPU tests run).
    n_gpus (0 = every GPU of the node): the closure is evaluated on several GPUs behind one synchronous call (GPzMulti /
    gpz_mgpu_*: rows sharded, RCCL inside the library) - what the MEX gateway does for a MATLAB train.m; reducer
    "loopback" puts the shards on one GPU (single-GPU machines)."""
    X = np.asarray(X, dtype=np.float64); Y = np.asarray(Y, dtype=np.float64)
    if Y.ndim == 1:
        Y = Y[:, None]
    n = X.shape[0]
    Yc = Y - model.muY                                                             # train.m:30
    Xn = (X - model.muX) / model.sdX                                               # train.m:32-33
    PsiN = fixPsi(Psi, n, model.sdX, model.method) if Psi is not None else None    # train.m:36-38
    training_only = validation is None or not np.asarray(validation).any()
    state = {"best_theta": model.sets["best"]["theta"].copy(), "best_valid": model.sets["best"].get("LL", -np.inf),
             "attempts": 0, "tic": time.time(), "log": []}
    if device_resident is None:
        device_resident = n_gpus is None and _device_vectors_available()
    if n_gpus is not None:
        if device_resident:
            raise ValueError("device_resident optimiser vectors live on one GPU: use it without n_gpus")
        ctx = api.GPzMulti(model, Xn, Yc, PsiN, omega, training, None if training_only else validation, n_gpus=n_gpus,
                           reducer=reducer, dtype=dtype)
    else:
        ctx = GPzContext(model, Xn, Yc, PsiN, omega, training, None if training_only else validation, device=device,
                         dtype=dtype)

    def fun(theta):
        if isinstance(theta, DevVec):
            f, g = ctx.eval_dev(theta.t)
            return f, DevVec(g)
        return ctx.eval(theta)                                                     # refreshes ctx.stats (the globals)

    def callback(theta, kind, i, evals, f, t, gtd, g, d, opt_cond):               # callBack.m
        st = ctx.stats
        if kind == "init":
            if verbose:
                print("\tIter\tlogML/n\t\tTrain RMSE\tTrain MLL" + ("" if training_only else "\tValid RMSE\tValid MLL") + "\tTime")
        elif kind == "iter":
            dt = time.time() - state["tic"]
            state["log"].append((i, -f, st["trainRMSE"], st["trainLL"], st.get("validRMSE", np.nan), st.get("validLL", np.nan)))
            if training_only:
                if verbose:
                    print(f"\t{i}\t{-f:1.5e}\t{st['trainRMSE']:1.5e}\t {st['trainLL']:1.5e}\t{dt:f}")
                state["best_valid"] = st["trainLL"]; state["best_theta"] = theta.copy()
            elif st["validLL"] >= state["best_valid"]:
                if verbose:
                    print(f"\t{i}\t{-f:1.5e}\t{st['trainRMSE']:1.5e}\t{st['trainLL']:1.5e}\t{st['validRMSE']:1.5e}\t[{st['validLL']:1.5e}]\t{dt:f}")
                state["best_valid"] = st["validLL"]; state["best_theta"] = theta.copy(); state["attempts"] = 0
            else:
                state["attempts"] += 1
                if verbose:
                    print(f"\t{i}\t{-f:1.5e}\t{st['trainRMSE']:1.5e}\t{st['trainLL']:1.5e}\t{st['validRMSE']:1.5e}\t {st['validLL']:1.5e}\t{dt:f}")
        state["tic"] = time.time()
        return state["attempts"] == maxAttempts

    try:
        theta0 = model.sets["last"]["theta"]
        if device_resident:
            theta0 = DevVec.from_host(theta0, device)
        theta, f, flag, evals, msg = minfunc_lbfgs(fun, theta0, maxIter, callback)   # train.m:42-48
        if verbose:
            print(msg)
        if device_resident:
            theta = theta.host()
            if isinstance(state["best_theta"], DevVec):
                state["best_theta"] = state["best_theta"].host()
        m, d, k, g_dim = model.m, model.d, model.k, model.g_dim
        for name, th in (("last", theta), ("best", state["best_theta"])):          # train.m:53-80
            w, iS, _ = ctx.solve(th)
            st = {"theta": th.copy(), "w": w, "iSigma_w": iS, "P": th[:m * d].reshape((m, d), order="F"),
                  "priors": api.getPrior(Xn, PsiN, th, model, training, device=device)}   # train.m:59,74
            if model.heteroscedastic:
                o = m * d + g_dim + m * k + k
                st["v"] = th[o:o + m * k].reshape((m, k), order="F")
            if name == "best":
                st["LL"] = model.sets["best"].get("LL", -np.inf)                  # never updated by the reference either
            model.sets[name] = st
        model.train_info = {"exitflag": flag, "funEvals": evals, "message": msg, "f": f,
                            "log": np.array(state["log"], dtype=np.float64).reshape(-1, 6)}    # the numbers of callBack.m's lines
    finally:
        ctx.close()
    return model
