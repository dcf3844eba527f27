"""TEST INFRASTRUCTURE — CPU restatement of the optimiser pieces that call the objective/gradient path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(gpz_amd/, the C ABI) never does.

What is restated, statement by statement, from the reference tree (paths relative to /root/reference):

    isLegal                 minFunc_2012/minFunc/isLegal.m:1-2
    lbfgsAdd                minFunc_2012/minFunc/lbfgsAdd.m:1-31        (+ mex/lbfgsAddC.c:35-38 for the store)
    lbfgsProd               minFunc_2012/minFunc/lbfgsProd.m:1-32
    lbfgsProdC              minFunc_2012/minFunc/mex/lbfgsProdC.c:46-88  (the loop order of the MEX twin)
    polyinterp              minFunc_2012/minFunc/polyinterp.m:1-116     (plot branch :118-145 left out)
    ArmijoBacktrack         minFunc_2012/minFunc/ArmijoBacktrack.m:1-138 (no Hessian output)
    WolfeLineSearch         minFunc_2012/minFunc/WolfeLineSearch.m:1-253 (LS_interp 0, 1, 2; the 'mixed'
                            interpolation of LS_interp = 3, :263-358, is not used by GPz and is left out)
    minFunc (LBFGS branch)  minFunc_2012/minFunc/minFunc.m:259-263,312-378,544-582 (non-damped),962-1160
                            with the defaults of minFunc_processInputOptions.m:60-67,123-152

Conventions kept from MATLAB: S and Y are p x corrections with one pair per COLUMN; lbfgs_start / lbfgs_end are
1-based; an unknown function value or derivative handed to polyinterp is sqrt(-1) (here: the Python complex 1j).

PARITY: pinned by the reference's own files run here.  MATLAB/Octave are absent from the image and the four MEX C files need mex.h
(absent; writing a stand-in header for a reference build is not allowed), but oracle/mlite.py executes minFunc.m,
minFunc_processInputOptions.m, WolfeLineSearch.m, ArmijoBacktrack.m, polyinterp.m, isLegal.m, lbfgsAdd.m and lbfgsProd.m where they lie
(oracle/run_reference.py: make_lbfgs, make_minfunc, make_train; the compiled lbfgsAddC / lbfgsProdC are stood in for by their MATLAB
twins, as minFunc itself does with useMex = 0) and tests/test_reference_run.py holds this file to their outputs: the ring memory call
by call, 13 line searches to 1e-13, 6 whole runs and 5 init.m -> train.m runs with GPz.m as the objective.  Pins that share no code
with either live in tests/test_minfunc_oracle.py: the dense BFGS inverse-Hessian recursion for the two-loop product, lbfgsProd vs
lbfgsProdC, polyinterp's closed-form cubic vs its general branch, analytic minimisers, and the Wolfe conditions at every accepted step.
"""
from __future__ import annotations

import math

import numpy as np

I = 1j   # sqrt(-1): "not known" marker of polyinterp.m:18


# ---- isLegal.m ---------------------------------------------------------------------------------------
def isLegal(v):
    """isLegal.m:2 — no imaginary part, no NaN, no Inf."""
    a = np.asarray(v)
    if np.iscomplexobj(a) and np.any(a.imag != 0):
        return False
    a = a.real if np.iscomplexobj(a) else a
    return bool(np.sum(np.isnan(a)) == 0 and np.sum(np.isinf(a)) == 0)


def _matlab_min(vals):
    """[m, pos] = min(v) for a short real vector: NaNs are ignored (all NaN -> NaN, position 1), first index on ties.
    pos is 1-based."""
    best, pos = None, 1
    for q, v in enumerate(vals):
        if v != v:
            continue
        if best is None or v < best:
            best, pos = v, q + 1
    if best is None:
        return float("nan"), 1
    return best, pos


# ---- lbfgsAdd.m -----------------------------------------------------------------------------------------
def lbfgsAdd(y, s, S, Y, YS, lbfgs_start, lbfgs_end, Hdiag):
    """lbfgsAdd.m:1-31.  S, Y (p x corrections) and YS (corrections) are updated in place like the MEX store
    (lbfgsAddC.c:35-38); returns (lbfgs_start, lbfgs_end, Hdiag, skipped)."""
    ys = float(y @ s)                                             # :2
    skipped = 0                                                   # :3
    corrections = S.shape[1]                                      # :4
    if ys > 1e-10:                                                # :5
        if lbfgs_end < corrections:                               # :6
            lbfgs_end = lbfgs_end + 1                             # :7
            if lbfgs_start != 1:                                  # :8
                if lbfgs_start == corrections:                    # :9
                    lbfgs_start = 1                               # :10
                else:
                    lbfgs_start = lbfgs_start + 1                 # :12
        else:
            lbfgs_start = min(2, corrections)                     # :16
            lbfgs_end = 1                                         # :17
        S[:, lbfgs_end - 1] = s                                   # :23
        Y[:, lbfgs_end - 1] = y                                   # :24
        YS[lbfgs_end - 1] = ys                                    # :26
        Hdiag = ys / float(y @ y)                                 # :29
    else:
        skipped = 1                                               # :31
    return lbfgs_start, lbfgs_end, Hdiag, skipped


# ---- lbfgsProd.m ----------------------------------------------------------------------------------------
def lbfgsProd(g, S, Y, YS, lbfgs_start, lbfgs_end, Hdiag):
    """lbfgsProd.m:1-32: d = -H g by the two-loop recursion over the circular memory."""
    nVars, maxCorrections = S.shape                               # :8
    if lbfgs_start == 1:                                          # :9
        ind = list(range(1, lbfgs_end + 1))                       # :10
        nCor = lbfgs_end - lbfgs_start + 1                        # :11
    else:
        ind = list(range(lbfgs_start, maxCorrections + 1)) + list(range(1, lbfgs_end + 1))   # :13
        nCor = maxCorrections                                     # :14
    al = np.zeros(nCor)                                           # :16
    be = np.zeros(nCor)                                           # :17
    d = -np.asarray(g, dtype=np.float64)                          # :19
    for j in range(1, len(ind) + 1):                              # :20
        i = ind[len(ind) - j]                                     # :21  ind(end-j+1)
        al[i - 1] = float(S[:, i - 1] @ d) / YS[i - 1]            # :22
        d = d - al[i - 1] * Y[:, i - 1]                           # :23
    d = Hdiag * d                                                 # :27
    for i in ind:                                                 # :29
        be[i - 1] = float(Y[:, i - 1] @ d) / YS[i - 1]            # :30
        d = d + S[:, i - 1] * (al[i - 1] - be[i - 1])             # :31
    return d


def lbfgsProdC(g, S, Y, YS, lbfgs_start, lbfgs_end, Hdiag, literal=False):
    """mex/lbfgsProdC.c:46-88: the same product with the MEX file's own loop structure (0-based columns, the block
    [0, lbfgs_end) first and then the wrapped block, alpha / beta indexed by column).  literal=True also spells out the
    element loops over j (accumulation in ascending j, as the C code does); otherwise they are NumPy dot / axpy."""
    nVars, maxCor = S.shape                                       # :29-30
    alpha = np.zeros(maxCor)                                      # :39 (the C code sizes it nCor and indexes it by column)
    beta = np.zeros(maxCor)                                       # :40

    def dot(A, i, d):
        if not literal:
            return float(A[:, i] @ d)
        acc = 0.0
        for j in range(nVars):
            acc += A[j, i] * d[j]
        return acc

    def axpy(d, a, A, i):
        if not literal:
            d += a * A[:, i]
            return
        for j in range(nVars):
            d[j] += a * A[j, i]

    d = np.empty(nVars)
    for j in range(nVars):                                        # :46-47
        d[j] = -g[j]
    for i in range(lbfgs_end - 1, -1, -1):                        # :49
        alpha[i] = dot(S, i, d) / YS[i]                           # :50-53
        axpy(d, -alpha[i], Y, i)                                  # :54-55
    if lbfgs_start != 1:                                          # :57
        for i in range(maxCor - 1, lbfgs_start - 2, -1):          # :58
            alpha[i] = dot(S, i, d) / YS[i]                       # :59-62
            axpy(d, -alpha[i], Y, i)                              # :63-64
    for j in range(nVars):                                        # :68-69
        d[j] *= Hdiag
    if lbfgs_start != 1:                                          # :71
        for i in range(lbfgs_start - 1, maxCor):                  # :72
            beta[i] = dot(Y, i, d) / YS[i]                        # :73-76
            axpy(d, alpha[i] - beta[i], S, i)                     # :77-78
    for i in range(0, lbfgs_end):                                 # :81
        beta[i] = dot(Y, i, d) / YS[i]                            # :82-85
        axpy(d, alpha[i] - beta[i], S, i)                         # :86-87
    return d


# ---- polyinterp.m -----------------------------------------------------------------------------------------
def polyinterp(points, xminBound=None, xmaxBound=None):
    """polyinterp.m:1-116 (doPlot = 0).  points: rows [x, f, g]; f or g = 1j when unknown.  Returns (minPos, fmin);
    the two-point cubic shortcut (:43-60) returns fmin = None like the reference (output never assigned)."""
    pts = np.array(points, dtype=np.complex128)
    nPoints = pts.shape[0]                                        # :24
    order = int(np.sum(pts[:, 1:3].imag == 0)) - 1                # :25
    xmin = float(np.min(pts[:, 0].real))                          # :27
    xmax = float(np.max(pts[:, 0].real))                          # :28
    if xminBound is None:                                         # :31-33
        xminBound = xmin
    if xmaxBound is None:                                         # :34-36
        xmaxBound = xmax
    if nPoints == 2 and order == 3:                               # :43
        x = pts[:, 0].real
        f = pts[:, 1].real
        g = pts[:, 2].real
        minPos = 0 if x[0] <= x[1] else 1                         # :49  [minVal minPos] = min(points(:,1))
        notMinPos = 1 - minPos                                    # :50
        d1 = g[minPos] + g[notMinPos] - 3.0 * (f[minPos] - f[notMinPos]) / (x[minPos] - x[notMinPos])   # :51
        rad = d1 ** 2 - g[minPos] * g[notMinPos]                  # :52
        if rad >= 0 or rad != rad:                                # isreal(d2): sqrt of a negative number is complex; NaN stays real
            d2 = math.sqrt(rad) if rad == rad else float("nan")
            t = x[notMinPos] - (x[notMinPos] - x[minPos]) * ((g[notMinPos] + d2 - d1) / (g[notMinPos] - g[minPos] + 2.0 * d2))   # :54
            return _min_nan(_max_nan(t, xminBound), xmaxBound), None      # :55
        return (xmaxBound + xminBound) / 2.0, None                # :57
    # constraints from the known function values  (:63-74)
    rows, rhs = [], []
    for i in range(nPoints):
        if pts[i, 1].imag == 0:
            x = pts[i, 0].real
            rows.append([x ** j for j in range(order, -1, -1)])   # :67-69
            rhs.append(pts[i, 1].real)
    # constraints from the known derivatives  (:77-86)
    for i in range(nPoints):
        if pts[i, 2].imag == 0:
            x = pts[i, 0].real
            c = [0.0] * (order + 1)
            for j in range(1, order + 1):                         # :80-82
                c[j - 1] = (order - j + 1) * x ** (order - j)
            rows.append(c)
            rhs.append(pts[i, 2].real)
    A = np.array(rows, dtype=np.float64).reshape(len(rows), order + 1)
    b = np.array(rhs, dtype=np.float64)
    params = _linsolve(A, b)                                      # :89
    dParams = np.zeros(order)                                     # :92
    for i in range(1, len(params)):                               # :93-95
        dParams[i - 1] = params[i - 1] * (order - i + 1)
    base = [xminBound, xmaxBound] + [float(v) for v in pts[:, 0].real]
    if np.any(np.isinf(dParams)):                                 # :97
        cp = [complex(v) for v in base]                           # :98
    else:
        cp = [complex(v) for v in base] + [complex(r) for r in _roots(dParams)]   # :100
    fmin = math.inf                                               # :104
    minPos = (xminBound + xmaxBound) / 2.0                        # :105
    for xCP in cp:                                                # :106
        if xCP.imag == 0 and xCP.real >= xminBound and xCP.real <= xmaxBound:   # :107
            fCP = _polyval(params, xCP.real)                      # :108
            if fCP < fmin:                                        # :109 (fCP is real here)
                minPos = xCP.real                                 # :110
                fmin = fCP                                        # :111
    return minPos, fmin


def _max_nan(a, b):
    """MATLAB max(a, b) for scalars: a NaN operand is ignored."""
    if a != a:
        return b
    if b != b:
        return a
    return max(a, b)


def _min_nan(a, b):
    if a != a:
        return b
    if b != b:
        return a
    return min(a, b)


def _linsolve(A, b):
    """linsolve(A, b): LU with partial pivoting for a square system, least squares otherwise (MATLAB warns on a singular
    matrix and still returns)."""
    if A.shape[0] == A.shape[1]:
        try:
            return np.linalg.solve(A, b)
        except np.linalg.LinAlgError:
            return np.full(A.shape[1], np.inf)
    return np.linalg.lstsq(A, b, rcond=None)[0]


def _roots(c):
    """roots(c): leading and trailing zeros stripped (trailing ones give zero roots), then the eigenvalues of the
    companion matrix."""
    c = np.asarray(c, dtype=np.float64)
    if c.size == 0 or not np.all(np.isfinite(c)):
        return []
    nz = np.flatnonzero(c)
    if nz.size == 0:
        return []
    nzero = c.size - 1 - nz[-1]
    c = c[nz[0]:nz[-1] + 1]
    r = [0.0] * nzero
    if c.size > 1:
        comp = np.diag(np.ones(c.size - 2), -1)
        comp[0, :] = -c[1:] / c[0]
        r = list(np.linalg.eigvals(comp)) + r
    return r


def _polyval(p, x):
    y = 0.0
    for c in p:                                                   # Horner, like polyval
        y = y * x + c
    return y


# ---- ArmijoBacktrack.m ---------------------------------------------------------------------------------------
def ArmijoBacktrack(x, t, d, f, fr, g, gtd, c1, LS_interp, LS_multi, progTol, funObj, trace=None):
    """ArmijoBacktrack.m:30-138 without the Hessian output.  Returns (t, x_new, f_new, g_new, funEvals)."""
    f_new, g_new = funObj(x + t * d)                              # :34
    funEvals = 1                                                  # :36
    if trace is not None:
        trace.append(("armijo", t, f_new))
    f_prev = t_prev = g_prev = None
    while f_new > fr + c1 * t * gtd or not isLegal(f_new):       # :38
        temp = t                                                  # :39
        if LS_interp == 0 or not isLegal(f_new):                  # :41
            t = 0.5 * t                                           # :46
        elif LS_interp == 1 or not isLegal(g_new):                # :47
            if funEvals < 2 or LS_multi == 0 or not isLegal(f_prev):   # :49
                t = polyinterp([[0, f, gtd], [t, f_new, I]], 0, t)[0]   # :54
            else:
                t = polyinterp([[0, f, gtd], [t, f_new, I], [t_prev, f_prev, I]], 0, t)[0]   # :60
        else:
            if funEvals < 2 or LS_multi == 0 or not isLegal(f_prev):   # :65
                t = polyinterp([[0, f, gtd], [t, f_new, float(g_new @ d)]], 0, t)[0]   # :70
            elif not isLegal(g_prev):                             # :71
                t = polyinterp([[0, f, gtd], [t, f_new, float(g_new @ d)], [t_prev, f_prev, I]], 0, t)[0]   # :77
            else:
                t = polyinterp([[0, f, gtd], [t, f_new, float(g_new @ d)], [t_prev, f_prev, float(g_prev @ d)]], 0, t)[0]   # :84
        if t < temp * 1e-3:                                       # :89
            t = temp * 1e-3                                       # :93
        elif t > temp * 0.6:                                      # :94
            t = temp * 0.6                                        # :98
        if LS_multi:                                              # :102
            f_prev = f_new                                        # :103
            t_prev = temp                                         # :104
            if LS_interp == 2:                                    # :105
                g_prev = g_new                                    # :106
        f_new, g_new = funObj(x + t * d)                          # :113
        funEvals = funEvals + 1                                   # :115
        if trace is not None:
            trace.append(("armijo", t, f_new))
        if np.max(np.abs(t * d)) <= progTol:                      # :118
            t = 0.0                                               # :122
            f_new = f                                             # :123
            g_new = g                                             # :124
            break
    x_new = x + t * d                                             # :135
    return t, x_new, f_new, g_new, funEvals


# ---- WolfeLineSearch.m ---------------------------------------------------------------------------------------
def WolfeLineSearch(x, t, d, f, g, gtd, c1, c2, LS_interp, LS_multi, maxLS, progTol, funObj, trace=None):
    """WolfeLineSearch.m:32-253 for LS_interp in {0, 1, 2}, no Hessian output.
    Returns (t, f_new, g_new, funEvals).  trace (optional list) receives (phase, t, f) for every evaluation."""
    if LS_interp == 3:
        raise NotImplementedError("mixed interpolation (WolfeLineSearch.m:263-358) is outside the restated path")
    f_new, g_new = funObj(x + t * d)                              # :34
    funEvals = 1                                                  # :36
    gtd_new = float(g_new @ d)                                    # :37
    if trace is not None:
        trace.append(("bracket", t, f_new))
    LSiter = 0                                                    # :42
    t_prev = 0.0                                                  # :43
    f_prev = f                                                    # :44
    g_prev = g                                                    # :45
    gtd_prev = gtd                                                # :46
    nrmD = float(np.max(np.abs(d)))                               # :47
    done = 0                                                      # :48
    bracket = bracketFval = bracketGval = None
    while LSiter < maxLS:                                         # :50
        if not isLegal(f_new) or not isLegal(g_new):              # :53
            t = (t + t_prev) / 2.0                                # :57
            t, x_new, f_new, g_new, armijoFunEvals = ArmijoBacktrack(   # :64-66
                x, t, d, f, f, g, gtd, c1, LS_interp, LS_multi, progTol, funObj, trace)
            funEvals = funEvals + armijoFunEvals                  # :68
            return t, f_new, g_new, funEvals                      # :69
        if f_new > f + c1 * t * gtd or (LSiter > 1 and f_new >= f_prev):   # :73
            bracket = [t_prev, t]                                 # :74
            bracketFval = [f_prev, f_new]                         # :75
            bracketGval = [g_prev, g_new]                         # :76
            break
        elif abs(gtd_new) <= -c2 * gtd:                           # :78
            bracket = [t]                                         # :79
            bracketFval = [f_new]                                 # :80
            bracketGval = [g_new]                                 # :81
            done = 1                                              # :82
            break
        elif gtd_new >= 0:                                        # :84
            bracket = [t_prev, t]                                 # :85
            bracketFval = [f_prev, f_new]                         # :86
            bracketGval = [g_prev, g_new]                         # :87
            break
        temp = t_prev                                             # :90
        t_prev = t                                                # :91
        minStep = t + 0.01 * (t - temp)                           # :92
        maxStep = t * 10                                          # :93
        if LS_interp <= 1:                                        # :94
            t = maxStep                                           # :98
        elif LS_interp == 2:                                      # :99
            t = polyinterp([[temp, f_prev, gtd_prev], [t, f_new, gtd_new]], minStep, maxStep)[0]   # :103
        f_prev = f_new                                            # :108
        g_prev = g_new                                            # :109
        gtd_prev = gtd_new                                        # :110
        f_new, g_new = funObj(x + t * d)                          # :114
        funEvals = funEvals + 1                                   # :116
        gtd_new = float(g_new @ d)                                # :117
        LSiter = LSiter + 1                                       # :118
        if trace is not None:
            trace.append(("bracket", t, f_new))
    if LSiter == maxLS:                                           # :121
        bracket = [0.0, t]                                        # :122
        bracketFval = [f, f_new]                                  # :123
        bracketGval = [g, g_new]                                  # :124
    insufProgress = 0                                             # :132
    while not done and LSiter < maxLS:                            # :135
        f_LO, LOpos = _matlab_min(bracketFval)                    # :138  (1-based)
        HIpos = -LOpos + 3                                        # :139
        if LS_interp <= 1 or not isLegal(bracketFval) or not isLegal(np.array(bracketGval)):   # :142
            t = (bracket[0] + bracket[1]) / 2.0                   # :146  mean(bracket)
        else:
            t = polyinterp([[bracket[0], bracketFval[0], float(bracketGval[0] @ d)],   # :151-152
                            [bracket[1], bracketFval[1], float(bracketGval[1] @ d)]])[0]
        bmax, bmin = max(bracket), min(bracket)
        if min(bmax - t, t - bmin) / (bmax - bmin) < 0.1:         # :167
            if insufProgress or t >= bmax or t <= bmin:           # :171
                if abs(t - bmax) < abs(t - bmin):                 # :175
                    t = bmax - 0.1 * (bmax - bmin)                # :176
                else:
                    t = bmin + 0.1 * (bmax - bmin)                # :178
                insufProgress = 0                                 # :180
            else:
                insufProgress = 1                                 # :185
        else:
            insufProgress = 0                                     # :188
        f_new, g_new = funObj(x + t * d)                          # :195
        funEvals = funEvals + 1                                   # :197
        gtd_new = float(g_new @ d)                                # :198
        LSiter = LSiter + 1                                       # :199
        if trace is not None:
            trace.append(("zoom", t, f_new))
        armijo = f_new < f + c1 * t * gtd                         # :201
        if not armijo or f_new >= f_LO:                           # :202
            bracket[HIpos - 1] = t                                # :205
            bracketFval[HIpos - 1] = f_new                        # :206
            bracketGval[HIpos - 1] = g_new                        # :207
        else:
            if abs(gtd_new) <= -c2 * gtd:                         # :210
                done = 1                                          # :212
            elif gtd_new * (bracket[HIpos - 1] - bracket[LOpos - 1]) >= 0:   # :213
                bracket[HIpos - 1] = bracket[LOpos - 1]           # :215
                bracketFval[HIpos - 1] = bracketFval[LOpos - 1]   # :216
                bracketGval[HIpos - 1] = bracketGval[LOpos - 1]   # :217
            bracket[LOpos - 1] = t                                # :229
            bracketFval[LOpos - 1] = f_new                        # :230
            bracketGval[LOpos - 1] = g_new                        # :231
        if not done and abs(bracket[0] - bracket[1]) * nrmD < progTol:   # :235
            break
    f_LO, LOpos = _matlab_min(bracketFval)                        # :250
    t = bracket[LOpos - 1]                                        # :251
    f_new = bracketFval[LOpos - 1]                                # :252
    g_new = bracketGval[LOpos - 1]                                # :253
    return t, f_new, g_new, funEvals


# ---- minFunc.m, LBFGS branch ---------------------------------------------------------------------------------
def minFunc(funObj, x0, maxIter=500, maxFunEvals=1000, optTol=1e-5, progTol=1e-9, corrections=100, c1=1e-4, c2=0.9,
            LS_init=0, LS_type=1, LS_interp=2, LS_multi=0, Fref=1, outputFcn=None, useMex=1, trace=None):
    """minFunc(funObj, x0, options) with options.Method = 'lbfgs' and Damped = 0; defaults of
    minFunc_processInputOptions.m:60-67,123-152.  outputFcn(x, state, i, funEvals, f, t, gtd, g, d, optCond) -> stop.
    Returns (x, f, exitflag, output) with output = dict(iterations, funcCount, firstorderopt, message, trace)."""
    p = x0.size                                                   # :260
    x = np.array(x0, dtype=np.float64)                            # :262
    t = 1.0                                                       # :263
    f, g = funObj(x)                                              # :314
    funEvals = 1                                                  # :320
    optCond = float(np.max(np.abs(g)))                            # :340
    tr = {"fval": [f], "funcCount": [funEvals], "optCond": [optCond], "t": [], "x": []}   # :342-347 (+ steps, iterates)
    if optCond <= optTol:                                         # :350
        return x, f, 1, dict(iterations=0, funcCount=1, firstorderopt=optCond, message="Optimality Condition below optTol", trace=tr)
    if outputFcn is not None:                                     # :364
        if outputFcn(x, "init", 0, funEvals, f, None, None, g, None, optCond):
            return x, f, -1, dict(iterations=0, funcCount=1, firstorderopt=optCond, message="Stopped by output function", trace=tr)
    exitflag, msg = 0, "Reached Maximum Number of Iterations"
    d = np.zeros(p)                                               # :261
    i = 0
    S = Y = YS = None
    lbfgs_start = lbfgs_end = 0
    Hdiag = 1.0
    g_old = gtd_old = f_old = None
    old_fvals = None
    for i in range(1, maxIter + 1):                               # :381
        if i == 1:                                                # :562
            d = -g                                                # :563
            S = np.zeros((p, corrections))                        # :564
            Y = np.zeros((p, corrections))                        # :565
            YS = np.zeros(corrections)                            # :566
            lbfgs_start = 1                                       # :567
            lbfgs_end = 0                                         # :568
            Hdiag = 1.0                                           # :569
        else:
            lbfgs_start, lbfgs_end, Hdiag, skipped = lbfgsAdd(g - g_old, t * d, S, Y, YS, lbfgs_start, lbfgs_end, Hdiag)   # :571
            if useMex:
                d = lbfgsProdC(g, S, Y, YS, lbfgs_start, lbfgs_end, Hdiag)   # :576
            else:
                d = lbfgsProd(g, S, Y, YS, lbfgs_start, lbfgs_end, Hdiag)   # :578
        g_old = g                                                 # :582
        if not isLegal(d):                                        # :962
            exitflag, msg = -3, "Step direction is illegal!"      # :963-965 (the reference pauses and returns)
            break
        gtd = float(g @ d)                                        # :971
        if gtd > -progTol:                                        # :974
            exitflag, msg = 2, "Directional Derivative below progTol"
            break
        if i == 1:                                                # :981
            t = min(1.0, 1.0 / float(np.sum(np.abs(g))))          # :983
        else:
            if LS_init == 0:                                      # :988
                t = 1.0
            elif LS_init == 1:
                t = t * min(2.0, gtd_old / gtd)                   # :993
            elif LS_init == 2:
                t = min(1.0, 2.0 * (f - f_old) / gtd)             # :996
            elif LS_init == 3:
                t = min(1.0, t * 2.0)                             # :999
            else:
                raise NotImplementedError("LS_init = 4 needs Hessian-vector products (minFunc.m:1000-1017)")
            if t <= 0:                                            # :1019
                t = 1.0
        f_old = f                                                 # :1023
        gtd_old = gtd                                             # :1024
        if Fref == 1:                                             # :1027
            fr = f
        else:
            if i == 1:
                old_fvals = [-math.inf] * Fref                    # :1031
            if i <= Fref:
                old_fvals[i - 1] = f                              # :1035
            else:
                old_fvals = old_fvals[1:] + [f]                   # :1037
            fr = max(old_fvals)                                   # :1039
        f_old = f                                                 # :1052
        if LS_type == 0:                                          # :1053
            t, x, f, g, LSfunEvals = ArmijoBacktrack(x, t, d, f, fr, g, gtd, c1, LS_interp, LS_multi, progTol, funObj, trace)   # :1058
            funEvals = funEvals + LSfunEvals
        else:
            t, f, g, LSfunEvals = WolfeLineSearch(x, t, d, f, g, gtd, c1, c2, LS_interp, LS_multi, 25, progTol, funObj, trace)   # :1067
            funEvals = funEvals + LSfunEvals                      # :1069
            x = x + t * d                                         # :1070
        optCond = float(np.max(np.abs(g)))                        # :1094
        tr["fval"].append(f); tr["funcCount"].append(funEvals); tr["optCond"].append(optCond)   # :1103-1105
        tr["t"].append(t); tr["x"].append(x.copy())
        if outputFcn is not None:                                 # :1109
            if outputFcn(x, "iter", i, funEvals, f, t, gtd, g, d, optCond):
                exitflag, msg = -1, "Stopped by output function"
                break
        if optCond <= optTol:                                     # :1119
            exitflag, msg = 1, "Optimality Condition below optTol"
            break
        if float(np.max(np.abs(t * d))) <= progTol:               # :1127
            exitflag, msg = 2, "Step Size below progTol"
            break
        if abs(f - f_old) < progTol:                              # :1134
            exitflag, msg = 2, "Function Value changing by less than progTol"
            break
        if funEvals >= maxFunEvals:                               # :1142
            exitflag, msg = 0, "Reached Maximum Number of Function Evaluations"
            break
        if i == maxIter:                                          # :1148
            exitflag, msg = 0, "Reached Maximum Number of Iterations"
            break
    if outputFcn is not None:                                     # :1165-1167
        outputFcn(x, "done", i, funEvals, f, t, None, g, d, float(np.max(np.abs(g))))
    return x, f, exitflag, dict(iterations=i, funcCount=funEvals, firstorderopt=float(np.max(np.abs(g))), message=msg, trace=tr)
