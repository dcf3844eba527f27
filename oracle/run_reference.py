"""TEST INFRASTRUCTURE — executes the reference's OWN MATLAB files with oracle/mlite.py and writes what they return to
tests/golden/ref_*.npz (inputs + outputs; data only — no reference text is stored).

    python oracle/run_reference.py            # regenerate every fixture from /root/reference/GPz/*.m (this container only)

Fixtures:
  ref_gpz_<case>.npz      [nlogML,grad,w,iSigma_w,PHI] = GPz(theta,model,X,Y,Psi,omega,training,validation) (GPz.m:1, which calls
                          getPHI.m and inv_logdet.m) + the four globals GPz.m:3-7 leaves behind; six methods x {plain, Psi, NaN,
                          Psi + NaN} x heteroscedastic on/off x k, with weights and a training / validation split
  ref_predict_<case>.npz  [mu,sigma,nu,beta_i,gamma,PHI] = predict(X,model,'Psi',Psi) (predict.m:1 -> fixPsi.m, predictDiag.m /
                          predictCov.m: predictFull, predictNoisy, predictMissing, predictNoisyMissing)
  ref_misc.npz            getPHI with all four outputs, inv_logdet (regular and rank-deficient), Dxy, getPrior, getOmega, fixPsi
  ref_lbfgs_mem.npz       minFunc's L-BFGS memory: lbfgsAdd.m / lbfgsProd.m over a wrapping ring with rejected pairs
  ref_minfunc.npz         WolfeLineSearch.m / ArmijoBacktrack.m / polyinterp.m and whole minFunc.m runs on the inputs of the mf_* fixtures
  ref_train_<case>.npz    init.m -> train.m end to end (minFunc.m, callBack.m, GPz.m, getPrior.m, pca.m, fillLinear.m ...): the model after
                          init, callBack's numbers per iteration, the model after train; ref_train_demo_sinc: demo_sinc.m's own
                          configuration (BASELINE config 1) with its predictions and test metrics

The inputs are drawn here with NumPy (seeded); the oracle is NOT involved in producing a fixture — tests/test_reference_run.py
compares it (CPU) and the HIP path (GPU) with these files, and re-executes the .m files when /root/reference exists."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mlite as ML  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
E = ML.EMPTY


def g_dim_of(method, m, d):
    return {"GL": 1, "VL": m, "GD": d, "VD": m * d, "GC": d * d, "VC": d * d * m}[method]


def draw_problem(rng, n, d, m, k, method, hetero, psi, nanfrac, omega_cols=1):
    """seeded inputs of one GPz() call (no oracle code involved: plain NumPy)"""
    X = rng.standard_normal((n, d))
    A = rng.standard_normal((d, k)) / np.sqrt(d)
    Y = np.sin(X @ A) + 0.1 * rng.standard_normal((n, k))
    Y = Y - Y.mean(0)
    P = X[rng.choice(n, m, replace=False)] + 0.1 * rng.standard_normal((m, d))
    gd = g_dim_of(method, m, d)
    if method[1] == "C":
        blocks = [np.eye(d) * (0.8 + 0.4 * rng.random()) + 0.15 * rng.standard_normal((d, d)) / max(1.0, np.sqrt(d / 3.0))
                  for _ in range(1 if method == "GC" else m)]
        G = np.concatenate([b.reshape(-1, order="F") for b in blocks])
    else:
        G = 0.6 + 0.6 * rng.random(gd)
    lnAlpha = 0.3 * rng.standard_normal(m * k)
    b = np.log(0.05) + 0.1 * rng.standard_normal(k)
    parts = [P.reshape(-1, order="F"), G, lnAlpha, b]
    if hetero:
        parts += [0.05 * rng.standard_normal(m * k), 0.2 * rng.standard_normal(m * k)]
    theta = np.concatenate(parts)
    Psi = None
    if psi:
        if method[1] == "C":
            Psi = np.zeros((d, d, n))
            for i in range(n):
                B = 0.3 * rng.standard_normal((d, d))
                Psi[:, :, i] = B @ B.T
        else:
            Psi = rng.gamma(1.0, 0.2, (n, d))
    if nanfrac > 0 and d > 1:
        miss = rng.random((n, d)) < nanfrac
        miss[miss.all(axis=1), 0] = False
        X = X.copy()
        X[miss] = np.nan
    omega = rng.random((n, omega_cols)) + 0.5            # n x 1, or n x k: per-output weights (GPz.m:48 omega(training,:))
    training = rng.random(n) < 0.75
    return dict(theta=theta, X=X, Y=Y, Psi=Psi, omega=omega, training=training, validation=~training)


def run_gpz(ip, method, m, d, k, hetero, pr, nargout=5):
    ms = ML.model_struct(m, d, k, method, hetero, g_dim_of(method, m, d))
    args = [ML.col(pr["theta"]), ms, pr["X"], pr["Y"], E if pr["Psi"] is None else pr["Psi"], pr["omega"],
            pr["training"].reshape(-1, 1), pr["validation"].reshape(-1, 1)]
    ip.globals.clear()
    f, g = ip.call("GPz", args, nargout=2)
    stats = {key: float(np.asarray(ip.globals[key]).reshape(-1)[0]) for key in ("trainRMSE", "trainLL", "validRMSE", "validLL")}
    out5 = ip.call("GPz", args, nargout=5)             # nargout > 2: the solve-only mode of GPz.m:84-87
    return dict(nlogML=float(np.asarray(f).reshape(-1)[0]), grad=np.asarray(g).reshape(-1), nlogML_solve=np.asarray(out5[0]).reshape(-1),
                w=np.asarray(out5[2]), iSigma_w=np.asarray(out5[3]).reshape(m, m, k, order="F"), PHI=np.asarray(out5[4]), **stats)


GPZ_CASES = []
for _method in ("GL", "VL", "GD", "VD", "GC", "VC"):
    for _psi, _nan in ((False, 0.0), (True, 0.0), (False, 0.3), (True, 0.3)):
        GPZ_CASES.append((_method, 1, True, _psi, _nan))
GPZ_CASES += [("VD", 2, True, False, 0.0), ("VC", 2, True, True, 0.0), ("GL", 2, False, False, 0.3), ("VL", 1, False, True, 0.0),
              ("GC", 2, False, False, 0.0), ("GD", 3, True, True, 0.3)]


# inputs wider than the register-resident kernels of the HIP path (its MFMA pair kernels, its runtime-d kernels): (.., n, d, m)
GPZ_CASES += [("VC", 1, True, True, 0.0, 40, 13, 4), ("GC", 1, True, True, 0.3, 30, 22, 3), ("VD", 2, True, True, 0.0, 50, 24, 4),
              ("VC", 1, True, False, 0.3, 30, 21, 3), ("VC", 1, True, True, 0.0, 24, 34, 3)]


# per-output weights: omega n x k (what getOmega.m:19 returns for a k-column Y); (.., n, d, m, "omk").  omega(training) in the two RMSE
# statistics is then the FIRST column (GPz.m:236,258), omega(training,:) everywhere else (GPz.m:48,82,93,110,237,259)
GPZ_CASES += [("VD", 2, True, False, 0.0, 70, 3, 5, "omk"), ("VC", 2, True, False, 0.0, 70, 3, 5, "omk"),
              ("GC", 3, True, True, 0.3, 70, 3, 5, "omk"), ("VL", 2, False, True, 0.0, 70, 3, 5, "omk")]


def gpz_case_name(c):
    if len(c) > 8:
        return "ref_gpz_%s_k%d_h%d_p%d_n%d_omk" % (c[0], c[1], int(c[2]), int(c[3]), int(c[4] > 0))
    return "ref_gpz_%s_k%d_h%d_p%d_n%d" % (c[0], c[1], int(c[2]), int(c[3]), int(c[4] > 0)) + ("_d%d" % c[6] if len(c) > 5 else "")


def make_gpz(case, seed):
    method, k, hetero, psi, nanfrac = case[:5]
    n, d, m = case[5:8] if len(case) > 5 else (70, 3, 5)
    rng = np.random.default_rng(seed)
    pr = draw_problem(rng, n, d, m, k, method, hetero, psi, nanfrac, omega_cols=(k if len(case) > 8 else 1))
    out = run_gpz(ML.Interp(), method, m, d, k, hetero, pr)
    return dict(method=method, m=m, d=d, k=k, heteroscedastic=int(hetero), has_psi=int(psi),
                Psi=(pr["Psi"] if psi else np.zeros(0)), **{key: pr[key] for key in ("theta", "X", "Y", "omega", "training", "validation")},
                **out)


PREDICT_CASES = [(mth, noisy, nanfrac) for mth in ("GL", "VL", "GD", "VD", "GC", "VC") for noisy in (False, True) for nanfrac in (0.0, 0.35)]


PREDICT_CASES += [("VC", True, 0.0, 13), ("GC", True, 0.0, 22), ("VD", True, 0.35, 23)]   # (.., d): beyond the register kernels of the HIP path
PREDICT_CASES += [("VC", False, 0.08, 40)]   # predictCov.m:134-229 at d = 40: the 64-wide scratch kernels of the HIP path (k_pmiss_cov64.hip)


def predict_case_name(c):
    return "ref_predict_%s_p%d_n%d" % (c[0], int(c[1]), int(c[2] > 0)) + ("_d%d" % c[3] if len(c) > 3 else "")


def make_predict(case, seed):
    method, noisy, nanfrac = case[:3]
    n, d, m, k, ns = 70, (case[3] if len(case) > 3 else 3), 4, 2, 11
    rng = np.random.default_rng(seed)
    pr = draw_problem(rng, n, d, m, k, method, True, False, 0.0)
    ip = ML.Interp()
    ms0 = ML.model_struct(m, d, k, method, True, g_dim_of(method, m, d))
    # w, iSigma_w as train.m:53 obtains them: the reference's own solve-only call
    args = [ML.col(pr["theta"]), ms0, pr["X"], pr["Y"], E, E, E, E]
    out4 = ip.call("GPz", args, nargout=4)
    w, iS = np.asarray(out4[2]), np.asarray(out4[3]).reshape(m, m, k, order="F")
    muX = 0.1 * rng.standard_normal(d); sdX = 1.0 + 0.5 * rng.random(d); muY = rng.standard_normal(k)
    pri = rng.random(m) + 0.2
    pri = pri / pri.sum()
    theta = pr["theta"]
    md, gd = m * d, g_dim_of(method, m, d)
    P = theta[:md].reshape((m, d), order="F")
    v = theta[md + gd + m * k + k: md + gd + 2 * m * k + k].reshape((m, k), order="F")
    st = ML.Struct(theta=ML.col(theta), w=w, iSigma_w=iS, priors=pri.reshape(1, -1), P=P, v=v)
    ms = ML.model_struct(m, d, k, method, True, gd, muX=muX.reshape(1, -1), sdX=sdX.reshape(1, -1), muY=muY.reshape(1, -1), best=st, last=st)
    Xs = rng.standard_normal((ns, d)) * sdX + muX
    if nanfrac > 0:
        miss = rng.random((ns, d)) < nanfrac
        miss[miss.all(axis=1), 0] = False
        Xs[miss] = np.nan
    Psi = None
    if noisy:
        if method[1] == "C":
            Psi = np.zeros((d, d, ns))
            for i in range(ns):
                B = 0.3 * rng.standard_normal((d, d))
                Psi[:, :, i] = B @ B.T
        else:
            Psi = rng.gamma(1.0, 0.1, (ns, d))
    res = ip.call("predict", [Xs, ms] + (["Psi", Psi] if noisy else []), nargout=6)
    names = ("mu", "sigma", "nu", "beta_i", "gamma", "PHIs")
    return dict(method=method, m=m, d=d, k=k, has_psi=int(noisy), theta=theta, w=w, iSigma_w=iS, priors=pri, muX=muX, sdX=sdX, muY=muY,
                Xs=Xs, Psi=(Psi if noisy else np.zeros(0)), **{nm: np.asarray(r) for nm, r in zip(names, res)})


def make_misc(seed):
    rng = np.random.default_rng(seed)
    ip = ML.Interp()
    out = {}
    # Dxy.m
    A, B = rng.standard_normal((17, 4)), rng.standard_normal((6, 4))
    out.update(dxy_X=A, dxy_Y=B, dxy_D=np.asarray(ip.call("Dxy", [A, B], 1)[0]))
    # inv_logdet.m: regular, and rank-deficient (singular values below max(size)*eps(norm(s,inf)) are dropped)
    M = rng.standard_normal((9, 9)); S1 = M @ M.T + 0.5 * np.eye(9)
    Xi, ld = ip.call("inv_logdet", [S1], 2)
    out.update(il_A=S1, il_Xi=np.asarray(Xi), il_logdet=float(np.asarray(ld).reshape(-1)[0]))
    L = rng.standard_normal((9, 5)); S2 = L @ L.T
    Xi, ld = ip.call("inv_logdet", [S2], 2)
    out.update(il2_A=S2, il2_Xi=np.asarray(Xi), il2_logdet=float(np.asarray(ld).reshape(-1)[0]))
    # getOmega.m (with hist and Dxy.m) and fixPsi.m
    Yo = rng.gamma(2.0, 0.4, (150, 1))
    out.update(om_Y=Yo, om_balanced=np.asarray(ip.call("getOmega", [Yo, "balanced"], 1)[0]),
               om_balanced_w=np.asarray(ip.call("getOmega", [Yo, "balanced", ML.mat(0.05)], 1)[0]),
               om_normalized=np.asarray(ip.call("getOmega", [Yo, "normalized"], 1)[0]))
    sd = 1.0 + rng.random(3)
    fp_in = {"nd": rng.gamma(1.0, 0.1, (6, 3)), "n1": rng.gamma(1.0, 0.1, (6, 1))}
    cube = np.zeros((3, 3, 6))
    for i in range(6):
        Bq = 0.3 * rng.standard_normal((3, 3))
        cube[:, :, i] = Bq @ Bq.T
    fp_in["cube"] = cube
    out["fp_sdX"] = sd
    for key, val in fp_in.items():
        out["fp_in_" + key] = val
        for method in ("VD", "VC"):
            out["fp_%s_%s" % (key, method)] = np.asarray(ip.call("fixPsi", [val, ML.mat(6.0), sd.reshape(1, -1), method], 1)[0])
    # getPHI.m with all four outputs, and getPrior.m, for a diagonal and a covariance kind with input noise and missing values
    for tag, method in (("vd", "VD"), ("vc", "VC")):
        n, d, m, k = 40, 3, 4, 1
        pr = draw_problem(rng, n, d, m, k, method, True, True, 0.25)
        ms = ML.model_struct(m, d, k, method, True, g_dim_of(method, m, d))
        sel = rng.random(n) < 0.8
        PHI, Gam, lnb, N = ip.call("getPHI", [pr["X"], pr["Psi"], ML.col(pr["theta"]), ms, sel.reshape(-1, 1)], 4)
        prior = ip.call("getPrior", [pr["X"], pr["Psi"], ML.col(pr["theta"]), ms, sel.reshape(-1, 1)], 1)[0]
        out.update({tag + "_theta": pr["theta"], tag + "_X": pr["X"], tag + "_Psi": pr["Psi"], tag + "_sel": sel, tag + "_PHI": np.asarray(PHI),
                    tag + "_Gamma": np.asarray(Gam), tag + "_lnBeta_i": np.asarray(lnb), tag + "_N": np.asarray(N),
                    tag + "_prior": np.asarray(prior).reshape(-1)})
    return out


MINFUNC_DIR = "/root/reference/minFunc_2012/minFunc"


def make_lbfgs(seed, p=200, corr=5, steps=14):
    """lbfgsAdd.m / lbfgsProd.m (minFunc's L-BFGS memory: a wrapping ring with rejected pairs) in the layout of the mf_mem_* fixtures:
    gradients G, directions D, step lengths T -> added flags, ring state and the lbfgsProd direction after every call"""
    ip = ML.Interp(ref_dir=MINFUNC_DIR)
    rng = np.random.default_rng(seed)
    S = np.zeros((p, corr)); Y = np.zeros((p, corr)); YS = np.zeros((corr, 1))
    start, end, hd = 1.0, 0.0, 1.0
    G = np.zeros((steps + 1, p)); D = np.zeros((steps, p)); T = np.zeros(steps)
    added = np.zeros(steps, dtype=np.int32); dirs = np.zeros((steps, p))
    starts = np.zeros(steps, dtype=np.int32); ends = np.zeros(steps, dtype=np.int32); hds = np.zeros(steps)
    G[0] = rng.standard_normal(p)
    Hm = np.diag(0.5 + rng.random(p))
    for it in range(steps):
        D[it] = rng.standard_normal(p)
        T[it] = 10.0 ** rng.uniform(-2, 0.5)
        s = T[it] * D[it]
        y = Hm @ s if it not in (4, 9) else -0.3 * s            # two pairs with y's <= 1e-10: rejected (lbfgsAdd.m:5,30)
        G[it + 1] = G[it] + y
        out = ip.call("lbfgsAdd", [ML.col(y), ML.col(s), S, Y, YS, ML.mat(start), ML.mat(end), ML.mat(hd), ML.mat(0.0)], 7)
        S, Y, YS = (np.asarray(o) for o in out[:3])
        start, end, hd, skipped = (float(np.asarray(o).reshape(-1)[0]) for o in out[3:])
        added[it] = 0 if skipped else 1
        starts[it], ends[it], hds[it] = int(start), int(end), hd
        if end > 0:
            dirs[it] = np.asarray(ip.call("lbfgsProd", [ML.col(G[it + 1]), S, Y, YS, ML.mat(start), ML.mat(end), ML.mat(hd)], 1)[0]).reshape(-1)
    return dict(p=p, corrections=corr, G=G, D=D, T=T, added=added, directions=dirs, starts=starts, ends=ends, Hdiag=hds, S=S, Y=Y,
                YS=YS.reshape(-1))


def full_interp(U=None, log=None):
    """GPz/ and minFunc/ on one path, as startup.m / demo_*.m set it up (addpath).  The two compiled MEX files minFunc calls by
    default (useMex = 1, train.m sets no option) are stood in for by what minFunc ships as their MATLAB twins: lbfgsAddC(y,s,Y,S,ys,end)
    stores the two columns in place (lbfgsAdd.m:22-23 is the same statement pair), lbfgsProdC is lbfgsProd.m executed.
    rand() returns the recorded uniform matrix U (init.m:58); fprintf records its arguments (callBack.m's per-iteration line)."""
    ip = ML.Interp(ref_dir=[ML.REF_DIR, MINFUNC_DIR])

    def add_c(a, nargout):
        y, s, Y, S, _, end = a
        e = int(np.asarray(end).reshape(-1)[0]) - 1
        S[:, e] = np.asarray(s).reshape(-1)
        Y[:, e] = np.asarray(y).reshape(-1)
        return []
    ip.extern["lbfgsAddC"] = add_c
    ip.extern["lbfgsProdC"] = lambda a, nargout: ip.call("lbfgsProd", a, 1)
    if U is not None:
        ip.extern["rand"] = lambda a, nargout: [np.array(U)]
    if log is not None:
        ip.extern["fprintf"] = lambda a, nargout: (log.append([a[0]] + [float(np.asarray(v).reshape(-1)[0]) for v in a[1:]]), [])[1]
    return ip


def py_handle(fun):
    """a Python objective fun(x) -> (f, g) as the function handle minFunc.m calls as [f,g] = funObj(x)"""
    def h(args, nargout):
        f, g = fun(np.asarray(args[0]).reshape(-1).copy())
        return [ML.mat(f), np.asarray(g, dtype=np.float64).reshape(-1, 1)]
    return h


def make_minfunc():
    """WolfeLineSearch.m / ArmijoBacktrack.m (with polyinterp.m, isLegal.m) and whole minFunc.m runs, executed on the INPUTS of the
    optimiser fixtures tests/golden/mf_ls_*.npz and mf_run_*.npz (analytic objectives of tests/minfunc_objectives.py; for
    mf_run_gpz_* the objective is GPz.m itself, executed).  One file, keys <fixture>__<quantity>: what the reference returns for
    the inputs those fixtures record - the restated optimiser that produced the mf_* files is checked against it."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import minfunc_objectives as F
    m = ML.mat
    out = {}
    for path in sorted(glob.glob(os.path.join(GOLD, "mf_ls_*.npz"))):
        z, name = np.load(path), os.path.splitext(os.path.basename(path))[0]
        fun = F.OBJECTIVES[str(z["objective"])]
        x, d = z["x"], z["d"]
        f, g = fun(x)
        common = [m(float(z["c1"]))] + ([m(float(z["c2"]))] if str(z["kind"]) == "wolfe" else []) + [m(float(z["ls_interp"])), m(0.0)]
        ip = full_interp()
        if str(z["kind"]) == "wolfe":       # WolfeLineSearch(x,t,d,f,g,gtd,c1,c2,LS_interp,LS_multi,maxLS,progTol,debug,doPlot,saveHessianComp,funObj)
            t, fn, gn, ev = ip.call("WolfeLineSearch", [ML.col(x), m(float(z["t0"])), ML.col(d), m(f), ML.col(g), m(float(g @ d))] + common
                                    + [m(25.0), m(1e-9), m(0.0), m(0.0), m(0.0), py_handle(fun)], 4)
        else:                               # ArmijoBacktrack(x,t,d,f,fr,g,gtd,c1,LS_interp,LS_multi,progTol,debug,doPlot,saveHessianComp,funObj)
            t, _, fn, gn, ev = ip.call("ArmijoBacktrack", [ML.col(x), m(float(z["t0"])), ML.col(d), m(f), m(f), ML.col(g), m(float(g @ d))]
                                       + common + [m(1e-9), m(0.0), m(0.0), m(0.0), py_handle(fun)], 5)
        out.update({name + "__t": float(np.asarray(t).reshape(-1)[0]), name + "__f_new": float(np.asarray(fn).reshape(-1)[0]),
                    name + "__g_new": np.asarray(gn).reshape(-1), name + "__funEvals": int(np.asarray(ev).reshape(-1)[0])})
    for path in sorted(glob.glob(os.path.join(GOLD, "mf_run_*.npz"))):
        z, name = np.load(path), os.path.splitext(os.path.basename(path))[0]
        ip = full_interp()
        if str(z["objective"]) == "gpz":
            mm, d = int(z["m"]), int(z["d"])
            ms = ML.model_struct(mm, d, 1, str(z["method"]), True, g_dim_of(str(z["method"]), mm, d))
            X, Y = z["X"], z["Y"]
            handle = lambda a, nargout: ip.call("GPz", [a[0], ms, X, Y, E, E, E, E], 2)
        else:
            handle = py_handle(F.OBJECTIVES[str(z["objective"])])
        steps = []
        opt = ML.Struct(Method="lbfgs", Display="off", MaxIter=m(float(z["max_iter"])),
                        outputFcn=lambda a, nargout: (steps.append(float(np.asarray(a[5]).reshape(-1)[0])) if a[1] == "iter" else None,
                                                      [m(False)])[1])
        if name.endswith("_c5"):
            opt.Corr = m(5.0)
        x, f, flag, o = ip.call("minFunc", [handle, ML.col(z["x0"]), opt], 4)
        out.update({name + "__x": np.asarray(x).reshape(-1), name + "__f": float(np.asarray(f).reshape(-1)[0]),
                    name + "__exitflag": int(np.asarray(flag).reshape(-1)[0]), name + "__message": str(o.message),
                    name + "__iterations": int(np.asarray(o.iterations).reshape(-1)[0]),
                    name + "__funcCount": int(np.asarray(o.funcCount).reshape(-1)[0]),
                    name + "__fval": np.asarray(o.trace.fval).reshape(-1), name + "__funcCounts": np.asarray(o.trace.funcCount).reshape(-1),
                    name + "__optCond": np.asarray(o.trace.optCond).reshape(-1), name + "__steps": np.array(steps)})
    return out


# (name, method, n, d, m, k, heteroscedastic, validation, nan fraction, Psi, omega, maxIter, maxAttempts)
TRAIN_CASES = [("VD_valid", "VD", 150, 3, 6, 1, True, True, 0.0, False, None, 20, np.inf),
               ("GC_nan_omega", "GC", 160, 3, 5, 1, True, True, 0.15, False, "normalized", 15, np.inf),
               ("VC_psi_trainonly", "VC", 40, 2, 3, 1, True, False, 0.0, True, None, 6, np.inf),
               ("GL_k2_attempts", "GL", 140, 3, 5, 2, True, True, 0.0, False, None, 40, 2),
               ("VL_d1_homo", "VL", 130, 1, 7, 1, False, True, 0.0, False, "balanced", 15, np.inf)]


def make_train(case, seed):
    """init.m -> train.m (minFunc.m, WolfeLineSearch.m, polyinterp.m, lbfgsAdd.m, lbfgsProd.m, callBack.m, GPz.m, getPHI.m, getPrior.m,
    pca.m, fillLinear.m, Dxy.m, fixPsi.m, getOmega.m): the model after init, callBack's line per iteration, the model after train"""
    name, method, n, d, m, k, hetero, valid, nanfrac, psi, omega_kind, max_iter, max_att = case
    rng = np.random.default_rng(seed)
    scale, shift = 0.5 + 2.0 * rng.random(d), rng.standard_normal(d)
    X = rng.standard_normal((n, d)) * scale + shift
    Y = np.sin((X - shift) / scale @ rng.standard_normal((d, k))) + 0.1 * rng.standard_normal((n, k)) + 0.7
    if omega_kind:
        Y = np.abs(Y) + 0.05
    U = rng.random((m, d))
    training = rng.random(n) < 0.7
    validation = ~training
    Psi = rng.gamma(1.0, 0.05, (n, d)) * scale ** 2 if psi else None
    if nanfrac > 0:
        miss = rng.random((n, d)) < nanfrac
        miss[miss.all(axis=1), 0] = False
        X[miss] = np.nan
    log = []
    ip = full_interp(U, log)
    opts = ["heteroscedastic", ML.mat(bool(hetero)), "training", training.reshape(-1, 1)]
    topts = ["maxIter", ML.mat(float(max_iter)), "maxAttempts", ML.mat(float(max_att)), "training", training.reshape(-1, 1)]
    omega = None
    if omega_kind:
        omega = np.asarray(ip.call("getOmega", [Y[:, :1], omega_kind], 1)[0])
        opts += ["omega", omega]
        topts += ["omega", omega]
    if psi:
        opts += ["Psi", Psi]
        topts += ["Psi", Psi]
    if valid:
        topts += ["validation", validation.reshape(-1, 1)]
    model = ip.call("init", [X, Y, method, ML.mat(float(m))] + opts, 1)[0]
    out = dict(method=method, method_after_init=model.method, m=m, d=d, k=k, heteroscedastic=int(hetero), X=X, Y=Y, U=U, training=training,
               validation=(validation if valid else np.zeros(0, dtype=bool)), omega=(omega if omega is not None else np.zeros(0)),
               Psi=(Psi if psi else np.zeros(0)), maxIter=max_iter, maxAttempts=max_att, muX=np.asarray(model.muX), sdX=np.asarray(model.sdX),
               muY=np.asarray(model.muY), g_dim=int(np.asarray(model.g_dim).reshape(-1)[0]), theta0=np.asarray(model.last.theta).reshape(-1),
               w0=np.asarray(model.last.w), iSigma_w0=np.asarray(model.last.iSigma_w).reshape(m, m, k, order="F"))
    del log[:]
    model = ip.call("train", [model, X, Y] + topts, 1)[0]
    rows = [r[1:6] + [r[6] if len(r) > 7 else np.nan] for r in log if r[0].startswith("\\t%d")]
    rows = [[r[0], r[1], r[2], r[3], r[4] if valid else np.nan, r[5] if valid else np.nan] for r in rows]
    out.update(log=np.array(rows), message=[r[0] for r in log][-1])
    for which in ("last", "best"):
        st = getattr(model, which)
        out.update({which + "_theta": np.asarray(st.theta).reshape(-1), which + "_w": np.asarray(st.w),
                    which + "_iSigma_w": np.asarray(st.iSigma_w).reshape(m, m, k, order="F"),
                    which + "_priors": np.asarray(st.priors).reshape(-1)})
    return out


def make_demo_sinc(seed=1, max_iter=40):
    """demo_sinc.m as shipped (the reference's own CPU-runnable case, BASELINE config 1): n = 10000 points on [-10, 10] minus the gap
    (-7, -2), y = sinc(x) + heteroscedastic noise, Gamma-distributed input-noise variances, x observed through that noise,
    sample(n, 0.7, 0.15, 0.15), init(X, Y, 'VL', 100, ...), train(..., 'maxAttempts', 50, ...), predict on a grid and on the test
    rows, RMSE and mean log-likelihood as the demo prints them (demo_sinc.m:7-32,36-66,71,104-122).  The draws come from NumPy
    (randn / gamrnd / randperm / rand are handed the recorded arrays); maxIter is 40 instead of the demo's 500 to keep the
    comparison on the part of the trajectory where two implementations still walk together."""
    rng = np.random.default_rng(seed)
    m, method = 100, "VL"
    x = np.linspace(-10.0, 10.0, 10000)
    x = x[(x < -7.0) | (x > -2.0)]
    n = x.size
    sx = 0.05 + (1.0 / (1.0 + np.exp(-0.2 * x))) * (1.0 + np.sin(2.0 * x)) * 0.2
    Y = (np.sinc(x / np.pi) + rng.standard_normal(n) * sx).reshape(-1, 1)
    E_, V_ = 0.5, 0.25
    Psi = rng.gamma(E_ ** 2 / V_, V_ / E_, n).reshape(-1, 1)
    X = (x + rng.standard_normal(n) * np.sqrt(Psi[:, 0])).reshape(-1, 1)
    perm = rng.permutation(n) + 1.0
    U = rng.random((m, 1))
    log = []
    ip = full_interp(U, log)
    ip.extern["randperm"] = lambda a, nargout: [perm.reshape(1, -1).copy()]
    tr, va, te = (np.asarray(q).astype(bool).reshape(-1) for q in
                  ip.call("sample", [ML.mat(float(n)), ML.mat(0.7), ML.mat(0.15), ML.mat(0.15)], 3))
    model = ip.call("init", [X, Y, method, ML.mat(float(m)), "normalize", ML.mat(True), "heteroscedastic", ML.mat(True),
                             "training", tr.reshape(-1, 1), "Psi", Psi], 1)[0]
    k = 1
    out = dict(method=method, method_after_init=model.method, m=m, d=1, k=k, heteroscedastic=1, X=X, Y=Y, U=U, perm=perm, training=tr,
               validation=va, testing=te, omega=np.zeros(0), Psi=Psi, maxIter=max_iter, maxAttempts=50.0, muX=np.asarray(model.muX),
               sdX=np.asarray(model.sdX), muY=np.asarray(model.muY), g_dim=int(np.asarray(model.g_dim).reshape(-1)[0]),
               theta0=np.asarray(model.last.theta).reshape(-1), w0=np.asarray(model.last.w),
               iSigma_w0=np.asarray(model.last.iSigma_w).reshape(m, m, k, order="F"))
    del log[:]
    model = ip.call("train", [model, X, Y, "maxIter", ML.mat(float(max_iter)), "maxAttempts", ML.mat(50.0), "training", tr.reshape(-1, 1),
                              "validation", va.reshape(-1, 1), "Psi", Psi], 1)[0]
    rows = [r[1:7] for r in log if r[0].startswith("\\t%d")]
    out.update(log=np.array(rows), message=[r[0] for r in log][-1])
    for which in ("last", "best"):
        st = getattr(model, which)
        out.update({which + "_theta": np.asarray(st.theta).reshape(-1), which + "_w": np.asarray(st.w),
                    which + "_iSigma_w": np.asarray(st.iSigma_w).reshape(m, m, k, order="F"),
                    which + "_priors": np.asarray(st.priors).reshape(-1)})
    Xs = np.linspace(-15.0, 15.0, 200).reshape(-1, 1)
    grid = ip.call("predict", [Xs, model], 5)                                                    # demo_sinc.m:71
    out.update(Xs=Xs, **{"grid_" + nm: np.asarray(v) for nm, v in zip(("mu", "sigma", "nu", "beta_i", "gamma"), grid)})
    mu, sigma = (np.asarray(v) for v in ip.call("predict", [X, model, "Psi", Psi, "selection", te.reshape(-1, 1)], 2))   # :104
    err = Y[te] - mu
    out.update(test_mu=mu, test_sigma=sigma, rmse=float(np.sqrt(np.mean(err ** 2))),
               mll=float(np.mean(-0.5 * err ** 2 / sigma - 0.5 * np.log(sigma)) - 0.5 * np.log(2 * np.pi)))      # :117-122
    return out


def make_demo_2D(seed=2, max_iter=40):
    """demo_2D.m's configuration (the reference's own end-to-end use of missing values: VD, m = 50, three Gaussian bumps in 2-D,
    n = 3000, Gamma-distributed input-noise variances, the first variable removed from a quarter of the rows and the second from
    another quarter, demo_2D.m:7-25,28-89): sample(n, 0.7, 0.15, 0.15), init(Xn, Y, 'VD', 50, ..., 'Psi', Psi) with NaNs in Xn,
    train(..., 'maxAttempt', 50, ...) - the demo's own spelling, which parseArgs resolves as a unique prefix of 'maxAttempts' -
    then predict on a grid without missing values (:96-100), predict with only ONE variable observed (:127-138) and the test-set
    error with the other variable missing (:155-159).  The draws come from NumPy and are handed over as recorded arrays; maxIter
    is 40 instead of 500, the grids are coarser than the demo's (30 x 30 and 200 points) to keep the interpreted run short; the
    single-variable reference models of :161-178 are the demo's comparison baseline, not the path under test, and are not trained."""
    rng = np.random.default_rng(seed)
    m, method = 50, "VD"
    means = [np.array([10.0, 0.0]), np.array([10.0, 10.0]), np.array([5.0, 5.0])]
    covs = [np.array([[10.0, 0.0], [0.0, 1.0]]), np.array([[5.0, -3.0], [-3.0, 3.0]]), np.array([[2.0, 0.0], [0.0, 2.0]])]
    X = np.vstack([rng.multivariate_normal(mu, S, 1000) for mu, S in zip(means, covs)])
    n, d = X.shape

    def mvnpdf(Z, mu, S):
        dz = Z - mu
        return np.exp(-0.5 * np.einsum("ij,jk,ik->i", dz, np.linalg.inv(S), dz)) / (2.0 * np.pi * np.sqrt(np.linalg.det(S)))
    w_true = np.array([-9.0, 6.0, 3.0])
    Y = (np.column_stack([mvnpdf(X, mu, S) for mu, S in zip(means, covs)]) @ w_true + 0.01 * rng.standard_normal(n)).reshape(-1, 1)
    E_, V_ = 0.5, 0.25
    Psi = rng.gamma(E_ ** 2 / V_, V_ / E_, (n, d))
    Xn = X + rng.standard_normal((n, d)) * np.sqrt(Psi)
    r = rng.permutation(n)
    psize = int(np.ceil(0.5 * n / 2))
    Xn[r[:psize], 0] = np.nan
    Xn[r[psize:2 * psize], 1] = np.nan
    perm = rng.permutation(n) + 1.0
    U = rng.random((m, d))
    log = []
    ip = full_interp(U, log)
    ip.extern["randperm"] = lambda a, nargout: [perm.reshape(1, -1).copy()]
    tr, va, te = (np.asarray(q).astype(bool).reshape(-1) for q in
                  ip.call("sample", [ML.mat(float(n)), ML.mat(0.7), ML.mat(0.15), ML.mat(0.15)], 3))
    model = ip.call("init", [Xn, Y, method, ML.mat(float(m)), "heteroscedastic", ML.mat(True), "normalize", ML.mat(True),
                             "training", tr.reshape(-1, 1), "Psi", Psi], 1)[0]
    k = 1
    out = dict(method=method, method_after_init=model.method, m=m, d=d, k=k, heteroscedastic=1, X=Xn, Xclean=X, Y=Y, U=U, perm=perm,
               training=tr, validation=va, testing=te, omega=np.zeros(0), Psi=Psi, maxIter=max_iter, maxAttempts=50.0,
               muX=np.asarray(model.muX), sdX=np.asarray(model.sdX), muY=np.asarray(model.muY),
               g_dim=int(np.asarray(model.g_dim).reshape(-1)[0]), theta0=np.asarray(model.last.theta).reshape(-1),
               w0=np.asarray(model.last.w), iSigma_w0=np.asarray(model.last.iSigma_w).reshape(m, m, k, order="F"))
    del log[:]
    model = ip.call("train", [model, Xn, Y, "maxIter", ML.mat(float(max_iter)), "maxAttempt", ML.mat(50.0), "training", tr.reshape(-1, 1),
                              "validation", va.reshape(-1, 1), "Psi", Psi], 1)[0]
    rows = [q[1:7] for q in log if q[0].startswith("\\t%d")]
    out.update(log=np.array(rows), message=[q[0] for q in log][-1])
    for which in ("last", "best"):
        st = getattr(model, which)
        out.update({which + "_theta": np.asarray(st.theta).reshape(-1), which + "_w": np.asarray(st.w),
                    which + "_iSigma_w": np.asarray(st.iSigma_w).reshape(m, m, k, order="F"),
                    which + "_priors": np.asarray(st.priors).reshape(-1)})
    gx, gy = np.meshgrid(np.linspace(X[:, 0].min() - 1, X[:, 0].max() + 1, 30), np.linspace(X[:, 1].min() - 1, X[:, 1].max() + 1, 30))
    Xs = np.column_stack([gx.reshape(-1, order="F"), gy.reshape(-1, order="F")])                 # demo_2D.m:96-97
    grid = ip.call("predict", [Xs, model], 5)                                                    # :100
    out.update(Xs=Xs, **{"grid_" + nm: np.asarray(v) for nm, v in zip(("mu", "sigma", "nu", "beta_i", "gamma"), grid)})
    rmses = np.zeros(2)
    for o in range(2):
        rng_o = X[:, o].max() - X[:, o].min()
        Xo = np.linspace(X[:, o].min() - rng_o / 10, X[:, o].max() + rng_o / 10, 200)
        Xm = np.full((Xo.size, 2), np.nan)
        Xm[:, o] = Xo                                                                            # :132-136
        res = ip.call("predict", [Xm, model], 5)
        out.update({"only%d_X" % o: Xm}, **{"only%d_%s" % (o, nm): np.asarray(v) for nm, v in zip(("mu", "sigma", "nu", "beta_i", "gamma"), res)})
        Xt = np.full((int(te.sum()), 2), np.nan)
        Xt[:, o] = X[te, o]                                                                      # :155-158
        mu = np.asarray(ip.call("predict", [Xt, model], 1)[0])
        rmses[o] = float(np.sqrt(np.mean((Y[te] - mu) ** 2)))                                    # :159
        out.update({"test%d_X" % o: Xt, "test%d_mu" % o: mu})
    out["rmses_predicted"] = rmses
    return out


def all_fixtures():
    """name -> maker()"""
    fx = {}
    for q, c in enumerate(GPZ_CASES):
        fx[gpz_case_name(c)] = (lambda c=c, q=q: make_gpz(c, 500 + q))
    for q, c in enumerate(PREDICT_CASES):
        fx[predict_case_name(c)] = (lambda c=c, q=q: make_predict(c, 800 + q))
    fx["ref_misc"] = lambda: make_misc(77)
    fx["ref_lbfgs_mem"] = lambda: make_lbfgs(91)
    fx["ref_minfunc"] = make_minfunc
    for i, c in enumerate(TRAIN_CASES):
        fx["ref_train_" + c[0]] = (lambda c=c, i=i: make_train(c, 400 + i))
    fx["ref_train_demo_sinc"] = make_demo_sinc
    fx["ref_train_demo_2D"] = make_demo_2D
    return fx


def main():
    if not ML.available():
        raise SystemExit("the reference tree is not present: fixtures can only be generated where /root/reference exists")
    os.makedirs(GOLD, exist_ok=True)
    only = sys.argv[1] if len(sys.argv) > 1 else ""      # optional: regenerate only the fixtures whose name contains this
    for name, make in all_fixtures().items():
        if only not in name:
            continue
        data = make()
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **data)
        print("wrote", name, flush=True)
    # which builtins of the interpreter the executed files reached (calls each): the list tests/test_reference_run.py pins one by one
    # against MATLAB's documentation
    import json
    if only:
        return
    with open(os.path.join(GOLD, "mlite_builtins_reached.json"), "w") as fh:
        json.dump(dict(sorted(ML.BUILTINS_REACHED.items())), fh, indent=0)


if __name__ == "__main__":
    main()
