"""Developer tool: wall time of predict() over methods x {plain, input noise, missing values, both}.
usage: sweep_predict.py [ns] [m] [d]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpz_amd
from helpers import make_problem
from oracle import gpz_oracle as O

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d = int(sys.argv[3]) if len(sys.argv) > 3 else 6
for method in ["VD", "GC", "VC"]:
    model, theta, X, Y, _, rng = make_problem(2000, d, m, 1, method, True, seed=7)
    ctx = gpz_amd.GPzContext(model, X, Y)
    w, iS, _ = ctx.solve(theta)
    ctx.close()
    model.sets = {"best": {"theta": theta, "w": w, "iSigma_w": iS}}
    Xs = rng.standard_normal((ns, d))
    Xn = Xs.copy(); Xn[rng.random((ns, d)) < 0.05] = np.nan; Xn[:, 0] = Xs[:, 0]
    diag = rng.gamma(1.0, 0.05, (ns, d))
    if method[1] == "C":
        Psi = np.zeros((d, d, ns)); Psi[np.arange(d), np.arange(d), :] = diag.T
    else:
        Psi = diag
    for name, XX, PP in [("plain", Xs, None), ("psi", Xs, Psi), ("nan", Xn, None), ("psi+nan", Xn, Psi)]:
        try:
            gpz_amd.predict(XX[:64], model, Psi=None if PP is None else (PP[:, :, :64] if PP.ndim == 3 else PP[:64]))
            t0 = time.perf_counter()
            out = gpz_amd.predict(XX, model, Psi=PP)
            dt = time.perf_counter() - t0
            print("%s %-8s ns=%d m=%d d=%d  %9.1f ms  finite=%s" % (method, name, ns, m, d, dt * 1e3, bool(np.isfinite(out[0]).all())), flush=True)
        except Exception as e:
            print(method, name, "ERR", str(e)[:120], flush=True)
