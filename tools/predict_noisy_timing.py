"""Developer tool: wall time of predict() with input noise (no missing values).  usage: predict_noisy_timing.py [ns] [m] [d] [method]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpz_amd
from helpers import make_problem
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 500
d = int(sys.argv[3]) if len(sys.argv) > 3 else 10
method = sys.argv[4] if len(sys.argv) > 4 else "GC"
model, theta, X, Y, _, rng = make_problem(2000, d, m, 1, method, True, seed=7)
ctx = gpz_amd.GPzContext(model, X, Y)
w, iS, _ = ctx.solve(theta)
ctx.close()
model.sets = {"best": {"theta": theta, "w": w, "iSigma_w": iS}}
Xs = rng.standard_normal((ns, d))
diag = rng.gamma(1.0, 0.05, (ns, d))
if method[1] == "C":
    Psi = np.zeros((d, d, ns)); Psi[np.arange(d), np.arange(d), :] = diag.T
else:
    Psi = diag
gpz_amd.predict(Xs[:64], model, Psi=Psi[:, :, :64] if Psi.ndim == 3 else Psi[:64])
for name, PP in (("plain", None), ("psi", Psi)):
    t0 = time.perf_counter()
    out = gpz_amd.predict(Xs, model, Psi=PP)
    print(f"{method} {name} ns={ns} m={m} d={d}: {(time.perf_counter() - t0) * 1e3:.1f} ms finite={bool(np.isfinite(out[0]).all())}", flush=True)
