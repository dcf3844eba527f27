"""Developer tool: one predict() call with missing values (diag kinds) for rocprofv3 --stats.
usage: predict_missing_profile.py [ns] [m] [d] [method]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpz_amd
from helpers import make_problem
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 500
d = int(sys.argv[3]) if len(sys.argv) > 3 else 10
method = sys.argv[4] if len(sys.argv) > 4 else "VD"
model, theta, X, Y, _, rng = make_problem(2000, d, m, 1, method, True, seed=7)
ctx = gpz_amd.GPzContext(model, X, Y)
w, iS, _ = ctx.solve(theta)
ctx.close()
model.sets = {"best": {"theta": theta, "w": w, "iSigma_w": iS}}
Xs = rng.standard_normal((ns, d))
Xn = Xs.copy(); Xn[rng.random((ns, d)) < 0.05] = np.nan; Xn[:, 0] = Xs[:, 0]
npat = len({tuple(r) for r in np.isnan(Xn)})
gpz_amd.predict(Xn[:64], model)
t0 = time.perf_counter()
out = gpz_amd.predict(Xn, model)
dt = time.perf_counter() - t0
print(f"{method} ns={ns} m={m} d={d} patterns={npat}: {dt * 1e3:.1f} ms  ({dt * 1e3 / npat:.2f} ms per pattern)  finite={bool(np.isfinite(out[0]).all())}")
