"""Timing of the diagonal kinds with input noise / missing values against the plain path (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpz_amd
import bench

n, d, m = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 10, 200
cfg = dict(n=n, d=d, m=m, method="VD", omega=None)
model, theta, X, y, _ = bench.synth(cfg)
rng = np.random.default_rng(5)
Psi = rng.gamma(1.0, 0.05, (n, d))
Xn = X.copy(); Xn[rng.random((n, d)) < 0.05] = np.nan
for name, Xi, kw in (("plain", X, {}), ("psi", X, {"Psi": Psi}), ("nan", Xn, {}), ("psi+nan", Xn, {"Psi": Psi})):
    ctx = gpz_amd.GPzContext(model, Xi, y, **kw)
    ctx.eval(theta)
    ctx.enable_timing(True); ctx.reset_timings()
    t0 = time.perf_counter(); K = 5
    for _ in range(K):
        f, g = ctx.eval(theta)
    dt = (time.perf_counter() - t0) / K
    tim = ctx.timings()
    print(name, "ms/eval %.2f" % (dt * 1e3), "f=%.6f" % f, " ".join("%s=%.2f" % (k, v[0] / K) for k, v in sorted(tim.items(), key=lambda x: -x[1][0])[:5]))
    ctx.close()
