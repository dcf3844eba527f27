"""Timing of the general GC/VC path (Psi cube / missing values) on a mid-size problem (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpz_amd
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 10
m = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cfg = dict(n=n, d=d, m=m, method="VC", omega=None)
model, theta, X, y, _ = bench.synth(cfg)
rng = np.random.default_rng(5)
Psi = np.zeros((d, d, n))
diag = rng.gamma(1.0, 0.05, (n, d))
Psi[np.arange(d), np.arange(d), :] = diag.T
full = len(sys.argv) > 5 and sys.argv[5] == "full"
if full:   # full (non-diagonal) cubes
    B = 0.05 * rng.standard_normal((n, d, d))
    Psi = Psi + np.einsum("nab,ncb->acn", B, B)
runs = [("plain", {}), ("psi f64", {"Psi": Psi}), ("psi f32", {"Psi": Psi, "dtype": "f32"})]
if d > 10 and n * m > 5e6: runs = [runs[0], runs[2]]     # the fp64 general kernels take seconds at this size
for name, kw in runs:
    ctx = gpz_amd.GPzContext(model, X, y, **kw)
    ctx.eval(theta)
    ctx.enable_timing(True); ctx.reset_timings()
    t0 = time.perf_counter(); K = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    for _ in range(K):
        f, g = ctx.eval(theta)
    dt = (time.perf_counter() - t0) / K
    tim = ctx.timings()
    print(name, "ms/eval %.2f" % (dt * 1e3), "f=%.6f" % f, " ".join("%s=%.2f" % (k, v[0] / K) for k, v in sorted(tim.items(), key=lambda x: -x[1][0])[:5]))
    ctx.close()
