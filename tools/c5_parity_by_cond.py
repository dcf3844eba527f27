import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, gpz_amd, bench
from oracle import gpz_oracle as O
cfg = dict(bench.CONFIGS["c5"]); cfg["n"] = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["n"]; rows = 60
model, theta, X, y, omega = bench.synth(cfg)
X, y = X[:rows], y[:rows]; Psi = bench.synth_psi(cfg, np.arange(rows), cube=True)
om = O.Model(m=model.m, d=model.d, k=1, method=model.method, heteroscedastic=True)
P, G, *_ = O.unpack_theta(theta, om); Gm = O.expand_gamma(G, om)
cond = np.array([np.linalg.cond(Gm[:, :, j].T @ Gm[:, :, j]) for j in range(model.m)])
ref = O.GPz(theta, om, X, y, Psi)
c32 = gpz_amd.GPzContext(model, X, y, Psi, dtype="f32"); f32, g32 = c32.eval(theta); c32.close()
c64 = gpz_amd.GPzContext(model, X, y, Psi); f64, g64 = c64.eval(theta); c64.close()
m, d = model.m, model.d; md = m * d
mx = np.abs(ref.grad).max()
def blk(g): return g[md:md + d * d * m].reshape((d, d, m), order="F")
e32 = np.abs(blk(g32) - blk(ref.grad)).max(axis=(0, 1)) / mx
e64 = np.abs(blk(g64) - blk(ref.grad)).max(axis=(0, 1)) / mx
e3264 = np.abs(blk(g32) - blk(g64)).max(axis=(0, 1)) / mx
print("max|g| %.3e; cond percentiles 50/90/99/100: %s" % (mx, " ".join("%.1e" % v for v in np.percentile(cond, [50, 90, 99, 100]))))
for lo, hi in ((0, 1e4), (1e4, 1e6), (1e6, 1e8), (1e8, 1e30)):
    sel = (cond >= lo) & (cond < hi)
    if sel.any():
        print(f"cond in [{lo:.0e},{hi:.0e}): {sel.sum():4d} bases  max err/max|g|: f32 vs oracle {e32[sel].max():.1e}  fp64 path vs oracle {e64[sel].max():.1e}  f32 vs fp64 path {e3264[sel].max():.1e}")
rest = np.r_[0:md, md + d * d * m:theta.size]
print("other blocks: f32 vs oracle %.1e, fp64 path vs oracle %.1e" % (np.abs(g32[rest] - ref.grad[rest]).max() / mx, np.abs(g64[rest] - ref.grad[rest]).max() / mx))
# the worst entry: who is right?  central differences of the (fp64 HIP) objective, which agrees with the oracle's to 1e-9
c64 = gpz_amd.GPzContext(model, X, y, Psi)
i = int(np.argmax(np.abs(g32 - ref.grad)))
for h in (1e-4, 1e-5, 1e-6):
    tp = theta.copy(); tp[i] += h; tm = theta.copy(); tm[i] -= h
    fd = (c64.eval(tp)[0] - c64.eval(tm)[0]) / (2 * h)
    fdo = (O.GPz(tp, om, X, y, Psi).nlogML - O.GPz(tm, om, X, y, Psi).nlogML) / (2 * h)
    print(f"worst entry theta[{i}] (basis {(i - md) // (d * d)}, cond {cond[(i - md) // (d * d)]:.1e}) h={h:.0e}: FD(fp64 HIP f) {fd:+.6e} FD(oracle f) {fdo:+.6e} | oracle grad {ref.grad[i]:+.6e}  fp64 path {g64[i]:+.6e}  f32 whitened path {g32[i]:+.6e}")
c64.close()
