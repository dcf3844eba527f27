"""Developer check: at config 5's shape and bench.py's own theta (Gamma_j = gamma_j I + 0.05 N(0,1): cond(Gamma'Gamma) is
huge at d = 20) which gradient is right — the oracle's (reference formula through Sigma = inv(Gamma'Gamma)), the fp64
general kernels', or the whitened fp32 path's?  Directional derivatives against central differences of the objective."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, gpz_amd, bench
from oracle import gpz_oracle as O
cfg = dict(bench.CONFIGS["c5"]); cfg["n"] = 400; cfg["m"] = 200
model, theta, X, y, omega = bench.synth(cfg)
Psi = bench.synth_psi(cfg, np.arange(cfg["n"]), cube=True)
om = O.Model(m=model.m, d=model.d, k=1, method=model.method, heteroscedastic=True)
P, G, *_ = O.unpack_theta(theta, om); Gm = O.expand_gamma(G, om)
print("cond(Gamma_j'Gamma_j): median %.1e max %.1e" % tuple(np.percentile([np.linalg.cond(Gm[:, :, j].T @ Gm[:, :, j]) for j in range(model.m)], [50, 100])))
ref = O.GPz(theta, om, X, y, Psi)
c64 = gpz_amd.GPzContext(model, X, y, Psi); f64, g64 = c64.eval(theta)
c32 = gpz_amd.GPzContext(model, X, y, Psi, dtype="f32"); f32, g32 = c32.eval(theta)
print("f: oracle %.12f  fp64 path %.12f  f32 path %.12f" % (ref.nlogML, f64, f32))
rng = np.random.default_rng(0)
md = model.m * model.d; gd = om.g_dim
for name, sl in (("dGamma block", slice(md, md + gd)), ("all", slice(0, theta.size))):
    v = np.zeros(theta.size); v[sl] = rng.standard_normal(v[sl].size); v /= np.linalg.norm(v)
    for h in (1e-4, 1e-5):
        fd = (c64.eval(theta + h * v)[0] - c64.eval(theta - h * v)[0]) / (2 * h)
        fdo = (O.GPz(theta + h * v, om, X, y, Psi).nlogML - O.GPz(theta - h * v, om, X, y, Psi).nlogML) / (2 * h)
        print(f"{name} h={h:.0e}: FD(fp64 path f) {fd:+.8e}  FD(oracle f) {fdo:+.8e} | g.v oracle {ref.grad @ v:+.8e}  fp64 path {g64 @ v:+.8e}  f32 path {g32 @ v:+.8e}")
c64.close(); c32.close()
