"""Developer tool: fp64 GC/VC + Psi beyond the register kernels (k_cpsi4.hip / k_cpsi4w.hip / k_cpsi.hip, 10 < d <= 64) against the oracle,
+- missing values; CPSI_D=11,20,... picks the widths, GPZ_CPSI4_OFF / GPZ_CPSI_OFF the route.  usage: cpsi_check.py [n] [m]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import gpz_amd
from oracle import gpz_oracle as O
from helpers import make_problem, rel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
m = int(sys.argv[2]) if len(sys.argv) > 2 else 7
bad = 0
for method in ("VC", "GC"):
    for d in ([int(x) for x in os.environ['CPSI_D'].split(',')] if 'CPSI_D' in os.environ else (11, 16, 17, 24, 32, 33, 48, 50, 64)):
        for nanfrac in (0.0, 0.3):
            model, theta, X, Y, Psi, rng = make_problem(n, d, m, 1, method, True, seed=100 + d, psi=True, nanfrac=nanfrac)
            ref = O.GPz(theta, model, X, Y, Psi)
            ctx = gpz_amd.GPzContext(model, X, Y, Psi)
            try:
                f, g = ctx.eval(theta)
                t0 = time.perf_counter()
                f, g = ctx.eval(theta)
                dt = time.perf_counter() - t0
            finally:
                ctx.close()
            ef, eg = abs(f - ref.nlogML) / abs(ref.nlogML), rel(g, ref.grad)
            P_, G_, *_ = O.unpack_theta(theta, model)
            Gam = O.expand_gamma(G_, model)
            cg = max(np.linalg.cond(Gam[:, :, q].T @ Gam[:, :, q]) for q in range(Gam.shape[2]))
            # the gates of tests/test_reference_run.py: the reference's dGamma chain through inv(Gamma'Gamma) loses cond^1.5 eps
            ok = ef < max(1e-8, 200 * cg * 2.2e-16) and eg < max(1e-8, 50 * ref.cond * 2.2e-16, 200 * cg * 2.2e-16, 10 * cg ** 1.5 * 2.2e-16)
            bad += not ok
            print(f"{method} d={d:2d} nan={nanfrac}: f {ef:.1e} grad {eg:.1e} cond {ref.cond:.1e} {dt*1e3:.1f} ms {'ok' if ok else 'FAIL'}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
