"""Developer tool: GC + input noise in fp64 (one covariance for every basis function), stage times per evaluation with and without
the per-row inverse table (GPZ_GC_MINV_OFF=1).  usage: python tools/gc_psi_timing.py [n m d ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import gpz_amd
from helpers import make_problem, recondition_gamma
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 256
for d in [int(a) for a in sys.argv[3:]] or [12, 20, 32]:
    model, theta, X, Y, Psi, rng = make_problem(n, d, m, 1, "GC", True, seed=3, psi=True)
    theta = recondition_gamma(model, theta, rng)
    res = {}
    for off in (False, True):
        if off: os.environ["GPZ_GC_MINV_OFF"] = "1"
        else: os.environ.pop("GPZ_GC_MINV_OFF", None)
        ctx = gpz_amd.GPzContext(model, X, Y, Psi)
        ctx.eval(theta)
        ctx.enable_timing(True); ctx.reset_timings()
        t0 = time.perf_counter()
        for _ in range(3): f, g = ctx.eval(theta)
        wall = (time.perf_counter() - t0) / 3 * 1e3
        tm = ctx.timings()
        res[off] = (f, g)
        print(f"GC + Psi fp64  n={n} m={m} d={d}  {'per-pair sweeps' if off else 'per-row inverse table'}: {wall:7.2f} ms/eval   phi_build {tm['phi_build'][0]/3:6.2f}  moments {tm['moments'][0]/3:6.2f}", flush=True)
        ctx.close()
    gm = np.abs(res[True][1]).max()
    print(f"    f equal: {res[False][0] == res[True][0]}   max|g - g'|/max|g| = {np.abs(res[False][1] - res[True][1]).max() / gm:.2e}", flush=True)
