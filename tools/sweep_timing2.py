"""Developer tool: ms/eval over all six methods x outputs x {plain, input noise, missing values, both}, with a validation
mask and weights, through tests/helpers.make_problem (n=50k, m=128).  usage: sweep_timing2.py [n] [m] [d]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpz_amd
from helpers import make_problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d = int(sys.argv[3]) if len(sys.argv) > 3 else 8
for method in ["GL", "VL", "GD", "VD", "GC", "VC"]:
    for k in [1, 2]:
        for name, psi, nanfrac in [("plain", False, 0.0), ("psi", True, 0.0), ("nan", False, 0.05), ("psi+nan", True, 0.05)]:
            model, theta, X, Y, Psi, rng = make_problem(n, d, m, k, method, True, seed=5, psi=psi, nanfrac=nanfrac)
            tr = rng.random(n) < 0.8
            try:
                ctx = gpz_amd.GPzContext(model, X, Y, Psi, None, tr, ~tr)
                ctx.eval(theta)
                ctx.enable_timing(True); ctx.reset_timings()
                t0 = time.perf_counter(); K = 2
                for _ in range(K): f, g = ctx.eval(theta)
                dt = (time.perf_counter() - t0) / K
                tim = ctx.timings()
                print("%s k=%d %-7s %8.2f ms  " % (method, k, name, dt * 1e3) +
                      " ".join("%s=%.2f" % (kk, v[0] / K) for kk, v in sorted(tim.items(), key=lambda x: -x[1][0])[:4]), flush=True)
                ctx.close()
            except Exception as e:
                print(method, k, name, "ERR", str(e)[:100], flush=True)
