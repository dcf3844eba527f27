"""Developer tool: wall time per L-BFGS iteration (minFunc driver of gpz_amd/host.py) on a bench workload, with the
optimiser vectors on the host and on the device.  usage: train_timing.py [config] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpz_amd, bench
from gpz_amd import host
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = dict(bench.CONFIGS[cfgname])
model, theta, X, y, omega = bench.synth(cfg)
ctx = gpz_amd.GPzContext(model, X, y, None, omega)
ctx.eval(theta)
for name, x0, fun in (("host vectors  ", theta, ctx.eval),
                      ("device vectors", host.DevVec.from_host(theta), lambda t: (lambda r: (r[0], host.DevVec(r[1])))(ctx.eval_dev(t.t)))):
    ev = []
    t0 = time.perf_counter()
    x, f, flag, evals, msg = host.minfunc_lbfgs(fun, x0, max_iter=iters, corrections=100)
    dt = time.perf_counter() - t0
    print(f"{cfgname} {name}: {iters} iterations, {evals} evaluations, {dt:.2f} s = {1e3 * dt / evals:.1f} ms per evaluation (f = {f:.9f}; {msg})")
ctx.close()
