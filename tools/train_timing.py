"""Developer tool: wall time per objective evaluation of a whole train() run (host.py: minFunc's L-BFGS around gpz_eval) against the bare
gpz_eval rate of the same problem - what the optimiser, the callback and the host round trips add.  usage: train_timing.py [config] [iters]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpz_amd
from gpz_amd import host
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cfg = dict(bench.CONFIGS[name])
model, theta, X, y, om = bench.synth(cfg)
ctx = gpz_amd.GPzContext(model, X, y, None, om)
for _ in range(3): ctx.eval(theta)
t0 = time.perf_counter()
for _ in range(50): ctx.eval(theta)
ev = (time.perf_counter() - t0) / 50
ctx.close()
print("%s: bare gpz_eval %.3f ms" % (name, ev * 1e3))
for resident in (False, True):
    res = []
    for it in (iters, 4 * iters):
        model.sets = {"last": {"theta": theta.copy()}, "best": {"theta": theta.copy(), "LL": -np.inf}}
        t0 = time.perf_counter()
        mdl = host.train(model, X, y, maxIter=it, omega=om, verbose=False, device_resident=resident)
        res.append((time.perf_counter() - t0, mdl.train_info["funEvals"]))
    (t1, e1), (t2, e2) = res
    per = (t2 - t1) / max(1, e2 - e1)
    print("  train(device_resident=%s): %d evaluations in %.1f ms, %d in %.1f ms -> %.3f ms per further evaluation (%.2f x the bare one), "
          "%.1f ms fixed (context, upload, graph recording, the two solves + getPrior at the end)"
          % (resident, e1, t1 * 1e3, e2, t2 * 1e3, per * 1e3, per / ev, (t1 - per * e1) * 1e3))
