"""Developer tool: where the fixed cost of a train() goes at c2 - context creation, the eager / recording / replayed evaluations, solve, getPrior."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpz_amd, bench
from gpz_amd import api
cfg = dict(bench.CONFIGS["c2"])
model, theta, X, y, om = bench.synth(cfg)
for rep in range(2):
    t0 = time.perf_counter(); ctx = gpz_amd.GPzContext(model, X, y, None, om); t1 = time.perf_counter()
    ctx.eval(theta); t2 = time.perf_counter(); ctx.eval(theta); t3 = time.perf_counter(); ctx.eval(theta); t4 = time.perf_counter(); ctx.eval(theta); t5 = time.perf_counter()
    w, iS, _ = ctx.solve(theta); t6 = time.perf_counter()
    pr = api.getPrior(X, None, theta, model, None); t7 = time.perf_counter()
    ctx.close(); t8 = time.perf_counter()
    print("create %.1f ms | eval 1 (eager) %.2f, 2 (record) %.2f, 3 (replay) %.2f, 4 %.2f | solve %.2f | getPrior %.2f | close %.2f" % tuple(1e3 * v for v in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7)))
