"""Developer tool: wall time of the single operations a device-resident train() iteration is made of (c2): the evaluation, x + t d, g'd with its
read-back, copies, the L-BFGS memory - what tools/train_timing.py measures as a whole."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpz_amd, bench
from gpz_amd import host
cfg = dict(bench.CONFIGS["c2"])
model, theta, X, y, om = bench.synth(cfg)
ctx = gpz_amd.GPzContext(model, X, y, None, om)
x = host.DevVec.from_host(theta); 
f, g = ctx.eval_dev(x.t); g = host.DevVec(g)
d = -g
def T(name, fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print("%-28s %.1f us" % (name, (time.perf_counter() - t0) / n * 1e6))
T("eval_dev", lambda: ctx.eval_dev(x.t))
T("x + t*d (axpy)", lambda: x + 0.5 * d)
def dot():
    gg = host.DevVec(g.t); return gg @ d
T("g @ d (stats + read-back)", dot)
T("g.copy()", lambda: g.copy())
T("-g", lambda: -g)
T("ctx.stats", lambda: ctx.stats)
mem = host._LBFGSDevice(x.size, 100)
g2 = host.DevVec(g.t * 1.01)
def lb():
    mem.add_step(g2, g, 0.5, d); return mem.direction(g2)
T("lbfgs add + direction", lb, 100)
