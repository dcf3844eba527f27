"""Developer tool: per-stage GPU times of bench.py's workload.  usage: stage_times.py [config] [n]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, gpz_amd, bench
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c4"
cfg = dict(bench.CONFIGS[cfgname])
if len(sys.argv) > 2: cfg["n"] = int(sys.argv[2])
model, theta, X, y, omega = bench.synth(cfg)
ctx = gpz_amd.GPzContext(model, X, y, None, omega)
for _ in range(2): ctx.eval(theta)
ctx.enable_timing(True); ctx.reset_timings()
K = 5
import time; t0 = time.perf_counter()
for _ in range(K): f, g = ctx.eval(theta)
dt = (time.perf_counter() - t0) / K
tim = ctx.timings()
print(os.environ.get("GPZ_HIP_LIB", "default"), cfgname, cfg["n"], "ms/eval %.3f" % (dt * 1e3), "f=%.10f" % f,
      " ".join("%s=%.3f" % (k, v[0] / K) for k, v in sorted(tim.items(), key=lambda x: -x[1][0])[:8]))
