"""Developer tool: ms/eval and the top stages over methods x input widths x {plain, input noise, missing values},
to spot configurations whose kernels fall off (n=100k, m=256 by default).  usage: sweep_timing.py [n] [m] [methods] [widths]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpz_amd
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(3)
for method in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["VD", "VC"]):
    for d in ([int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [2, 5, 10, 16, 20]):
        cfg = dict(n=n, d=d, m=m, method=method, omega=None)
        model, theta, X, y, _ = bench.synth(cfg)
        cov = method[1] == "C"
        diag = rng.gamma(1.0, 0.05, (n, d))
        if cov:
            Psi = np.zeros((d, d, n)); Psi[np.arange(d), np.arange(d), :] = diag.T
        else:
            Psi = diag
        Xn = X.copy()
        Xn[rng.random((n, d)) < 0.02] = np.nan
        Xn[:, 0] = X[:, 0]
        for name, kw, XX in [("plain", {}, X), ("psi", {"Psi": Psi}, X), ("nan", {}, Xn)]:
            if cov and d > 10 and name == "psi" and not os.environ.get("GPZ_SWEEP_F64"): kw = dict(kw, dtype="f32")   # GPZ_SWEEP_F64=1: the fp64 pair kernels instead
            try:
                ctx = gpz_amd.GPzContext(model, XX, y, **kw)
                for _ in range(3): ctx.eval(theta)                    # eager, recording, first replay
                t0 = time.perf_counter(); f, g = ctx.eval(theta); one = time.perf_counter() - t0
                K = max(2, min(200, int(0.1 / max(one, 1e-5))))      # ~0.1 s per case: two evaluations of a 0.4 ms shape are timer noise
                for _ in range(K): ctx.eval(theta)                    # warm-up of the same length: the first case after the host-side data
                t0 = time.perf_counter()                              # generation otherwise runs at the clock of an idle device
                for _ in range(K): f, g = ctx.eval(theta)
                dt = (time.perf_counter() - t0) / K                   # the replayed evaluation: what a caller gets
                ctx.enable_timing(True); ctx.eval(theta); ctx.reset_timings()   # stage split from a separate pass (eager launches with events)
                for _ in range(K): ctx.eval(theta)
                tim = ctx.timings()
                print("%s d=%-2d %-5s %8.2f ms  " % (method, d, name, dt * 1e3) +
                      " ".join("%s=%.2f" % (k, v[0] / K) for k, v in sorted(tim.items(), key=lambda x: -x[1][0])[:int(os.environ.get("GPZ_SWEEP_TOP", "4"))]), flush=True)
                ctx.close()
            except Exception as e:
                print(method, d, name, "ERR", str(e)[:100], flush=True)
