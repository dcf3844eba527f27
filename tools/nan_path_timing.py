"""Developer tool: timing of GC/VC and VD with missing values (no input noise) on a mid-size problem."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpz_amd, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 10
m = int(sys.argv[3]) if len(sys.argv) > 3 else 200
for method in ("VC", "VD"):
    cfg = dict(n=n, d=d, m=m, method=method, omega=None)
    model, theta, X, y, _ = bench.synth(cfg)
    rng = np.random.default_rng(5)
    Xn = X.copy()
    pat = rng.integers(0, 6, n)            # 6 NaN patterns: none, {0}, {1}, {0,1}, {d-1}, {2,3}
    for p, cols in enumerate(([], [0], [1], [0, 1], [d - 1], [2, 3])):
        for c in cols:
            Xn[pat == p, c] = np.nan
    for name, XX in (("complete", X), ("missing ", Xn)):
        ctx = gpz_amd.GPzContext(model, XX, y)
        ctx.eval(theta)
        ctx.enable_timing(True); ctx.reset_timings()
        t0 = time.perf_counter(); K = 3
        for _ in range(K):
            f, g = ctx.eval(theta)
        dt = (time.perf_counter() - t0) / K
        tim = ctx.timings()
        print(method, name, "ms/eval %.2f" % (dt * 1e3), "f=%.6f" % f, " ".join("%s=%.2f" % (k, v[0] / K) for k, v in sorted(tim.items(), key=lambda x: -x[1][0])[:5]))
        ctx.close()
