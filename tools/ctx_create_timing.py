import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, gpz_amd, bench
for name, n in (("c4", 1000000), ("c2", 100000), ("c5", 250000)):
    cfg = dict(bench.CONFIGS[name]); cfg["n"] = n
    model, theta, X, y, omega = bench.synth(cfg)
    psi = bench.synth_psi(cfg, np.arange(n)) if cfg.get("psi") else None
    for rep in range(2):
        t0 = time.perf_counter()
        ctx = gpz_amd.GPzContext(model, X, y, psi, omega, dtype=cfg.get("dtype", "f64"))
        t1 = time.perf_counter()
        ctx.eval(theta)
        t2 = time.perf_counter()
        ctx.close()
        print(f"{name} n={n}: create {1e3*(t1-t0):.0f} ms, first eval {1e3*(t2-t1):.0f} ms, close {1e3*(time.perf_counter()-t2):.0f} ms", flush=True)
# with missing values (GC/VC general path: pattern search per row)
cfg = dict(bench.CONFIGS["c4"]); cfg["n"] = 200000; cfg["m"] = 256
model, theta, X, y, omega = bench.synth(cfg)
Xn = X.copy(); rng = np.random.default_rng(1); Xn[rng.random(Xn.shape) < 0.03] = np.nan; Xn[:, 0] = X[:, 0]
t0 = time.perf_counter(); ctx = gpz_amd.GPzContext(model, Xn, y, None, omega); t1 = time.perf_counter(); ctx.eval(theta); ctx.close()
print(f"VC n=200000 d=10 3% NaN: create {1e3*(t1-t0):.0f} ms, patterns {len({tuple(r) for r in np.isnan(Xn)})}")
