#!/usr/bin/env python3
"""Developer tool: cost line of prediction with missing values beyond the tuned widths (GC/VC d > 64: k_pmiss_covg.hip, diagonal kinds
d > 144: the LDS-free instantiations of k_pmiss.hip).  Run on the GPU box: python tools/pm_wide_timing.py"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gpz_amd
from helpers import make_problem, recondition_gamma
from oracle import gpz_oracle as O

for method, d, m, ns in [("VC", 64, 4, 9), ("GC", 66, 4, 9), ("VC", 100, 4, 9), ("VC", 100, 16, 9), ("VD", 144, 8, 30), ("VD", 150, 8, 30), ("VD", 260, 8, 30)]:
    model, theta, X, Y, _, rng = make_problem(120, d, m, 1, method, True, seed=1)
    if method in ("GC", "VC"):
        theta = recondition_gamma(model, theta, rng)
    elif d > 200:
        theta = theta.copy(); theta[m * d:2 * m * d] = 0.3
    r4 = O.GPz(theta, model, X, Y, nargout=4)
    model.sets["best"] = {"theta": theta, "w": r4.w, "iSigma_w": r4.iSigma_w, "priors": np.full(m, 1.0 / m)}
    Xs = rng.standard_normal((ns, d)); Xs[:, [1, d - 1]] = np.nan
    gpz_amd.predict(Xs, model)
    t0 = time.perf_counter(); gpz_amd.predict(Xs, model); t = time.perf_counter() - t0
    print(f"{method} d={d} m={m} rows={ns} (one NaN pattern): {t * 1e3:.1f} ms")
