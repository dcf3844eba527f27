"""Developer tool: wall time and host-side profile of predict() without missing values at n_s = 1e5, m = 500 (most of it is the
n_s x m PHI output crossing PCIe).  Run on the GPU box: python tools/predict_full_timing.py"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gpz_amd
from gpz_amd import _lib, api
from helpers import make_problem
from oracle import gpz_oracle as O
n, d, m = 100000, 10, 500
model, theta, X, Y, _, rng = make_problem(4000, d, m, 1, "VC", True, seed=1)
r4 = O.GPz(theta, model, X, Y, nargout=4)
model.sets["best"] = {"theta": theta, "w": r4.w, "iSigma_w": r4.iSigma_w, "priors": np.full(m, 1.0 / m)}
model.muX = np.zeros(d); model.sdX = np.ones(d); model.muY = np.zeros(1)
Xs = rng.standard_normal((n, d))
for rep in range(3):
    t0 = time.perf_counter(); out = gpz_amd.predict(Xs, model); t = time.perf_counter() - t0
    print("predict", t)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); gpz_amd.predict(Xs, model); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
