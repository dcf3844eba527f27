"""Developer check of the native multi-GPU driver (gpz_mgpu_*): loopback shards on one GPU against the plain context."""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import gpz_amd
from helpers import make_problem
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
n = 3001
psi = len(sys.argv) > 3 and sys.argv[3] == "psi"
model, theta, X, Y, Psi, rng = make_problem(n, 6, 40, 1, "VC", True, seed=33, psi=psi)
r2 = np.random.default_rng(1)
om = r2.random((n, 1)) + 0.5
tr = r2.random(n) < 0.8
args = {"plain": (None, None, None), "omega": (om, None, None), "train": (om, tr, None), "valid": (om, tr, ~tr)}[mode]
print("create", mode, flush=True)
mg = gpz_amd.GPzMulti(model, X, Y, Psi, *args, n_gpus=shards, reducer="loopback")
print("rows", mg.rows_per_gpu, flush=True)
f, g = mg.eval(theta)
print("eval", f, mg.stats, flush=True)
f2, g2 = mg.eval(theta)
print("eval2", f2 == f, np.array_equal(g, g2), flush=True)
w, iS, part = mg.solve(theta)
print("solve", part, flush=True)
one = gpz_amd.GPzContext(model, X, Y, Psi, *args)
f1, g1 = one.eval(theta)
print("plain", f1, abs(f - f1), np.max(np.abs(g - g1)), one.stats, flush=True)
mg.close(); one.close()
