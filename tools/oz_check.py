"""Developer tool: the int8-sliced T-GEMM route (GPZ_TGEMM_INT8=1, k_oz.hip; developer build of the library: run with
GPZ_HIP_LIB=gpz_amd/lib/libgpz_hip_dev.so) against the fp64 MFMA route and the oracle.
usage: GPZ_HIP_LIB=... python tools/oz_check.py [n m d method]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpz_amd
from oracle import gpz_oracle as O
from helpers import make_problem, rel
n, m, d = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 6000), (2, 300), (3, 6)))
method = sys.argv[4] if len(sys.argv) > 4 else "VC"
model, theta, X, Y, _, rng = make_problem(n, d, m, 1, method, True, seed=11)
out = {}
for tag, env in (("fp64 MFMA", None), ("int8 x 28", "1")):
    if env: os.environ["GPZ_TGEMM_INT8"] = env
    else: os.environ.pop("GPZ_TGEMM_INT8", None)
    ctx = gpz_amd.GPzContext(model, X, Y)
    f, g = ctx.eval(theta)
    for _ in range(3): f2, g2 = ctx.eval(theta)
    assert f2 == f and np.array_equal(g2, g), "replay differs"
    t0 = time.perf_counter()
    for _ in range(5): ctx.eval(theta)
    dt = (time.perf_counter() - t0) / 5
    out[tag] = (f, g, dict(ctx.stats))
    print(f"{tag}: f = {f!r}  {dt * 1e3:.3f} ms/eval  {ctx.route()}")
    ctx.close()
ref = O.GPz(theta, model, X, Y)
for tag, (f, g, st) in out.items():
    print(f"{tag}: rel_f {abs(f - ref.nlogML) / abs(ref.nlogML):.2e}  rel_g {rel(g, ref.grad):.2e}  (cond {ref.cond:.1e}, tol {max(1e-8, 50 * ref.cond * 2.2e-16):.1e})  "
          f"stats {max(abs(st[k] - v) for k, v in ref.stats.items() if np.isfinite(v)):.1e}")
fa, ga, _ = out["fp64 MFMA"]; fb, gb, _ = out["int8 x 28"]
print(f"int8 vs fp64 route: rel_f {abs(fa - fb) / abs(fa):.2e}  rel_g {rel(gb, ga):.2e}")
