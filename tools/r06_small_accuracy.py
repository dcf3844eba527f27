"""k_small_tail against the separate kernels: gradient error of both routes relative to the fp64 oracle, on the training fixtures' first
theta (developer build for the separate route).  usage (GPU box): python tools/r06_small_accuracy.py"""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:          # child: evaluate with the library GPZ_HIP_LIB names
    import gpz_amd
    z = np.load(sys.argv[1])
    model = gpz_amd.Model(m=int(z["m"]), d=int(z["d"]), k=1, method=str(z["method"]), heteroscedastic=bool(int(z["h"])))
    ctx = gpz_amd.GPzContext(model, z["X"], z["Y"])
    f, g = ctx.eval(z["theta"]); route = ctx.route(); ctx.close()
    np.savez(sys.argv[2], f=f, g=g, small=int("k_small_tail" in route))
    sys.exit(0)
from oracle import gpz_oracle as O
from helpers import make_problem
cases = [("VL", 3000, 1, 30, False), ("VL", 3000, 1, 100, True), ("VD", 20000, 10, 200, True), ("GL", 2000, 2, 50, True), ("VC", 4000, 5, 90, True)]
for method, n, d, m, h in cases:
    model, theta, X, Y, _, rng = make_problem(n, d, m, 1, method, h, seed=77)
    ref = O.GPz(theta, model, X, Y)
    np.savez("/tmp/acc_in.npz", X=X, Y=Y, theta=theta, m=m, d=model.d, method=model.method, h=int(h))
    out = {}
    for tag, env in (("small", {}), ("separate", {"GPZ_SMALL_TAIL_OFF": "1"})):
        e = dict(os.environ, GPZ_HIP_LIB=os.path.join(ROOT, "gpz_amd", "lib", "libgpz_hip_dev.so"), **env)
        subprocess.run([sys.executable, __file__, "/tmp/acc_in.npz", "/tmp/acc_out.npz"], check=True, env=e)
        o = np.load("/tmp/acc_out.npz")
        out[tag] = (float(o["f"]), o["g"], int(o["small"]))
    gs, gp = out["small"][1], out["separate"][1]
    sc = np.max(np.abs(ref.grad))
    print("%s n=%d d=%d m=%d hetero=%d cond %.1e | rel_g vs oracle: small %.2e (route small=%d)  separate %.2e | small vs separate %.2e | rel_f small %.1e separate %.1e"
          % (method, n, model.d, m, h, ref.cond, np.max(np.abs(gs - ref.grad)) / sc, out["small"][2], np.max(np.abs(gp - ref.grad)) / sc,
             np.max(np.abs(gs - gp)) / sc, abs(out["small"][0] - ref.nlogML) / abs(ref.nlogML), abs(out["separate"][0] - ref.nlogML) / abs(ref.nlogML)))
