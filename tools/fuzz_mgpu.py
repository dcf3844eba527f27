"""Developer tool: randomised sweep of the NATIVE multi-GPU driver (gpz_mgpu_*, all shards on this box's one GPU with the
in-library loopback reducer) against the oracle on the unsharded data: methods, input noise, missing values, weights, masks,
outputs, fp32 pair path, 2-6 shards; eval, solve and the statistics.  usage: fuzz_mgpu.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import gpz_amd
from gpz_amd import _lib
from oracle import gpz_oracle as O
from helpers import grad_tol, rel
from fuzz_sharded import draw, build

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = refused = 0
t0 = time.time()
for c in range(cases):
    cfg = draw(rng)
    shards = int(rng.integers(2, 7))
    model, theta, X, Y, Psi, om, tr, va = build(cfg)
    if os.environ.get("FUZZ_VERBOSE"): print("case", c, cfg, "shards", shards, flush=True)
    ntr = int(tr.sum()) if tr is not None else X.shape[0]
    try:
        mg = gpz_amd.GPzMulti(model, X, Y, Psi, om, tr, va, n_gpus=shards, reducer="loopback", dtype="f32" if cfg["f32"] else "f64")
    except _lib.GpzError as e:
        if ntr < shards and "cannot be split" in str(e):
            refused += 1
            continue
        bad += 1; print("ERROR create", cfg, shards, e); continue
    try:
        f, g = mg.eval(theta); st = dict(mg.stats)
        f2, g2 = mg.eval(theta)
        w, iS, part = mg.solve(theta)
    except Exception as e:
        bad += 1; print("ERROR eval", cfg, shards, repr(e)[:300]); mg.close(); continue
    mg.close()
    ref = O.GPz(theta, model, X, Y, Psi, om, tr, va)
    r4 = O.GPz(theta, model, X, Y, Psi, om, tr, va, nargout=4)
    f32 = cfg["f32"] and cfg["nanfrac"] == 0.0
    tol_f, tol_g = (1e-4, 1e-3) if f32 else (1e-8, grad_tol(ref.cond))
    if model.method[1] == "C" and not f32:
        P, G, *_ = O.unpack_theta(theta, model); Gm = O.expand_gamma(G, model)
        cg = max(np.linalg.cond(Gm[:, :, j].T @ Gm[:, :, j]) for j in range(Gm.shape[2]))
        tol_g = max(tol_g, 50 * cg * 2.2e-16)
        if (cfg["psi"] or cfg["nanfrac"] > 0) and cg > 1e4: tol_g = max(tol_g, 1e-2)
    es = max((0.0 if (np.isnan(v) and np.isnan(st.get(kk, np.nan))) else abs(st.get(kk, np.nan) - v) / max(1.0, abs(v)))
             for kk, v in ref.stats.items())
    ok = (abs(f - ref.nlogML) <= max(tol_f, tol_g) * abs(ref.nlogML) and rel(g, ref.grad) <= tol_g and es <= max(1e-9, tol_f)
          and f2 == f and np.array_equal(g2, g)
          and rel(w, r4.w) <= max(tol_g, tol_f) and rel(part, r4.nlogML) <= max(tol_f, 1e-8))
    if not ok:
        bad += 1
        print("FAIL", cfg, "shards", shards, f"ef={abs(f - ref.nlogML) / abs(ref.nlogML):.1e} eg={rel(g, ref.grad):.1e} es={es:.1e} "
              f"ew={rel(w, r4.w):.1e} repeat={f2 == f and np.array_equal(g2, g)} tol_g={tol_g:.1e}")
print(f"{cases} multi-GPU (loopback) cases, {bad} failures, {refused} refused (rows < shards), {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
