"""Developer experiment (GPU): config 5's shard evaluated twice - as shipped, and with PHI rounded to fp32 before the two MFMA
contractions (what fp32-operand MFMAs with ideal fp64 accumulation would see; GPZ_EXPERIMENT_ROUND_PHI32).  The difference in f
and g is a LOWER bound on the error fp32 MFMA contractions would add at this size.  usage: f32_operand_experiment.py [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import gpz_amd

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
cfg = dict(bench.CONFIGS["c5"]); cfg["n"] = rows
model, theta, X, y, _ = bench.synth(cfg)
Psi = bench.synth_psi(cfg, np.arange(rows), cube=True)
out = {}
for tag, env in (("fp64 PHI", None), ("PHI rounded to fp32", "1")):
    if env: os.environ["GPZ_EXPERIMENT_ROUND_PHI32"] = env
    else: os.environ.pop("GPZ_EXPERIMENT_ROUND_PHI32", None)
    ctx = gpz_amd.GPzContext(model, X, y, Psi, dtype="f32")
    out[tag] = ctx.eval(theta)
    ctx.close()
(f0, g0), (f1, g1) = out["fp64 PHI"], out["PHI rounded to fp32"]
m, d = cfg["m"], cfg["d"]
blocks = {"dP": slice(0, m * d), "dGamma": slice(m * d, m * d + d * d * m), "dlnAlpha": slice(m * d + d * d * m, m * d + d * d * m + m),
          "db,dv,dlnTau": slice(m * d + d * d * m + m, theta.size)}
print(f"rows={rows} m={m} d={d}: f {f0:.12g} vs {f1:.12g}  rel {abs(f1 - f0) / abs(f0):.2e}")
print(f"g: max|dg|/max|g| = {np.abs(g1 - g0).max() / np.abs(g0).max():.2e}")
for k, sl in blocks.items():
    print(f"   {k:14s} max|dg| / max|g_block| = {np.abs(g1[sl] - g0[sl]).max() / np.abs(g0[sl]).max():.2e}")
