"""Developer experiment (CPU, NumPy): what fp32 MFMA contractions would do to config 5.
SIGMA = PHI'W PHI + diag(alpha) and T = PHI*inv(SIGMA) formed (a) from fp32-rounded operands with fp32 accumulation
(v_mfma_f32_32x32x2_f32), (b) from fp32-rounded operands with fp64 accumulation, against the fp64 products; the errors are
carried through w = inv(SIGMA) PHI'W y, the objective's data term and nu.  c5's shape at reduced n, m (bench.py's theta recipe)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import gpz_oracle as O

n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 40000, int(sys.argv[2]) if len(sys.argv) > 2 else 500
cfg = dict(bench.CONFIGS["c5"]); cfg["n"] = n; cfg["m"] = m
model, theta, X, y, _ = bench.synth(cfg)
om = O.Model(m=m, d=cfg["d"], k=1, method="VC", heteroscedastic=True)
PHI, _, lnB = O.getPHI(X, None, theta, om)
P, G, lnAlpha, b, v, lnTau = O.unpack_theta(theta, om)
wb = np.exp(-lnB[:, 0]); alpha = np.exp(lnAlpha[:, 0])
S64 = (PHI * wb[:, None]).T @ PHI + np.diag(alpha)
rhs = (PHI * wb[:, None]).T @ y[:, 0]
w64 = np.linalg.solve(S64, rhs)
iS64 = np.linalg.inv(S64)
nu64 = np.einsum("ij,ij->i", PHI, PHI @ iS64)
print(f"n={n} m={m} d={cfg['d']}  cond(SIGMA) = {np.linalg.cond(S64):.3e}")
P32 = PHI.astype(np.float32)
for name, prod in [("fp32 operands, fp32 accumulate", lambda A, B: (A.astype(np.float32) @ B.astype(np.float32)).astype(np.float64)),
                   ("fp32 operands, fp64 accumulate", lambda A, B: A.astype(np.float32).astype(np.float64) @ B.astype(np.float32).astype(np.float64))]:
    S = prod((PHI * wb[:, None]).T, PHI) + np.diag(alpha)
    S = 0.5 * (S + S.T)
    w = np.linalg.solve(S, rhs)
    iS = np.linalg.inv(S)
    T = prod(PHI, iS)
    nu = np.einsum("ij,ij->i", PHI, T)
    d64 = PHI @ w64 - y[:, 0]; d = PHI @ w - y[:, 0]
    f64 = -0.5 * (wb * d64) @ d64 - 0.5 * alpha @ w64 ** 2 - 0.5 * np.linalg.slogdet(S64)[1]
    f = -0.5 * (wb * d) @ d - 0.5 * alpha @ w ** 2 - 0.5 * np.linalg.slogdet(S)[1]
    print(f"{name}: rel err SIGMA {np.abs(S - S64).max() / np.abs(S64).max():.2e}   w {np.abs(w - w64).max() / np.abs(w64).max():.2e}"
          f"   data+prior+logdet part of f {abs(f - f64) / abs(f64):.2e}   nu {np.abs(nu - nu64).max() / np.abs(nu64).max():.2e}")
