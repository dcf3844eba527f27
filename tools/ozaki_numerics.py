"""CPU emulation of an int8-sliced (Ozaki scheme) evaluation of T = PHI * [inv(SIGMA) | w] at c4's shape (tools only; not product).

PHI in (0, 1] is sliced as FIXED-POINT balanced base-256 digits (int8, [-128, 127]) of rint(PHI * 2^54) — no per-row scaling, so the
same digits could serve any contraction direction; B = inv(SIGMA) gets a per-column power-of-two scale.  Products of digit planes are
exact in int32 (emulated here with fp64 matmuls of small integers, exact below 2^53).  Levels s + t <= S + 1 are kept.
Compares against numpy's fp64 matmul and an 80-bit long-double reference on a row subset."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import gpz_oracle as O


def digits(N, S):
    """balanced base-256 digits of int64 N, most significant first; exact: N = sum d_s 256^(S-s)"""
    out = []
    N = N.copy()
    for _ in range(S):
        d = ((N + 128) % 256) - 128
        out.append(d.astype(np.float64))
        N = (N - d) // 256
    assert np.all(N == 0), "top digit overflow"
    return out[::-1]


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    cfg = dict(bench.CONFIGS["c4"]); cfg["n"] = rows
    model, theta, X, y, omega = bench.synth(cfg)
    Om = O.Model(m=model.m, d=model.d, k=1, method=model.method, heteroscedastic=True)
    PHI, _, lnb = O.getPHI(X, None, theta, Om, None)[:3]
    beta = np.exp(-lnb[:, 0])
    m = model.m
    lna = theta[m * model.d + model.g_dim: m * model.d + model.g_dim + m]
    # scale the row count up: SIGMA of the full problem is ~ n/rows times this one's data term
    scale = 1_000_000 / rows
    Sig = scale * (PHI * beta[:, None]).T @ PHI + np.diag(np.exp(lna))
    iS = np.linalg.inv(Sig)
    print("cond(Sigma) = %.2e" % np.linalg.cond(Sig))
    sub = slice(0, 256)
    A = PHI[sub]
    ref = (A.astype(np.longdouble) @ iS.astype(np.longdouble))
    T64 = A @ iS
    scl = np.max(np.abs(ref))
    print("fp64 matmul : max|dT|/max|T| = %.2e" % float(np.max(np.abs(T64 - ref)) / scl))
    for S in (6, 7, 8):
        bits = 8 * S - 2
        NA = np.rint(A * 2.0 ** bits).astype(np.int64)
        dA = digits(NA, S)
        f = np.ceil(np.log2(np.max(np.abs(iS), axis=0))) + 2            # |B| / 2^f <= 1/4
        NB = np.rint(iS / 2.0 ** f[None, :] * 2.0 ** (8 * S)).astype(np.int64)
        dB = digits(NB, S)
        T = np.zeros_like(T64)
        nprod = 0
        for lvl in range(2, S + 2):                                      # s + t = lvl, 1-based
            acc = np.zeros_like(T64)
            for s in range(1, lvl):
                t = lvl - s
                if s <= S and t <= S:
                    acc += dA[s - 1] @ dB[t - 1]
                    nprod += 1
            T += acc * 2.0 ** (-8 * lvl)
        T *= 2.0 ** (8 * S - bits) * 2.0 ** f[None, :]
        err = float(np.max(np.abs(T - ref)) / scl)
        nu = np.sum(A * T, axis=1); nur = np.sum(A.astype(np.longdouble) * ref, axis=1)
        print("ozaki S=%d (%2d products): max|dT|/max|T| = %.2e   nu rel err max = %.2e (fp64: %.2e)"
              % (S, nprod, err, float(np.max(np.abs(nu - nur) / np.abs(nur))),
                 float(np.max(np.abs(np.sum(A * T64, axis=1) - nur) / np.abs(nur)))))


if __name__ == "__main__":
    main()
