"""A 25-band "photometric catalogue": the flow of the reference's demo_photoz.m (init -> train with a validation split,
cost-sensitive weights -> predict with per-band flux errors and missing bands), at an input width the instantiated kernels
do not cover (d = 25 > 20: the runtime-d route of k_wide.hip).  Synthetic data; needs an MI355X.

    python examples/demo_photoz_wide.py [--n 20000] [--d 25] [--m 60] [--method VD]

demo_photoz.m:22-24 (method, m), :33-45 (split, omega = getOmega(Y, 'normalized')), :56-61 (init / train), :66-75 (predict, metrics).
"""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpz_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--d", type=int, default=25)
    ap.add_argument("--m", type=int, default=60)
    ap.add_argument("--method", default="VD")
    ap.add_argument("--maxIter", type=int, default=150)
    ap.add_argument("--maxAttempts", type=int, default=30)
    args = ap.parse_args()

    rng = np.random.default_rng(7)
    n, d = args.n, args.d
    z = rng.gamma(2.0, 0.35, n)                                        # "redshift"
    knots = np.linspace(0.0, 3.0, d)
    mags = 22.0 - 2.0 * np.exp(-0.5 * ((z[:, None] - knots[None, :]) / 0.7) ** 2) + 0.3 * rng.standard_normal((n, 1))
    err = 0.02 + 0.08 * rng.random((n, d))                              # per-band magnitude errors
    X = mags + err * rng.standard_normal((n, d))
    Psi = err ** 2                                                      # n x d variances: fixPsi.m:42-53 (diag kinds), :27-31 (GC/VC)
    Y = z[:, None]

    tr, va, te = gpz_amd.sample(n, 0.6, 0.2, 0.2, rng)
    omega = gpz_amd.getOmega(Y, "normalized")                           # demo_photoz.m:45: weights (1 + z)^-2
    model = gpz_amd.init(X, Y, args.method, args.m, heteroscedastic=True, omega=omega, training=tr, Psi=Psi, rng=rng)
    model = gpz_amd.train(model, X, Y, maxIter=args.maxIter, maxAttempts=args.maxAttempts, omega=omega, training=tr,
                          validation=va, Psi=Psi)

    def report(name, mu, sigma):
        e = (mu[:, 0] - Y[te, 0]) / (1.0 + Y[te, 0])
        print(f"{name}: RMSE(dz/(1+z)) = {math.sqrt(np.mean(e ** 2)):.4f}   bias = {np.mean(e):+.4f}   "
              f"FR15 = {100.0 * np.mean(np.abs(e) < 0.15):.1f} %   mean predictive sd = {np.mean(np.sqrt(sigma[:, 0])):.4f}")

    print(f"{int(te.sum())} test objects, {d} bands, m = {args.m}, method = {model.method}")
    mu, sigma = gpz_amd.predict(X, model, Psi=Psi, selection=te)[:2]
    report("all bands, with flux errors ", mu, sigma)
    Xm = X.copy()
    drop = rng.random((n, d)) < 0.05                                    # 5 % of the band measurements missing
    drop[:, 0] = False
    Xm[drop] = np.nan
    mu, sigma = gpz_amd.predict(Xm, model, Psi=Psi, selection=te)[:2]   # predict.m:45-69: one call per NaN pattern
    report("5 % of the bands missing    ", mu, sigma)


if __name__ == "__main__":
    main()
