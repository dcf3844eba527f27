"""1-D heteroscedastic sinc regression through the HIP path: the flow of the reference's demo_sinc.m
(init -> train with a validation split and early stopping -> predict -> RMSE / mean log-likelihood),
without the plotting.  Needs an MI355X.

    python examples/demo_sinc.py [--n 10000] [--m 100] [--method VL]
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
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--m", type=int, default=100)          # demo_sinc.m:7
    ap.add_argument("--method", default="VL")              # demo_sinc.m:9
    ap.add_argument("--maxIter", type=int, default=500)
    ap.add_argument("--maxAttempts", type=int, default=50)
    ap.add_argument("--device-resident", action="store_true", help="keep theta, g and the L-BFGS memory on the GPU (also the default where device vectors are available)")
    ap.add_argument("--host-vectors", action="store_true", help="optimiser vectors on the host: minFunc's own arithmetic order")
    args = ap.parse_args()

    rng = np.random.default_rng(1)                         # demo_sinc.m:1  rng(1)
    X = np.linspace(-10, 10, args.n)[:, None]
    keep = (X[:, 0] < -7) | (X[:, 0] > -3)                 # a gap in the inputs, as in the demo
    X = X[keep]
    n = X.shape[0]
    fx = np.sinc(X[:, 0] / math.pi)
    noise_sd = 0.05 + 0.2 * (1 + np.sin(2 * X[:, 0] / 3)) / 2
    Y = (fx + noise_sd * rng.standard_normal(n))[:, None]
    Psi = rng.gamma(1.0, 0.5, n) * 1e-2                    # input-noise variances, Gamma(a=1, b=0.5) shaped (demo_sinc.m:39-45)
    Xn = X + np.sqrt(Psi)[:, None] * rng.standard_normal((n, 1))

    tr, va, te = gpz_amd.sample(n, 0.70, 0.15, 0.15, rng)  # demo_sinc.m:30-32
    model = gpz_amd.init(Xn, Y, args.method, args.m, heteroscedastic=True, training=tr, Psi=Psi, rng=rng)
    model = gpz_amd.train(model, Xn, Y, maxIter=args.maxIter, maxAttempts=args.maxAttempts, training=tr, validation=va,
                          Psi=Psi, device_resident=True if args.device_resident else (False if args.host_vectors else None))

    def report(name, mu, sigma):
        err = mu[:, 0] - Y[te, 0]
        rmse = math.sqrt(np.mean(err ** 2))
        mll = np.mean(-0.5 * err ** 2 / sigma[:, 0] - 0.5 * np.log(sigma[:, 0])) - 0.5 * math.log(2 * math.pi)
        print(f"{name}: test RMSE = {rmse:.5f}   test MLL = {mll:.5f}")

    print(f"{te.sum()} test points, m = {args.m}, method = {model.method}")
    mu, sigma = gpz_amd.predict(X, model, selection=te)[:2]                  # noise-free inputs      (predictFull)
    report("clean inputs      ", mu, sigma)
    mu, sigma = gpz_amd.predict(Xn, model, Psi=Psi, selection=te)[:2]        # noisy inputs, known Psi (predictNoisy, demo_sinc.m:101)
    report("noisy inputs + Psi", mu, sigma)


if __name__ == "__main__":
    main()
