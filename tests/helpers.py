"""Shared helpers for the test-suite."""
import glob
import os

import numpy as np

from oracle import gpz_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(prefix="g_"):
    """g_: GPz / getPHI / predictFull cases (make_golden.py); p_: predict with every branch; s_: truncating inv_logdet
    (make_golden_predict.py)."""
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_predict_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: z[k] for k in z.files}
    model = O.Model(m=int(g["m"]), d=int(g["d"]), k=int(g["k"]), method=str(g["method"]), heteroscedastic=True)
    model.muX, model.sdX, model.muY = g["muX"], g["sdX"], g["muY"]
    model.sets["best"] = {"theta": g["theta"], "w": g["w"], "iSigma_w": g["iSigma_w"], "priors": g["priors"]}
    return g, model, (g["Psi"] if int(g["has_psi"]) else None)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: z[k] for k in z.files}
    model = O.Model(m=int(g["m"]), d=int(g["d"]), k=int(g["k"]), method=str(g["method"]),
                    heteroscedastic=bool(int(g["heteroscedastic"])))
    Psi = g["Psi"] if int(g["has_psi"]) else None
    if int(g["has_masks"]):
        omega, training, validation = g["omega"], g["training"], g["validation"]
    else:
        omega = training = validation = None
    return g, model, Psi, omega, training, validation


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def make_problem(n, d, m, k, method, hetero, seed, psi=False, nanfrac=0.0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    A = rng.standard_normal((d, k)) / np.sqrt(d)
    Y = np.sin(X @ A) + 0.1 * rng.standard_normal((n, k))
    Y -= Y.mean(0)
    model, theta = O.init_theta(X, Y, method, m, hetero, rng)
    theta = theta + 0.05 * rng.standard_normal(theta.size)
    if hetero:
        o = theta.size - 2 * m * k
        theta[o:o + m * k] = 0.05 * rng.standard_normal(m * k)
    Psi = None
    if psi:
        if model.method[1] == "C":
            Psi = np.zeros((d, d, n))
            for i in range(n):
                B = 0.3 * rng.standard_normal((d, d))
                Psi[:, :, i] = B @ B.T
        else:
            Psi = rng.gamma(1.0, 0.2, (n, d))
    if nanfrac > 0 and d > 1:
        rows = rng.random(n) < nanfrac
        X[rows, rng.integers(0, d, n)[rows]] = np.nan
    return model, theta, X, Y, Psi, rng


def recondition_gamma(model, theta, rng, amp=0.3):
    """Redraw the Gamma blocks of a GC/VC theta as gamma_j (I + amp G / sqrt(d)).  make_problem's 0.05 N(0,1) perturbation of
    gamma_j I is larger than gamma_j itself once d ~ 20 (cond(Gamma'Gamma) ~ 1e8 ... 1e9): with input noise the reference's own
    dGamma chain loses cond^1.5 eps there and both sides of a comparison are rounding noise."""
    if model.method[1] != "C":
        return theta
    m, d = model.m, model.d
    md = m * d
    for q in range(1 if model.method == "GC" else m):
        blk = theta[md + q * d * d: md + (q + 1) * d * d].reshape((d, d), order="F")
        gam = float(np.mean(np.diag(blk)))
        theta[md + q * d * d: md + (q + 1) * d * d] = (gam * (np.eye(d) + amp * rng.standard_normal((d, d)) / np.sqrt(d))).reshape(-1, order="F")
    return theta


def grad_tol(cond):
    """Parity gate of BASELINE.md §6."""
    return max(1e-8, 50.0 * cond * 2.2e-16)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV_LIB = os.path.join(ROOT, "gpz_amd", "lib", "libgpz_hip_dev.so")


def eval_with_dev_switches(tmp_path, method, m, d, k, hetero, theta, X, Y, Psi, switches):
    """One evaluation in a FRESH process on the developer build of the library (./build.sh --dev: -DGPZ_DEV_SWITCHES, the only build
    in which the A/B switches of gpz_amd/csrc/gpz_options.h exist), with `switches` in its environment.  -> (f, g, info)"""
    import subprocess
    import sys
    if not os.path.exists(DEV_LIB):      # (normally built by __graft_entry__.build() and shipped in-tree; a bare checkout builds it here)
        subprocess.run(["bash", os.path.join(ROOT, "build.sh"), "--dev"], cwd=ROOT, check=True, capture_output=True, timeout=1800)
    assert os.path.exists(DEV_LIB), "developer build missing: ./build.sh --dev failed"
    np.savez(tmp_path / "in.npz", theta=theta, X=X, Y=Y, Psi=(Psi if Psi is not None else np.zeros(0)), has_psi=int(Psi is not None))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import gpz_amd\n"
            "z = np.load(%r)\n"
            "model = gpz_amd.Model(m=%d, d=%d, k=%d, method=%r, heteroscedastic=%r)\n"
            "ctx = gpz_amd.GPzContext(model, z['X'], z['Y'], z['Psi'] if int(z['has_psi']) else None)\n"
            "f, g = ctx.eval(z['theta']); info = ctx.info; ctx.close()\n"
            "np.savez(%r, f=f, g=g, info=info)\n") % (ROOT, str(tmp_path / "in.npz"), m, d, k, method, bool(hetero), str(tmp_path / "out.npz"))
    env = dict(os.environ, GPZ_HIP_LIB=DEV_LIB, **{k_: str(v) for k_, v in switches.items()})
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=600)
    o = np.load(tmp_path / "out.npz")
    return float(o["f"]), o["g"], int(o["info"])
