"""Pins of the CPU oracle (oracle/gpz_oracle.py).

The reference cannot be executed here and ships no golden vectors, so the restatement is
pinned by properties derived from the reference's own code:
  1. its gradient is the derivative of its objective (minFunc derivative-check protocol,
     autoDif/autoGrad.m:34-45, derivativeCheck.m:29-40) for every method / option combination;
  2. the method-nesting identities implied by getPHI.m:26-40 and GPz.m:215-225;
  3. Psi = 0 == no Psi (getPHI.m:86,104), omega = 1 == default (GPz.m:20-22), mask = all == default (:16-18);
  4. an independent dense n x n Gaussian log-density (Woodbury) evaluation of GPz.m:65-82,110;
  5. a hand-computed m = 1, d = 1 case;
  6. the committed golden fixtures (regression).
"""
import math
import os

import numpy as np
import pytest

from oracle import gpz_oracle as O
from helpers import golden_names, load_golden, load_predict_golden, make_problem, rel

METHODS = ["GL", "VL", "GD", "VD", "GC", "VC"]


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("hetero", [True, False])
@pytest.mark.parametrize("psi,nanfrac", [(False, 0.0), (True, 0.0), (False, 0.4), (True, 0.4)])
def test_fd_gradient(method, hetero, psi, nanfrac):
    model, theta, X, Y, Psi, rng = make_problem(40, 3, 4, 2, method, hetero, seed=7, psi=psi, nanfrac=nanfrac)
    om = rng.random((40, 1)) + 0.5
    tr = rng.random(40) < 0.8
    r = O.GPz(theta, model, X, Y, Psi, om, tr, None)
    g = O.fd_gradient(lambda t: O.GPz(t, model, X, Y, Psi, om, tr, None).nlogML, theta)
    diff = np.max(np.abs(g - r.grad))
    assert diff <= 1e-4                      # the reference's own threshold (derivativeCheck.m:29)
    assert diff / np.max(np.abs(g)) <= 1e-6  # and the tighter bound an exact derivative meets


def _theta_with_gamma(model_to, model_from, theta_from, gamma_block):
    """theta for model_to sharing P / lnAlpha / b / v / lnTau with theta_from, with the given Gamma block."""
    md = model_from.m * model_from.d
    return np.concatenate([theta_from[:md], np.ravel(gamma_block, order="F"), theta_from[md + model_from.g_dim:]])


def test_method_nesting_identities():
    n, d, m, k = 60, 3, 5, 1
    model_gl, theta_gl, X, Y, _, rng = make_problem(n, d, m, k, "GL", True, seed=11)
    gam = theta_gl[m * d]
    ref = O.GPz(theta_gl, model_gl, X, Y)
    blocks = {
        "VL": np.full(m, gam), "GD": np.full(d, gam), "VD": np.full((m, d), gam),
        "GC": np.eye(d) * gam, "VC": np.repeat((np.eye(d) * gam)[:, :, None], m, axis=2),
    }
    md = m * d
    for method, blk in blocks.items():
        mdl = O.Model(m=m, d=d, k=k, method=method, heteroscedastic=True)
        th = _theta_with_gamma(mdl, model_gl, theta_gl, blk)
        r = O.GPz(th, mdl, X, Y)
        assert abs(r.nlogML - ref.nlogML) <= 1e-12 * abs(ref.nlogML), method
        assert rel(r.w, ref.w) < 1e-9
        # gradients: non-Gamma blocks agree; the Gamma blocks are related by the sums of GPz.m:215-225
        assert rel(r.grad[:md], ref.grad[:md]) < 1e-9
        assert rel(r.grad[md + mdl.g_dim:], ref.grad[md + 1:]) < 1e-9
        gG = r.grad[md:md + mdl.g_dim]
        if method in ("VL", "GD", "VD"):
            assert abs(gG.sum() - ref.grad[md]) <= 1e-9 * max(1.0, abs(ref.grad[md]))
        else:
            # d/dgamma of Gamma = gamma*I is the trace of the matrix gradient(s)
            G = gG.reshape((d, d, -1), order="F")
            tr = sum(np.trace(G[:, :, j]) for j in range(G.shape[2]))
            assert abs(tr - ref.grad[md]) <= 1e-9 * max(1.0, abs(ref.grad[md]))


@pytest.mark.parametrize("method", ["VD", "VC"])
def test_psi_zero_equals_no_psi(method):
    model, theta, X, Y, _, rng = make_problem(50, 3, 4, 1, method, True, seed=3)
    Psi0 = np.zeros((3, 3, 50)) if method == "VC" else np.zeros((50, 3))
    a = O.GPz(theta, model, X, Y, None)
    b = O.GPz(theta, model, X, Y, Psi0)
    assert abs(a.nlogML - b.nlogML) < 1e-12 * abs(a.nlogML)
    assert rel(b.grad, a.grad) < 1e-9


def test_defaults_equal_explicit():
    model, theta, X, Y, _, rng = make_problem(50, 3, 4, 2, "VD", True, seed=4)
    a = O.GPz(theta, model, X, Y)
    b = O.GPz(theta, model, X, Y, None, np.ones((50, 1)), np.ones(50, bool), None)
    assert a.nlogML == b.nlogML and np.array_equal(a.grad, b.grad)


def test_mask_equals_subset():
    model, theta, X, Y, _, rng = make_problem(80, 3, 4, 1, "VC", True, seed=5)
    tr = rng.random(80) < 0.7
    om = rng.random((80, 1)) + 0.5
    a = O.GPz(theta, model, X, Y, None, om, tr, ~tr)
    b = O.GPz(theta, model, X[tr], Y[tr], None, om[tr])
    assert abs(a.nlogML - b.nlogML) < 1e-13 * abs(a.nlogML)
    assert rel(a.grad, b.grad) < 1e-12
    c = O.GPz(theta, model, X[~tr], Y[~tr], None, om[~tr])          # validation statistics == training statistics
    PHIv, _, lnBv = O.getPHI(X, None, theta, model, ~tr)            # of the validation rows under the same w
    delta = PHIv @ a.w - Y[~tr]
    assert abs(a.stats["validRMSE"] - math.sqrt(np.sum(delta ** 2 * om[~tr]) / (~tr).sum())) < 1e-14
    assert c.stats["trainRMSE"] > 0


def test_dense_gaussian_density():
    """GPz.m:65-82,110 with omega = 1 is ln N(y | 0, B^-1 + PHI A^-1 PHI') (Woodbury + determinant lemma):
    pins SIGMA, inv_logdet, w, delta and the constants without sharing code with GPz()."""
    n, d, m = 30, 2, 6
    model, theta, X, Y, _, rng = make_problem(n, d, m, 1, "VD", False, seed=9)
    r = O.GPz(theta, model, X, Y)
    PHI, _, lnB = O.getPHI(X, None, theta, model)
    P, G, lnAlpha, b, v, lnTau = O.unpack_theta(theta, model)
    beta = np.exp(-lnB[:, 0])
    C = np.diag(1.0 / beta) + PHI @ np.diag(np.exp(-lnAlpha[:, 0])) @ PHI.T
    sign, ld = np.linalg.slogdet(C)
    logp = -0.5 * Y[:, 0] @ np.linalg.solve(C, Y[:, 0]) - 0.5 * ld - 0.5 * n * math.log(2 * math.pi)
    assert abs(-logp / n - r.nlogML) < 1e-10 * abs(r.nlogML)


def test_hand_computed_m1_d1():
    # one basis function, one input, two samples, homoscedastic: everything by hand
    X = np.array([[0.0], [1.0]]); Y = np.array([[1.0], [2.0]])
    p, gam, lnA, b = 0.5, 2.0, math.log(3.0), math.log(0.25)
    model = O.Model(m=1, d=1, k=1, method="GL", heteroscedastic=False)
    theta = np.array([p, gam, lnA, b])
    phi = np.exp(-0.5 * (X[:, 0] - p) ** 2 * gam ** 2)
    beta = math.exp(-b)
    sigma = beta * phi @ phi + 3.0
    w = beta * phi @ Y[:, 0] / sigma
    delta = phi * w - Y[:, 0]
    L = (-0.5 * beta * delta @ delta - 0.5 * 3.0 * w * w + 0.5 * lnA - 0.5 * math.log(sigma) + 0.5 * 2 * (-b)
         - 0.5 * math.log(2 * math.pi) * 2)
    r = O.GPz(theta, model, X, Y)
    assert abs(r.nlogML - (-L / 2)) < 1e-14
    assert abs(r.w[0, 0] - w) < 1e-15


def test_inv_logdet_truncation_and_values():
    rng = np.random.default_rng(0)
    A = rng.standard_normal((8, 20)); S = A @ A.T
    Xi, ld = O.inv_logdet(S)
    assert rel(Xi, np.linalg.inv(S)) < 1e-10 and abs(ld - np.linalg.slogdet(S)[1]) < 1e-10
    S2 = np.diag([1.0, 1e-20])                                     # second singular value below 2*eps(1)
    Xi2, ld2 = O.inv_logdet(S2)
    assert Xi2[1, 1] == 0.0 and ld2 == 0.0                          # inv_logdet.m:7-15: truncated, log-det of kept values


def test_dxy_and_groups():
    rng = np.random.default_rng(2)
    X = rng.standard_normal((20, 4)); Y = rng.standard_normal((7, 4))
    D = O.Dxy(X, Y)
    ref = ((X[:, None, :] - Y[None, :, :]) ** 2).sum(-1)
    assert rel(D, ref) < 1e-13
    Xn = X.copy(); Xn[3, 1] = np.nan; Xn[9, 1] = np.nan; Xn[5, [0, 2]] = np.nan
    gid, pats = O.nan_groups(Xn)
    assert pats.shape[0] == 3 and gid[0] == 0 and gid[3] == 1 and gid[9] == 1 and gid[5] == 2


def test_fixpsi_and_omega_shapes():
    sd = np.array([2.0, 4.0])
    P = O.fixPsi(np.array([1.0, 2.0, 3.0]), 3, sd, "VD")
    assert P.shape == (3, 2) and np.allclose(P[1], [2.0 / 4, 2.0 / 16])
    C = O.fixPsi(np.array([1.0, 2.0, 3.0]), 3, sd, "VC")
    assert C.shape == (2, 2, 3) and abs(C[1, 1, 2] - 3.0 / 16) < 1e-15 and C[0, 1, 2] == 0
    y = np.linspace(0, 1, 50)
    assert O.getOmega(y, "normalized").shape == (50, 1) and O.getOmega(y, "balanced").shape == (50, 1)


@pytest.mark.parametrize("name", golden_names())
def test_golden_regression(name):
    g, model, Psi, omega, training, validation = load_golden(name)
    r = O.GPz(g["theta"], model, g["X"], g["Y"], Psi, omega, training, validation)
    assert abs(r.nlogML - float(g["nlogML"])) <= 1e-12 * abs(float(g["nlogML"]))
    assert rel(r.grad, g["grad"]) < 1e-9
    r4 = O.GPz(g["theta"], model, g["X"], g["Y"], Psi, omega, training, validation, nargout=4)
    assert rel(r4.w, g["w"]) < 1e-9


@pytest.mark.parametrize("method", ["VD", "GL", "VC", "GC"])
def test_predict_noisy_reduces_to_predict_full(method):
    """predictNoisy (predictDiag.m:75-125 / predictCov.m:70-132) with Psi -> 0 must equal predictFull: the pairwise
    product-of-Gaussians sums collapse to PHI*iSigma_w*PHI' and gamma, VlnS vanish."""
    model, theta, X, Y, _, rng = make_problem(60, 3, 5, 2, method, True, seed=3)
    r4 = O.GPz(theta, model, X, Y, nargout=4)
    model.sets["best"] = {"theta": theta, "w": r4.w, "iSigma_w": r4.iSigma_w}
    Xs = rng.standard_normal((11, 3))
    full = O.predict(Xs, model)
    noisy = O.predict_noisy(Xs, np.full((11, 3), 1e-12), model)
    for i in range(4):
        assert rel(noisy[i], full[i]) < 1e-10
    assert np.abs(noisy[4]).max() < 1e-10
    wide = O.predict_noisy(Xs, np.full((11, 3), 0.3), model)
    assert (wide[4] > -1e-12).all() and (wide[1] > 0).all()          # gamma is a variance


def test_get_prior_is_a_distribution():
    model, theta, X, Y, _, rng = make_problem(80, 2, 6, 1, "VD", True, seed=8)
    pr = O.getPrior(X, None, theta, model)
    assert abs(pr.sum() - 1.0) < 1e-12 and (pr >= 0).all()


# ---- predict.m with missing values: predictMissing / predictNoisyMissing (predictDiag.m:127-297, predictCov.m:134-337) ----
def _pm_models(d, m, k, seed):
    """A VD model and the VC model with Gamma_j = diag(gamma_j): the same predictor through two different code paths."""
    model, theta, X, Y, _, rng = make_problem(60, d, m, k, "VD", True, seed=seed)
    r4 = O.GPz(theta, model, X, Y, nargout=4)
    pri = rng.random(m); pri /= pri.sum()
    model.sets["best"] = {"theta": theta, "w": r4.w, "iSigma_w": r4.iSigma_w, "priors": pri}
    mc = O.Model(m=m, d=d, k=k, method="VC", heteroscedastic=True)
    P, G, lnA, b, v, lnT = O.unpack_theta(theta, model)
    Gd = O.expand_gamma(G, model)
    Gc = np.zeros((d, d, m))
    for j in range(m):
        Gc[:, :, j] = np.diag(Gd[j])
    thc = np.concatenate([P.ravel(order="F"), Gc.ravel(order="F"), theta[m * d + m * d:]])
    mc.sets["best"] = dict(model.sets["best"], theta=thc)
    return model, mc, rng


def test_predict_missing_diag_and_cov_restatements_agree():
    """predictDiag.m and predictCov.m are separate code (GEMM-style sums vs per-pair d x d solves, X_hat / Psi_hat
    conditioning); on a diagonal covariance they must give the same predictor.  Patterns whose [o u] ordering is
    an involution only: for the others predictCov.m:268's scatter through `unshuffle` differs (next test)."""
    d, m, k, n = 3, 4, 2, 9
    model, mc, rng = _pm_models(d, m, k, 5)
    Xs = rng.standard_normal((n, d))
    pats = [[2], [1], [], [1, 2]]
    for r in range(n):
        Xs[r, pats[r % len(pats)]] = np.nan
    a = O.predict_any(Xs, model); c = O.predict_any(Xs, mc)
    for x, y in zip(a, c):
        assert rel(x, y) < 1e-12
    Psi = rng.gamma(1.0, 0.1, (n, d))
    a = O.predict_any(Xs, model, Psi=Psi); c = O.predict_any(Xs, mc, Psi=Psi)
    for x, y in zip(a, c):
        assert rel(x, y) < 1e-12
    z = O.predict_any(Xs, model, Psi=np.full((n, d), 1e-13))       # Psi -> 0: predictNoisyMissing -> predictMissing
    p0 = O.predict_any(Xs, model)
    for x, y in zip(z, p0):
        assert rel(x, y) < 1e-10


def test_predict_noisy_missing_cov_unshuffle_quirk_is_kept():
    """predictCov.m:266-268 assigns T*Psi_oo*T' through `unshuffle` (the inverse of [find(o) find(~o)]); as an
    assignment target that is the intended placement only when the permutation is its own inverse.  With the first
    dimension missing (perm = [2 3 1]) the cov path therefore differs from the diag path; without Psi it does not."""
    d, m, k, n = 3, 4, 1, 4
    model, mc, rng = _pm_models(d, m, k, 6)
    Xs = rng.standard_normal((n, d)); Xs[:, 0] = np.nan
    a = O.predict_any(Xs, model); c = O.predict_any(Xs, mc)
    assert rel(a[1], c[1]) < 1e-12
    Psi = rng.gamma(1.0, 0.1, (n, d))
    a = O.predict_any(Xs, model, Psi=Psi); c = O.predict_any(Xs, mc, Psi=Psi)
    assert rel(a[1], c[1]) > 1e-5


def test_predict_missing_marginalises_a_single_basis_exactly():
    """m = 1: the conditional mixture is the basis function itself, so PHI for a row with missing dimensions is the
    basis function of the observed dimensions times sqrt|Sigma_uu|-style constants — hand-computed."""
    d, m = 2, 1
    model = O.Model(m=m, d=d, k=1, method="VD", heteroscedastic=False)
    P = np.array([[0.3, -0.2]]); G = np.array([[1.5, 0.7]])
    theta = np.concatenate([P.ravel(), G.ravel(), [0.1], [-0.4]])
    model.sets["best"] = {"theta": theta, "w": np.array([[0.8]]), "iSigma_w": np.array([[[0.05]]]), "priors": np.ones(1)}
    x0 = 0.9
    out = O.predict_any(np.array([[x0, np.nan]]), model)
    sig = G[0] ** -2.0
    No = math.exp(-0.5 * (x0 - P[0, 0]) ** 2 / sig[0] - 0.5 * math.log(sig[0]))
    Nij = math.exp(-0.5 * math.log(2 * sig[1]))
    phi = No * 1.0 * Nij * math.exp(-0.5 * np.sum(np.log(G[0] ** 2)))
    assert abs(out[5][0, 0] - phi) < 1e-15 and abs(out[0][0, 0] - 0.8 * phi) < 1e-15


@pytest.mark.parametrize("name", golden_names("p_"))
def test_predict_golden_regression(name):
    g, model, Psi = load_predict_golden(name)
    out = O.predict_any(g["Xs"], model, Psi=Psi)
    for i, key in enumerate(("mu", "sigma", "nu", "beta_i", "gamma", "PHI")):
        assert rel(out[i], g[key]) < 1e-12, key
    Xn = (g["Xs"] - model.muX) / model.sdX
    assert rel(O.getPrior(Xn, O.fixPsi(Psi, Xn.shape[0], model.sdX, model.method), g["theta"], model), g["prior"]) < 1e-12


@pytest.mark.parametrize("name", golden_names("s_"))
def test_pinv_golden_regression(name):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    Xi, ld = O.inv_logdet(z["S"])
    assert np.linalg.matrix_rank(Xi) == int(z["rank"]) and rel(Xi, z["Xi"]) < 1e-9 and abs(ld - float(z["logdet"])) < 1e-9


# ---- independent evaluations of the objective AND of the gradient (SURVEY.md §8c pin 5) ----------------------------
# tests/mp_reference.py shares no code with the oracle: one general ln PHI expression instead of the reference's four
# branches, LU solve + determinant instead of the SVD inverse, and the gradient by differencing at 50 digits (mpmath) or by
# autograd (torch fp64) instead of the hand-derived blocks of GPz.m:89-231.
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("psi,nanfrac", [(False, 0.0), (True, 0.0), (False, 0.4), (True, 0.4)])
def test_mpmath_50_digit_objective_and_gradient(method, psi, nanfrac):
    import mp_reference as R
    n, d, m, k = 14, 2, 3, 1
    model, theta, X, Y, Psi, rng = make_problem(n, d, m, k, method, True, seed=5, psi=psi, nanfrac=nanfrac)
    om = rng.random((n, 1)) + 0.5
    r = O.GPz(theta, model, X, Y, Psi, om)
    f = R.nlogml(theta, model.method, m, d, k, True, X, Y, Psi, om)
    assert abs(float(f) - r.nlogML) <= 1e-13 * abs(float(f))
    g = np.array([float(v) for v in R.gradient(theta, model.method, m, d, k, True, X, Y, Psi, om)])
    assert np.max(np.abs(g - r.grad)) <= max(1e-12, 50 * r.cond * 2.2e-16) * np.max(np.abs(g))


@pytest.mark.parametrize("method,hetero,k", [("VD", False, 1), ("VC", False, 2), ("VL", True, 2)])
def test_mpmath_50_digit_homoscedastic_and_multi_output(method, hetero, k):
    import mp_reference as R
    n, d, m = 12, 2, 3
    model, theta, X, Y, Psi, rng = make_problem(n, d, m, k, method, hetero, seed=15)
    r = O.GPz(theta, model, X, Y)
    f = R.nlogml(theta, model.method, m, d, k, hetero, X, Y)
    g = np.array([float(v) for v in R.gradient(theta, model.method, m, d, k, hetero, X, Y)])
    assert abs(float(f) - r.nlogML) <= 1e-13 * abs(float(f))
    assert np.max(np.abs(g - r.grad)) <= max(1e-12, 50 * r.cond * 2.2e-16) * np.max(np.abs(g))


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("hetero", [True, False])
@pytest.mark.parametrize("psi,nanfrac,k", [(False, 0.0, 1), (True, 0.0, 2), (False, 0.3, 2), (True, 0.3, 1)])
def test_torch_autograd_gradient(method, hetero, psi, nanfrac, k):
    import torch
    import mp_reference as R
    n, d, m = 150, 3, 6
    model, theta, X, Y, Psi, rng = make_problem(n, d, m, k, method, hetero, seed=25, psi=psi, nanfrac=nanfrac)
    om = rng.random((n, 1)) + 0.5
    r = O.GPz(theta, model, X, Y, Psi, om)
    th = torch.tensor(theta, requires_grad=True)
    f = R.torch_nlogml(th, model.method, m, d, k, hetero, X, Y, Psi, om)
    f.backward()
    g = th.grad.numpy()
    assert abs(f.item() - r.nlogML) <= 1e-12 * abs(r.nlogML)
    assert np.max(np.abs(g - r.grad)) <= max(1e-11, 50 * r.cond * 2.2e-16) * np.max(np.abs(g))


@pytest.mark.parametrize("method,hetero,weights", [("VC", True, True), ("VD", True, False), ("VC", False, False), ("VD", False, True)])
def test_vectorised_baseline_matches_the_statement_level_oracle(method, hetero, weights):
    """oracle/gpz_vectorised.py (the "vectorised CPU" mode bench.py times next to the as-written one) is the same function."""
    from oracle import gpz_vectorised as V
    from helpers import make_problem
    model, theta, X, Y, _, rng = make_problem(700, 6, 30, 1, method, hetero, seed=5)
    om = rng.random((700, 1)) + 0.5 if weights else None
    ref = O.GPz(theta, model, X, Y, None, om)
    f, g = V.GPz(theta, model, X, Y, om, chunk=256)
    assert abs(f - ref.nlogML) <= 1e-11 * abs(ref.nlogML)
    assert np.max(np.abs(g - ref.grad)) <= max(1e-10, 50 * ref.cond * 2.2e-16) * np.max(np.abs(ref.grad))


# ---- independent pin of the prediction branches (VERDICT r02 "missing 4") --------------------------------------------------
# tests/quad_reference.py evaluates predict.m's outputs as the Gaussian expectations they are (Gauss-Hermite quadrature of
# phi'w, (phi'w)^2, phi' iSigma_w phi, phi'v, (phi'v)^2 under the input's distribution); it shares no formula with
# predictDiag.m / predictCov.m or with the oracle's restatement of them.
def _quad_case(method, d, m, k, seed, psi, miss):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((60, d))
    Y = np.sin(X @ rng.standard_normal((d, k))) + 0.1 * rng.standard_normal((60, k))
    Y -= Y.mean(0)
    model, theta = O.init_theta(X, Y, method, m, True, rng)
    theta = theta + 0.1 * rng.standard_normal(theta.size)
    model.muX = 0.1 * rng.standard_normal(d); model.sdX = 1.0 + 0.5 * rng.random(d); model.muY = rng.standard_normal(k)
    r4 = O.GPz(theta, model, X, Y, nargout=4)
    pri = rng.random(m) + 0.2
    model.sets["best"] = {"theta": theta, "w": r4.w, "iSigma_w": r4.iSigma_w, "priors": pri / pri.sum()}
    ns = 7
    Xs = rng.standard_normal((ns, d)) * model.sdX + model.muX
    Psi = None
    if psi == "diag":
        Psi = rng.gamma(2.0, 0.05, (ns, d))
    elif psi == "full":
        Psi = np.zeros((d, d, ns))
        for i in range(ns):
            B = 0.3 * rng.standard_normal((d, d))
            Psi[:, :, i] = B @ B.T + 0.02 * np.eye(d)
    if miss:
        Xs[rng.integers(0, ns, 4), rng.integers(0, d, 4)] = np.nan
        Xs[np.isnan(Xs).all(1), 0] = 0.3
    return model, theta, Xs, Psi


QUAD_CASES = [("VD", 2, 3, 2, "diag", False), ("GL", 1, 3, 1, "diag", False), ("VC", 2, 3, 1, "full", False),
              ("GC", 2, 2, 2, "diag", False), ("VD", 2, 3, 1, None, True), ("VL", 2, 3, 2, "diag", True),
              ("VC", 2, 3, 1, None, True), ("GC", 2, 3, 1, "full", True), ("VC", 2, 2, 2, "diag", True)]


@pytest.mark.parametrize("method,d,m,k,psi,miss", QUAD_CASES)
def test_predict_branches_against_quadrature(method, d, m, k, psi, miss):
    """predictNoisy / predictMissing / predictNoisyMissing of the oracle against the quadrature reference: every output
    (mu, sigma, nu, beta_i, gamma, PHI) to 1e-9."""
    import quad_reference as Q
    model, theta, Xs, Psi = _quad_case(method, d, m, k, 900 + d + m + k, psi, miss)
    st = model.sets["best"]
    ref = Q.predict(Xs, model, theta, st["w"], st["iSigma_w"], st["priors"], Psi)
    got = O.predict_any(Xs, model, Psi=Psi)
    for name, a, b in zip(("mu", "sigma", "nu", "beta_i", "gamma", "PHI"), got, ref):
        assert rel(a, b) <= 1e-9, (name, rel(a, b))
