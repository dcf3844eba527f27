"""GC/VC with input noise in fp64 beyond the register kernels (10 < d <= 64): the MFMA-accumulator kernels of k_cpsi4.hip (four
pairs per wave, d <= 32) and k_cpsi.hip (one pair per wave, d <= 64) - getPHI.m:78-89, GPz.m:164-185 - against the oracle:
objective, gradient, statistics, solve, PHI; missing values (identity block + zero Delta), weights, training / validation masks,
every tile count, d not a multiple of 4, m not a multiple of the 4 / 16 pairs a wave / workgroup takes, row shards."""
import numpy as np
import pytest

import gpz_amd
from oracle import gpz_oracle as O
from helpers import make_problem, recondition_gamma, rel
from test_gpu_parity import phi_tol
from test_wide import _gate

pytestmark = pytest.mark.gpu


def _problem(n, d, m, k, method, hetero, seed, nanfrac=0.0):
    """make_problem with the Gamma blocks redrawn as gamma_j (I + 0.3 G / sqrt(d)): its 0.05 N(0,1) perturbation of gamma_j I is larger
    than gamma_j itself at d ~ 20 (cond(Gamma'Gamma) ~ 1e8: both sides of the comparison are rounding noise there)"""
    model, theta, X, Y, Psi, rng = make_problem(n, d, m, k, method, hetero, seed=seed, psi=True, nanfrac=nanfrac)
    recondition_gamma(model, theta, rng)
    return model, theta, X, Y, Psi, rng


def _loose(model, theta):
    # the reference's chain dS -> diS = -Sigma dS Sigma -> dGamma goes through inv(Gamma'Gamma) twice and loses cond^1.5 * eps
    # (DESIGN.md section 4); oracle and HIP path run the same formula and differ by that rounding noise
    c = phi_tol(model, theta) / (200.0 * 2.2e-16)
    return max(2.0, 10.0 * c ** 1.5 * 2.2e-16 / phi_tol(model, theta))


@pytest.mark.parametrize("method", ["VC", "GC"])
@pytest.mark.parametrize("d", [11, 12, 13, 16, 19, 20, 21, 24, 27, 28, 32, 33, 41, 48, 49, 64])
@pytest.mark.parametrize("nanfrac", [0.0, 0.3])
def test_pair_kernels_every_tile_count(method, d, nanfrac):
    if d > 32 and nanfrac and method == "GC":
        pytest.skip("covered by VC")
    n, m = (230, 7) if d <= 32 else (90, 5)
    model, theta, X, Y, Psi, rng = _problem(n, d, m, 1, method, True, 500 + d, nanfrac)
    om = rng.random((n, 1)) + 0.5
    tr = rng.random(n) < 0.8
    _gate(model, theta, X, Y, Psi, om, tr, ~tr, loose=_loose(model, theta))


@pytest.mark.parametrize("d,m,n", [(14, 1, 37), (20, 17, 130), (24, 33, 64), (30, 4, 3)])
def test_pair_kernels_ragged_sizes(d, m, n):
    """fewer pairs than a wave holds, m / n not multiples of 4 or 16, fewer rows than chunks"""
    model, theta, X, Y, Psi, rng = _problem(n, d, m, 1, "VC", True, 700 + d)
    _gate(model, theta, X, Y, Psi, loose=_loose(model, theta))


def test_pair_kernels_two_outputs_and_homoscedastic():
    for hetero, k in ((True, 2), (False, 1)):
        model, theta, X, Y, Psi, rng = _problem(150, 18, 6, k, "VC", hetero, 810 + k)
        _gate(model, theta, X, Y, Psi, loose=_loose(model, theta))


@pytest.mark.parametrize("d", [20, 36])
def test_pair_kernels_over_row_shards(d):
    """three loopback shards (rows split, records summed by the reducer) against the single context"""
    n, m = 300, 6
    model, theta, X, Y, Psi, rng = _problem(n, d, m, 1, "VC", True, 900 + d)
    ctx = gpz_amd.GPzContext(model, X, Y, Psi)
    try:
        f1, g1 = ctx.eval(theta)
    finally:
        ctx.close()
    multi = gpz_amd.GPzMulti(model, X, Y, Psi, n_gpus=3, reducer="loopback")
    try:
        f3, g3 = multi.eval(theta)
    finally:
        multi.close()
    assert abs(f3 - f1) <= 1e-12 * abs(f1) and rel(g3, g1) <= 1e-9


def test_fp64_and_fp32_pair_kernels_agree_at_d20():
    """config 5's shape in the reference's own precision (k_cpsi4) against the fp32 pair kernels (k_psi32) on diagonal Psi"""
    n, d, m = 400, 20, 16
    model, theta, X, Y, _, rng = _problem(n, d, m, 1, "VC", True, 77)
    var = rng.gamma(1.0, 0.2, (n, d))
    Psi = np.zeros((d, d, n))
    Psi[np.arange(d), np.arange(d), :] = var.T
    out = {}
    for dtype in ("f64", "f32"):
        ctx = gpz_amd.GPzContext(model, X, Y, Psi, dtype=dtype)
        try:
            out[dtype] = ctx.eval(theta)
        finally:
            ctx.close()
    ref = O.GPz(theta, model, X, Y, Psi)
    assert abs(out["f64"][0] - ref.nlogML) <= 1e-9 * abs(ref.nlogML)
    assert abs(out["f32"][0] - out["f64"][0]) <= 1e-4 * abs(out["f64"][0]) and rel(out["f32"][1], out["f64"][1]) <= 1e-3


@pytest.mark.parametrize("method", ["VC", "GC"])
@pytest.mark.parametrize("d,k,cube", [(12, 1, False), (17, 2, True), (20, 1, True), (29, 1, False), (32, 2, False), (37, 1, False),
                                      (44, 1, True), (48, 1, False)])
def test_predict_with_input_noise_on_the_pair_kernels(method, d, k, cube):
    """predictNoisy for GC/VC at 10 < d <= 48 (predictCov.m:70-132): one sweep per (sample, pair) on 4 x 4 tiles; GC shares the
    factorisation over the pairs (Cij = Sigma/2).  Psi as n x d variances or as full d x d x n cubes; ns not a multiple of 16."""
    m, ns = 6, 37
    model, theta, X, Y, _, rng = _problem(260, d, m, k, method, True, 4200 + d)
    r4 = O.GPz(theta, model, X, Y, nargout=4)
    model.sets["best"] = {"theta": theta, "w": r4.w, "iSigma_w": r4.iSigma_w}
    Xs = rng.standard_normal((ns, d))
    if cube:
        Psi = np.zeros((d, d, ns))
        for i in range(ns):
            B = 0.2 * rng.standard_normal((d, d))
            Psi[:, :, i] = B @ B.T
    else:
        Psi = rng.gamma(1.0, 0.1, (ns, d))
    ref = O.predict_noisy(Xs, Psi, model)
    out = gpz_amd.predict(Xs, model, Psi=Psi)
    tol = max(1e-9, phi_tol(model, theta))
    for i, name in enumerate(("mu", "sigma", "nu", "beta_i", "gamma", "PHI")):
        assert rel(out[i], ref[i]) <= tol, name


@pytest.mark.parametrize("method", ["GC", "VC"])
@pytest.mark.parametrize("d,noisy,k", [(12, False, 1), (13, True, 2), (22, False, 1), (22, True, 1), (30, True, 1), (32, False, 2)])
def test_predict_with_missing_values_on_the_pair_kernels(method, d, noisy, k):
    """predictMissing / predictNoisyMissing for GC/VC at 10 < d <= 32 (predictCov.m:134-337): the record sums
    sum_l N(X_hat_l - c; C + Psi_hat_l) Pio_l as sweeps on 4 x 4 tiles, four components per wave; without input noise the
    factorisation is shared by the rows of the group.  Several NaN patterns, rows not a multiple of anything."""
    m, ns = 5, 11
    model, theta, X, Y, _, rng = _problem(200, d, m, k, method, True, 6100 + d)
    r4 = O.GPz(theta, model, X, Y, nargout=4)
    pri = rng.random(m) + 0.2
    model.sets["best"] = {"theta": theta, "w": r4.w, "iSigma_w": r4.iSigma_w, "priors": pri / pri.sum()}
    Xs = rng.standard_normal((ns, d))
    Xs[:4, 1] = np.nan
    Xs[4:9, [0, d - 1, d // 2]] = np.nan
    Xs[9:, 2:d - 3] = np.nan                                  # most dimensions missing
    Psi = None
    if noisy:
        Psi = np.zeros((d, d, ns))
        for i in range(ns):
            B = 0.2 * rng.standard_normal((d, d))
            Psi[:, :, i] = B @ B.T
    ref = O.predict_any(Xs, model, Psi=Psi)
    out = gpz_amd.predict(Xs, model, Psi=Psi)
    tol = max(1e-8, 10.0 * phi_tol(model, theta))
    for i, name in enumerate(("mu", "sigma", "nu", "beta_i", "gamma", "PHI")):
        assert rel(out[i], ref[i]) <= tol, (name, rel(out[i], ref[i]))


def test_predict_with_missing_values_wide_and_many_outputs_takes_the_scratch_kernels():
    """k = 9 > 8 at d = 22: the 3k sums do not fit the record kernels' 24 slots per pass, the scratch-resident kernels (runtime d <= 32)
    take the group"""
    d, m, k, ns = 22, 4, 9, 7
    model, theta, X, Y, _, rng = _problem(160, d, m, k, "GC", True, 6400)
    r4 = O.GPz(theta, model, X, Y, nargout=4)
    pri = rng.random(m) + 0.2
    model.sets["best"] = {"theta": theta, "w": r4.w, "iSigma_w": r4.iSigma_w, "priors": pri / pri.sum()}
    Xs = rng.standard_normal((ns, d))
    Xs[:, [1, 7, 20]] = np.nan
    ref = O.predict_any(Xs, model)
    out = gpz_amd.predict(Xs, model)
    tol = max(1e-8, 10.0 * phi_tol(model, theta))
    for i, name in enumerate(("mu", "sigma", "nu", "beta_i", "gamma", "PHI")):
        assert rel(out[i], ref[i]) <= tol, (name, rel(out[i], ref[i]))


@pytest.mark.parametrize("d,switch", [(20, "GPZ_CPSI4_OFF"), (20, "GPZ_CPSI_OFF"), (40, "GPZ_CPSI4_OFF")])
def test_the_pair_kernel_routes_agree_with_each_other(d, switch, tmp_path):
    """the same evaluation through the route a developer switch selects in a fresh process on the developer build of the library (4 x 4 tiles -> 16 x 16 tiles -> general
    kernels) against this process's default route: three independent implementations of getPHI.m:78-89 / GPz.m:164-185"""
    from helpers import eval_with_dev_switches
    n, m = 120, 5
    model, theta, X, Y, Psi, rng = _problem(n, d, m, 1, "VC", True, 7300 + d)
    ctx = gpz_amd.GPzContext(model, X, Y, Psi)
    try:
        f, g = ctx.eval(theta)
    finally:
        ctx.close()
    of, og, _ = eval_with_dev_switches(tmp_path, "VC", m, d, 1, True, theta, X, Y, Psi, {switch: "1"})
    o = {"f": of, "g": og}
    assert abs(float(o["f"]) - f) <= 1e-11 * abs(f)
    assert rel(o["g"], g) <= max(1e-9, _loose(model, theta) * phi_tol(model, theta))


@pytest.mark.parametrize("method", ["VC", "GC"])
@pytest.mark.parametrize("d", [25, 30, 36, 50])
def test_missing_values_without_input_noise_wide(method, d):
    """GC/VC with missing values and no Psi at d = 25 ... 50: the per-pattern route on the runtime-d kernels (routing these through
    the pair kernels with Psi = 0 was measured and dropped: the per-(pattern, basis) finish dominates either way - d = 32,
    ~2000 patterns: 540 ms per pattern route, 890 ms through the pair kernels); several missing dimensions per row, validation split"""
    n, m = 260, 6
    model, theta, X, Y, _, rng = _problem(n, d, m, 1, method, True, 8100 + d)
    miss = rng.random((n, d)) < 0.08
    miss[:, 0] = False
    X = X.copy()
    X[miss] = np.nan
    om = rng.random((n, 1)) + 0.5
    tr = rng.random(n) < 0.8
    _gate(model, theta, X, Y, None, om, tr, ~tr, loose=_loose(model, theta))


@pytest.mark.parametrize("d", [12, 20])
def test_gc_dense_phi_build_far_from_the_origin(d):
    """GC + Psi, 10 < d <= 32 builds PHI as one dense product of the EXPANDED quadratic form x'M^-1 x - 2 p'M^-1 x + p'M^-1 p
    (k_cpsi4_minv<QROW> x k_gcq_tab), which cancels (|x| / |x - p|)^2 eps when the inputs sit far from the origin.  Both factors
    are taken about the mean basis centre: inputs and centres shifted by 3e4 length scales (1e-7 of ln PHI uncentred) must still
    give the oracle's PHI - whose Delta = x - p never sees the offset - to the tolerance of the unshifted problem."""
    n, m = 200, 9
    model, theta, X, Y, Psi, rng = _problem(n, d, m, 1, "GC", True, 8100 + d)
    shift = 3.0e4 * (1.0 + rng.random(d))
    Xs = X + shift
    ts = theta.copy()
    ts[:m * d] = (theta[:m * d].reshape(d, m) + shift[:, None]).ravel()      # P is m x d column-major: P(j, a) at j + m a
    ref = O.getPHI(X, Psi, theta, model, None)[0]                            # the same problem about the origin
    ctx = gpz_amd.GPzContext(model, Xs, Y, Psi)
    try:
        assert "k_cpsi4" in ctx.route()
        f, g = ctx.eval(ts)
        PHI = ctx.phi()
    finally:
        ctx.close()
    assert rel(PHI, ref) <= 1e-9, rel(PHI, ref)
    r = O.GPz(theta, model, X, Y, Psi)
    assert abs(f - r.nlogML) <= 1e-9 * abs(r.nlogML)
