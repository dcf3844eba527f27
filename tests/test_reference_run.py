"""The reference's own MATLAB files, executed (oracle/mlite.py + oracle/run_reference.py -> tests/golden/ref_*.npz), as the
pin of the oracle and — on the GPU — of the HIP path.

* CPU: the oracle's restatement against what GPz.m / getPHI.m / inv_logdet.m / predict.m / predictDiag.m / predictCov.m / fixPsi.m /
  getPrior.m / Dxy.m returned on the same inputs; and, where /root/reference exists, a re-execution of those files that must
  reproduce the committed vectors (so the fixtures are what the reference's text computes, not something edited by hand).
* GPU: the HIP path through the C ABI against the same vectors, at the gates of BASELINE.md section 6.
"""
import glob
import os

import numpy as np
import pytest

from oracle import gpz_oracle as O
from oracle import mlite as ML
from oracle import run_reference as RR
from helpers import GOLDEN, grad_tol, rel

GPZ = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "ref_gpz_*.npz")))
PRED = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "ref_predict_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def gpz_inputs(g):
    model = O.Model(m=int(g["m"]), d=int(g["d"]), k=int(g["k"]), method=str(g["method"]), heteroscedastic=bool(int(g["heteroscedastic"])))
    Psi = g["Psi"] if int(g["has_psi"]) else None
    return model, g["theta"], g["X"], g["Y"], Psi, g["omega"], g["training"].astype(bool), g["validation"].astype(bool)


def cov_cond(model, theta):
    if model.method[1] != "C":
        return 1.0
    P, G, *_ = O.unpack_theta(theta, model)
    Gam = O.expand_gamma(G, model)
    return max(np.linalg.cond(Gam[:, :, j].T @ Gam[:, :, j]) for j in range(Gam.shape[2]))


def predict_inputs(g):
    model = O.Model(m=int(g["m"]), d=int(g["d"]), k=int(g["k"]), method=str(g["method"]), heteroscedastic=True)
    model.muX, model.sdX, model.muY = g["muX"], g["sdX"], g["muY"]
    model.sets["best"] = {"theta": g["theta"], "w": g["w"], "iSigma_w": g["iSigma_w"], "priors": g["priors"]}
    return model, g["Xs"], (g["Psi"] if int(g["has_psi"]) else None)


def test_fixture_inventory():
    assert len(GPZ) == len(RR.GPZ_CASES) == 30 and len(PRED) == len(RR.PREDICT_CASES) == 24
    assert os.path.exists(os.path.join(GOLDEN, "ref_misc.npz")) and os.path.exists(os.path.join(GOLDEN, "ref_lbfgs_mem.npz"))


# ---- CPU: oracle against the executed reference ------------------------------------------------------------------------
@pytest.mark.parametrize("name", GPZ)
def test_oracle_against_the_executed_reference_gpz(name):
    g = load(name)
    model, theta, X, Y, Psi, om, tr, va = gpz_inputs(g)
    ref = O.GPz(theta, model, X, Y, Psi, om, tr, va)
    tol = max(1e-11, 50.0 * ref.cond * 2.2e-16, 50.0 * cov_cond(model, theta) * 2.2e-16)
    assert abs(ref.nlogML - float(g["nlogML"])) <= 1e-12 * abs(float(g["nlogML"]))
    assert rel(ref.grad, g["grad"]) <= tol
    for key in ("trainRMSE", "trainLL", "validRMSE", "validLL"):
        assert abs(ref.stats[key] - float(g[key])) <= 1e-12 * max(1.0, abs(float(g[key]))), key
    r4 = O.GPz(theta, model, X, Y, Psi, om, tr, va, nargout=5)
    assert rel(r4.w, g["w"]) <= tol and rel(r4.iSigma_w, g["iSigma_w"]) <= tol and rel(r4.PHI, g["PHI"]) <= 1e-12
    assert rel(np.atleast_1d(r4.nlogML), g["nlogML_solve"]) <= 1e-12


@pytest.mark.parametrize("name", PRED)
def test_oracle_against_the_executed_reference_predict(name):
    g = load(name)
    model, Xs, Psi = predict_inputs(g)
    out = O.predict_any(Xs, model, Psi=Psi)
    tol = max(1e-11, 200.0 * cov_cond(model, g["theta"]) * 2.2e-16)
    for key, val in zip(("mu", "sigma", "nu", "beta_i", "gamma", "PHIs"), out):
        assert rel(val, g[key]) <= tol, (key, rel(val, g[key]))


def test_oracle_against_the_executed_reference_misc():
    g = load("ref_misc")
    assert rel(O.Dxy(g["dxy_X"], g["dxy_Y"]), g["dxy_D"]) <= 1e-14
    Xi, ld = O.inv_logdet(g["il_A"])
    assert rel(Xi, g["il_Xi"]) <= 1e-12 and abs(ld - float(g["il_logdet"])) <= 1e-12 * abs(float(g["il_logdet"]))
    Xi, ld = O.inv_logdet(g["il2_A"])                       # rank 5 of 9: the truncating branch of inv_logdet.m:7-12
    assert rel(Xi, g["il2_Xi"]) <= 1e-9 and abs(ld - float(g["il2_logdet"])) <= 1e-10 * abs(float(g["il2_logdet"]))
    for key, args in (("om_balanced", ("balanced",)), ("om_balanced_w", ("balanced", 0.05)), ("om_normalized", ("normalized",))):
        assert rel(O.getOmega(g["om_Y"], *args), g[key]) <= 1e-14, key
    from gpz_amd import host as H                                   # host.fixPsi is NumPy: checked here as well
    for key in ("nd", "n1", "cube"):
        for method in ("VD", "VC"):
            want = g["fp_%s_%s" % (key, method)]
            assert rel(O.fixPsi(g["fp_in_" + key], 6, g["fp_sdX"], method), want) <= 1e-15, (key, method)
            assert rel(H.fixPsi(g["fp_in_" + key], 6, g["fp_sdX"], method), want) <= 1e-15, (key, method)
    for tag, method in (("vd", "VD"), ("vc", "VC")):
        model = O.Model(m=4, d=3, k=1, method=method, heteroscedastic=True)
        sel = g[tag + "_sel"].astype(bool)
        PHI, Gam, lnb, N = O.getPHI(g[tag + "_X"], g[tag + "_Psi"], g[tag + "_theta"], model, sel, want_N=True)
        tol = max(1e-11, 200.0 * cov_cond(model, g[tag + "_theta"]) * 2.2e-16)
        assert rel(PHI, g[tag + "_PHI"]) <= tol and rel(lnb, g[tag + "_lnBeta_i"]) <= tol and rel(N, g[tag + "_N"]) <= tol
        assert rel(Gam, g[tag + "_Gamma"]) == 0.0
        assert rel(O.getPrior(g[tag + "_X"], g[tag + "_Psi"], g[tag + "_theta"], model, sel), g[tag + "_prior"]) <= 1e-9


def test_minfunc_restatement_against_the_executed_lbfgs_files():
    """oracle/minfunc_oracle.py's lbfgsAdd / lbfgsProd against what lbfgsAdd.m / lbfgsProd.m returned: ring indices, rejected pairs,
    Hdiag and the direction after every call."""
    from oracle import minfunc_oracle as MF
    z = load("ref_lbfgs_mem")
    p, corr = int(z["p"]), int(z["corrections"])
    S = np.zeros((p, corr)); Y = np.zeros((p, corr)); YS = np.zeros(corr)
    start, end, hd = 1, 0, 1.0
    for it in range(z["T"].size):
        s = z["T"][it] * z["D"][it]
        y = z["G"][it + 1] - z["G"][it]
        start, end, hd, skipped = MF.lbfgsAdd(y, s, S, Y, YS, start, end, hd)
        assert (not skipped) == bool(z["added"][it]) and start == int(z["starts"][it]) and end == int(z["ends"][it])
        assert abs(hd - float(z["Hdiag"][it])) <= 1e-14 * abs(hd)
        if end > 0:
            assert rel(MF.lbfgsProd(z["G"][it + 1], S, Y, YS, start, end, hd), z["directions"][it]) <= 1e-12
    assert rel(S, z["S"]) <= 1e-15 and rel(Y, z["Y"]) <= 1e-13 and rel(YS, z["YS"]) <= 1e-13


@pytest.mark.skipif(not ML.available(), reason="the reference tree exists only in the build container")
def test_committed_vectors_are_what_the_reference_files_return():
    """Re-executes the reference's .m files and compares with every committed ref_*.npz: the vectors are the reference's own
    outputs on the recorded inputs (bit for bit up to BLAS summation order), not data that could drift from it."""
    for name, make in RR.all_fixtures().items():
        if name.startswith("ref_gpz_") and not name.endswith(("_p0_n0", "_p1_n1")):
            continue                                    # a third of the GPz cases keeps the CPU suite short; all predict / misc cases
        fresh, old = make(), load(name)
        assert set(fresh) == set(old), name
        for key, val in fresh.items():
            a, b = np.asarray(val), old[key]
            if a.dtype.kind in "fc" and a.size:
                assert np.allclose(a, b, rtol=1e-13, atol=1e-300, equal_nan=True), (name, key)
            else:
                assert np.array_equal(a, b), (name, key)


def test_interpreter_basics():
    """The MATLAB semantics the interpreter has to get right, on expressions whose value is known."""
    ip = ML.Interp(ref_dir="/nonexistent")

    def ev(src, **vars):
        f = ML.Parser(ML.lex("function r = t()\n r = %s;\nend\n" % src)).parse_file()["t"]
        scope = {"__globals__": set()}
        scope.update({k: (v if isinstance(v, str) else ML.mat(v)) for k, v in vars.items()})
        ip.run_block(f["body"], scope)
        return scope["r"]

    A = np.arange(1.0, 13.0).reshape(3, 4, order="F")
    assert np.array_equal(ev("A(2,:)", A=A), A[1:2, :]) and np.array_equal(ev("A(:)", A=A), A.reshape(-1, 1, order="F"))
    assert np.array_equal(ev("A(end,end)", A=A), [[12.0]]) and np.array_equal(ev("A(5)", A=A), [[5.0]])
    assert np.array_equal(ev("sum(A)", A=A), A.sum(0, keepdims=True)) and np.array_equal(ev("sum(A,2)", A=A), A.sum(1, keepdims=True))
    assert np.array_equal(ev("sum(A(1,:))", A=A), [[A[0].sum()]])                      # first non-singleton dimension
    assert np.array_equal(ev("2^-1"), [[0.5]]) and np.array_equal(ev("-2^2"), [[-4.0]]) and np.array_equal(ev("1./A", A=A), 1.0 / A)
    assert np.array_equal(ev("[1 -2]"), [[1.0, -2.0]]) and np.array_equal(ev("[1 - 2]"), [[-1.0]]) and np.array_equal(ev("[A(1,:); A(2,:)]", A=A), A[:2])
    assert np.array_equal(ev("A'", A=A), A.T) and np.array_equal(ev("A(:,[1 3])'", A=A), A[:, [0, 2]].T)
    assert np.array_equal(ev("A(logical([1 0 1]),2)", A=A), A[[0, 2], 1:2])
    assert np.array_equal(ev("1:3"), [[1.0, 2.0, 3.0]]) and ev("5:1").shape == (1, 0) and np.array_equal(ev("0:0.5:1"), [[0.0, 0.5, 1.0]])
    assert np.allclose(ev("A/B", A=np.eye(2), B=np.array([[2.0, 1.0], [0.0, 4.0]])), np.linalg.inv([[2.0, 1.0], [0.0, 4.0]]))
    assert np.array_equal(ev("reshape(1:6,2,3)"), np.arange(1.0, 7.0).reshape(2, 3, order="F"))
    assert np.array_equal(ev("repmat([1 2],2,1)"), [[1.0, 2.0], [1.0, 2.0]]) and np.array_equal(ev("find([0 1 1],1)"), [[2.0]])
    assert ev("method(2)=='C'", method="VC")[0, 0] and not ev("method(2)=='C'", method="VD")[0, 0]
    assert np.array_equal(ev("eps(1)"), [[2.0 ** -52]]) and np.array_equal(ev("size(zeros(3,0),1)"), [[3.0]])
    assert np.array_equal(ev("bsxfun(@minus,[1;2],[1 2])"), [[0.0, -1.0], [1.0, 0.0]])
    with pytest.raises(ML.MError):
        ev("[1 2 3]+[1 2]")                                                            # no implicit expansion in the reference's MATLAB


# ---- GPU: the HIP path against the executed reference ---------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", GPZ)
def test_hip_path_against_the_executed_reference_gpz(name):
    import gpz_amd
    g = load(name)
    model, theta, X, Y, Psi, om, tr, va = gpz_inputs(g)
    ctx = gpz_amd.GPzContext(model, X, Y, Psi, om, tr, va)
    try:
        f, grad = ctx.eval(theta)
        cond = np.linalg.cond(np.linalg.inv(g["iSigma_w"][:, :, 0]))
        cg = cov_cond(model, theta)
        tol = max(grad_tol(cond), 200.0 * cg * 2.2e-16)
        if model.method[1] == "C" and Psi is not None:
            tol = max(tol, 10.0 * cg ** 1.5 * 2.2e-16)                  # the reference's dGamma chain through inv(Gamma'Gamma), DESIGN.md section 4
        assert abs(f - float(g["nlogML"])) <= max(1e-8, 200.0 * cg * 2.2e-16) * abs(float(g["nlogML"]))
        assert rel(grad, g["grad"]) <= tol, (rel(grad, g["grad"]), tol)
        for key in ("trainRMSE", "trainLL", "validRMSE", "validLL"):
            assert abs(ctx.stats[key] - float(g[key])) <= max(1e-10, 200.0 * cg * 2.2e-16) * max(1.0, abs(float(g[key]))), key
        w, iS, part = ctx.solve(theta)
        assert rel(w, g["w"]) <= tol and rel(iS, g["iSigma_w"]) <= tol and rel(ctx.phi(), g["PHI"]) <= max(1e-12, 200.0 * cg * 2.2e-16)
        assert rel(part, g["nlogML_solve"]) <= max(1e-8, 200.0 * cg * 2.2e-16)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", PRED)
def test_hip_path_against_the_executed_reference_predict(name):
    import gpz_amd
    g = load(name)
    model, Xs, Psi = predict_inputs(g)
    out = gpz_amd.predict(Xs, model, Psi=Psi)
    tol = max(1e-8, 2000.0 * cov_cond(model, g["theta"]) * 2.2e-16)
    for key, val in zip(("mu", "sigma", "nu", "beta_i", "gamma", "PHIs"), out):
        assert rel(val, g[key]) <= tol, (key, rel(val, g[key]))


@pytest.mark.gpu
def test_hip_path_against_the_executed_reference_misc():
    import gpz_amd
    g = load("ref_misc")
    assert rel(gpz_amd.Dxy(g["dxy_X"], g["dxy_Y"]), g["dxy_D"]) <= 1e-13
    Xi, ld = gpz_amd.inv_logdet(g["il_A"])
    assert rel(Xi, g["il_Xi"]) <= 1e-10 and abs(ld - float(g["il_logdet"])) <= 1e-11 * abs(float(g["il_logdet"]))
    Xi, ld = gpz_amd.inv_logdet(g["il2_A"])
    assert rel(Xi, g["il2_Xi"]) <= 1e-8 and abs(ld - float(g["il2_logdet"])) <= 1e-9 * abs(float(g["il2_logdet"]))
    from gpz_amd import host as H
    for key, args in (("om_balanced", ("balanced",)), ("om_balanced_w", ("balanced", 0.05)), ("om_normalized", ("normalized",))):
        assert rel(H.getOmega(g["om_Y"], *args), g[key]) <= 1e-13, key                 # getOmega.m:16 goes through the device Dxy
    for tag, method in (("vd", "VD"), ("vc", "VC")):
        model = gpz_amd.Model(m=4, d=3, k=1, method=method, heteroscedastic=True)
        sel = g[tag + "_sel"].astype(bool)
        PHI, Gam, lnb, N = gpz_amd.getPHI(g[tag + "_X"], g[tag + "_Psi"], g[tag + "_theta"], model, sel, want_N=True)
        tol = max(1e-11, 2000.0 * cov_cond(O.Model(m=4, d=3, k=1, method=method, heteroscedastic=True), g[tag + "_theta"]) * 2.2e-16)
        assert rel(PHI, g[tag + "_PHI"]) <= tol and rel(lnb, g[tag + "_lnBeta_i"]) <= tol and rel(N, g[tag + "_N"]) <= tol
        assert rel(gpz_amd.getPrior(g[tag + "_X"], g[tag + "_Psi"], g[tag + "_theta"], model, sel), g[tag + "_prior"]) <= 1e-8
