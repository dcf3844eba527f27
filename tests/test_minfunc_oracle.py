"""The optimiser restatement (oracle/minfunc_oracle.py: lbfgsAdd.m, lbfgsProd.m / mex/lbfgsProdC.c, polyinterp.m,
WolfeLineSearch.m, ArmijoBacktrack.m, minFunc.m's LBFGS branch) pinned by checks that share no code with it, the
committed fixtures (tests/golden/mf_*.npz) as regression vectors, and the product's host driver (gpz_amd/host.py)
compared with both.  CPU only; the device-resident memory (k_lbfgs.hip) replays the same fixtures in test_host.py."""
import glob
import os

import numpy as np
import pytest

from gpz_amd import host
from oracle import minfunc_oracle as M
from oracle import gpz_oracle as O
import minfunc_objectives as F
from helpers import GOLDEN, rel


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def names(prefix):
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


# ---- pins of the restatement that do not share code with it ----------------------------------------------------
def dense_bfgs_direction(pairs, hdiag, g):
    """-H g with H the BFGS inverse-Hessian built densely from the (s, y) pairs, oldest first, starting at Hdiag*I
    (Nocedal & Wright eq. 6.17) — what the two-loop recursion computes without forming H."""
    p = g.size
    H = hdiag * np.eye(p)
    for s, y in pairs:
        rho = 1.0 / (y @ s)
        V = np.eye(p) - rho * np.outer(y, s)
        H = V.T @ H @ V + rho * np.outer(s, s)
    return -H @ g


@pytest.mark.parametrize("p,corr,nadd", [(12, 4, 3), (12, 4, 4), (12, 4, 5), (12, 4, 9), (30, 1, 3), (9, 2, 7), (20, 6, 40)])
def test_two_loop_equals_dense_bfgs_through_the_ring(p, corr, nadd):
    rng = np.random.default_rng(100 * p + nadd)
    S = np.zeros((p, corr)); Y = np.zeros((p, corr)); YS = np.zeros(corr)
    start, end, hd = 1, 0, 1.0
    accepted = []
    A = rng.standard_normal((p, p)); A = A @ A.T / p + np.eye(p)          # y = A s keeps y's > 0
    for q in range(nadd):
        s = rng.standard_normal(p)
        y = A @ s if q % 4 != 2 else -s                                   # every fourth pair is rejected (lbfgsAdd.m:5)
        start, end, hd, skipped = M.lbfgsAdd(y, s, S, Y, YS, start, end, hd)
        assert skipped == (1 if q % 4 == 2 else 0)
        if not skipped:
            accepted.append((s, y))
    kept = accepted[-corr:]
    g = rng.standard_normal(p)
    ref = dense_bfgs_direction(kept, hd, g)
    assert hd == pytest.approx((kept[-1][1] @ kept[-1][0]) / (kept[-1][1] @ kept[-1][1]), rel=1e-15)
    d1 = M.lbfgsProd(g, S, Y, YS, start, end, hd)
    d2 = M.lbfgsProdC(g, S, Y, YS, start, end, hd, literal=True)
    d3 = M.lbfgsProdC(g, S, Y, YS, start, end, hd)
    assert rel(d1, ref) < 1e-11 and rel(d2, ref) < 1e-11 and rel(d2, d1) < 1e-13 and rel(d3, d1) < 1e-13
    # the ring holds exactly the newest `corr` accepted pairs, oldest first in lbfgsProd's index order (lbfgsProd.m:9-15)
    ind = list(range(1, end + 1)) if start == 1 else list(range(start, corr + 1)) + list(range(1, end + 1))
    assert len(ind) == len(kept)
    for col, (s, y) in zip(ind, kept):
        assert np.array_equal(S[:, col - 1], s) and np.array_equal(Y[:, col - 1], y)


def test_polyinterp_closed_form_cubic_equals_general_branch_and_analytic_minimisers():
    rng = np.random.default_rng(3)
    for _ in range(200):
        c = rng.standard_normal(4)                                        # a random cubic
        poly = lambda x: ((c[0] * x + c[1]) * x + c[2]) * x + c[3]
        dpoly = lambda x: (3 * c[0] * x + 2 * c[1]) * x + c[2]
        x0, x1 = np.sort(rng.uniform(-2, 2, 2))
        if x1 - x0 < 1e-2:
            continue
        fast, _ = M.polyinterp([[x0, poly(x0), dpoly(x0)], [x1, poly(x1), dpoly(x1)]])
        xm = 0.5 * (x0 + x1)                                              # a third row with nothing known: general branch (:62-116)
        slow, fmin = M.polyinterp([[x0, poly(x0), dpoly(x0)], [x1, poly(x1), dpoly(x1)], [xm, M.I, M.I]])
        # the general branch minimises over the interval (end points included); the shortcut returns the cubic's local
        # minimiser clipped to the interval, or the midpoint when the cubic has no real critical point
        disc = (2 * c[1]) ** 2 - 12 * c[0] * c[2]
        if disc > 1e-8:
            r = (-2 * c[1] + np.sqrt(disc)) / (6 * c[0])                  # the root with positive curvature
            if x0 + 1e-6 < r < x1 - 1e-6:
                assert abs(fast - r) < 1e-7 * max(1, abs(r))
                if poly(r) < min(poly(x0), poly(x1)) - 1e-9:
                    assert abs(slow - r) < 1e-6 and abs(fmin - poly(r)) < 1e-7
    # quadratic from (f, g) at 0 and f at t:  (x - 0.3)^2
    t, fm = M.polyinterp([[0.0, 0.09, -0.6], [1.0, 0.49, M.I]], 0.0, 1.0)
    assert abs(t - 0.3) < 1e-12 and abs(fm) < 1e-12
    # secant-type fit from two derivatives and one value
    t, _ = M.polyinterp([[0.0, 0.09, -0.6], [1.0, M.I, 1.4]], 0.0, 1.0)
    assert abs(t - 0.3) < 1e-12
    # x^3 - 3x on [0, 2]: minimum at 1
    assert abs(M.polyinterp([[0.0, 0.0, -3.0], [2.0, 2.0, 9.0]])[0] - 1.0) < 1e-14
    # no real critical point -> bisection of the bounds (polyinterp.m:57)
    assert M.polyinterp([[0.0, 0.0, 1.0], [1.0, 0.7, 1.0]], 0.0, 4.0)[0] == 2.0


@pytest.mark.parametrize("name", [n for n in names("mf_ls_") if "armijo" not in n and "nanwall" not in n])
def test_wolfe_fixture_steps_satisfy_the_strong_wolfe_conditions(name):
    z = load(name)
    fun = F.OBJECTIVES[str(z["objective"])]
    x, d, t = z["x"], z["d"], float(z["t"])
    f, g = fun(x)
    fn, gn = fun(x + t * d)
    gtd = g @ d
    assert fn <= f + float(z["c1"]) * t * gtd + 1e-15
    assert abs(gn @ d) <= -float(z["c2"]) * gtd * (1 + 1e-12)
    assert fn == pytest.approx(float(z["f_new"]), rel=1e-14) and int(z["funEvals"]) == z["trial_t"].size


# ---- fixtures are what the restatement produces today (regression guard) --------------------------------------
@pytest.mark.parametrize("name", names("mf_mem_"))
def test_memory_fixtures_regenerate(name):
    z = load(name)
    p, corr = int(z["p"]), int(z["corrections"])
    S = np.zeros((p, corr)); Y = np.zeros((p, corr)); YS = np.zeros(corr)
    start, end, hd = 1, 0, 1.0
    for it in range(z["T"].size):
        g, g_old = z["G"][it + 1], z["G"][it]
        start, end, hd, skipped = M.lbfgsAdd(g - g_old, z["T"][it] * z["D"][it], S, Y, YS, start, end, hd)
        assert (skipped == 0) == bool(z["added"][it]) and start == z["lbfgs_start"][it] and end == z["lbfgs_end"][it]
        assert hd == z["Hdiag"][it]
        d = M.lbfgsProdC(g, S, Y, YS, start, end, hd)
        assert rel(d, z["directions"][it]) < 1e-13


@pytest.mark.parametrize("name", names("mf_ls_"))
def test_line_search_fixtures_regenerate(name):
    z = load(name)
    fun = F.OBJECTIVES[str(z["objective"])]
    x, d = z["x"], z["d"]
    f, g = fun(x)
    tr = []
    if str(z["kind"]) == "wolfe":
        t, fn, gn, ev = M.WolfeLineSearch(x, float(z["t0"]), d, f, g, float(g @ d), float(z["c1"]), float(z["c2"]),
                                          int(z["ls_interp"]), 0, 25, 1e-9, fun, tr)
    else:
        t, _, fn, gn, ev = M.ArmijoBacktrack(x, float(z["t0"]), d, f, f, g, float(g @ d), float(z["c1"]),
                                             int(z["ls_interp"]), 0, 1e-9, fun, tr)
    assert ev == int(z["funEvals"]) and [r[0] for r in tr] == [str(s) for s in z["phase"]]
    assert np.allclose([r[1] for r in tr], z["trial_t"], rtol=1e-12, atol=0) and t == pytest.approx(float(z["t"]), rel=1e-12)


# ---- the product's host driver against the restatement --------------------------------------------------------
@pytest.mark.parametrize("name", names("mf_mem_"))
def test_host_lbfgs_memory_matches_fixture(name):
    z = load(name)
    mem = host._LBFGS(int(z["p"]), int(z["corrections"]))
    for it in range(z["T"].size):
        added = mem.add_step(z["G"][it + 1], z["G"][it], float(z["T"][it]), z["D"][it])
        assert added == bool(z["added"][it])
        assert mem.hdiag == pytest.approx(float(z["Hdiag"][it]), rel=1e-14)
        if mem.count:
            assert rel(mem.direction(z["G"][it + 1]), z["directions"][it]) < 1e-12, it


def test_host_polyinterp_matches_restatement():
    rng = np.random.default_rng(8)
    for q in range(300):
        x0, x1 = rng.uniform(-1, 3, 2)
        if abs(x1 - x0) < 1e-3:
            continue
        f0, f1, g0, g1 = rng.standard_normal(4)
        lo, hi = (None, None) if q % 2 else (min(x0, x1) - rng.random(), max(x0, x1) + 3 * rng.random())
        a = host._polyinterp([(x0, f0, g0), (x1, f1, g1)], lo, hi)
        b = M.polyinterp([[x0, f0, g0], [x1, f1, g1]], lo, hi)[0]
        assert a == pytest.approx(b, rel=1e-12, abs=1e-14)
        t = abs(x1) + 0.1
        a = host._polyinterp([(0.0, f0, -abs(g0)), (t, f1, None)], 0.0, t)
        b = M.polyinterp([[0.0, f0, -abs(g0)], [t, f1, M.I]], 0.0, t)[0]
        assert a == pytest.approx(b, rel=1e-10, abs=1e-13)


def _recording(fun, x, d):
    """Wrap an objective so that every evaluation's step length along d and value are recorded."""
    rec = []

    def wrapped(xx):
        f, g = fun(xx)
        rec.append((float((xx - x) @ d / (d @ d)), f))
        return f, g
    return wrapped, rec


@pytest.mark.parametrize("name", [n for n in names("mf_ls_") if "_i1" not in n and "bisect" not in n])
def test_host_line_searches_take_the_restatement_s_steps(name):
    z = load(name)
    fun = F.OBJECTIVES[str(z["objective"])]
    x, d = z["x"], z["d"]
    f, g = fun(x)
    wrapped, rec = _recording(fun, x, d)
    if str(z["kind"]) == "wolfe":
        t, fn, gn, ev = host._wolfe(wrapped, x, float(z["t0"]), d, f, g, float(g @ d), float(z["c1"]), float(z["c2"]), 25, 1e-9)
    else:
        t, fn, gn, ev = host._armijo(wrapped, x, float(z["t0"]), d, f, f, g, float(g @ d), float(z["c1"]), 1e-9)
    assert ev == int(z["funEvals"]) == len(rec)
    assert np.allclose([r[0] for r in rec], z["trial_t"], rtol=1e-9, atol=1e-15)
    fin = np.isfinite(z["trial_f"])
    assert np.allclose(np.array([r[1] for r in rec])[fin], z["trial_f"][fin], rtol=1e-10, atol=1e-14)
    assert t == pytest.approx(float(z["t"]), rel=1e-9) and fn == pytest.approx(float(z["f_new"]), rel=1e-10, abs=1e-14)


def _gpz_objective(z):
    model = O.Model(m=int(z["m"]), d=int(z["d"]), k=1, method=str(z["method"]), heteroscedastic=True)
    return lambda th: (lambda q: (q.nlogML, q.grad))(O.GPz(th, model, z["X"], z["Y"]))


@pytest.mark.parametrize("name", names("mf_run_"))
def test_host_minfunc_reproduces_the_restatement_s_trajectory(name):
    z = load(name)
    fun = _gpz_objective(z) if str(z["objective"]) == "gpz" else F.OBJECTIVES[str(z["objective"])]
    corr = 5 if name.endswith("_c5") else 100
    it_f, it_t, it_ev = [], [], []

    def out(x, kind, i, evals, f, t, gtd, g, d, opt):
        if kind == "iter":
            it_f.append(f); it_t.append(t); it_ev.append(evals)
        return False
    x, f, flag, evals, msg = host.minfunc_lbfgs(fun, z["x0"], max_iter=int(z["max_iter"]), output_fcn=out, corrections=corr)
    nit = int(z["iterations"])
    # a run that ends on the directional-derivative test leaves the loop before the line search of its last iteration
    assert len(it_f) == z["steps"].size and evals == int(z["funcCount"]) and flag == int(z["exitflag"])
    assert msg.rstrip("!") == str(z["message"]).rstrip("!")
    assert it_ev == list(z["funcCounts"][1:])
    # early iterations agree to rounding; later ones drift with the conditioning of the problem (different summation
    # order in the two-loop product), so the step sequence is gated on a prefix and the outcome on the whole run
    k = min(10, len(it_t))
    assert np.allclose(it_t[:k], z["steps"][:k], rtol=1e-7) and np.allclose(it_f[:k], z["fval"][1:k + 1], rtol=1e-9, atol=1e-12)
    assert f == pytest.approx(float(z["f"]), rel=1e-4, abs=1e-9) and rel(x, z["x"]) < 1e-3
    assert nit >= len(it_t)


# ---- whole-run regression of the restatement itself -----------------------------------------------------------
@pytest.mark.parametrize("name", names("mf_run_"))
def test_minfunc_fixtures_regenerate(name):
    z = load(name)
    fun = _gpz_objective(z) if str(z["objective"]) == "gpz" else F.OBJECTIVES[str(z["objective"])]
    x, f, flag, out = M.minFunc(fun, z["x0"], maxIter=int(z["max_iter"]), corrections=5 if name.endswith("_c5") else 100)
    assert flag == int(z["exitflag"]) and out["funcCount"] == int(z["funcCount"]) and out["iterations"] == int(z["iterations"])
    assert np.allclose(out["trace"]["t"], z["steps"], rtol=1e-9) and rel(x, z["x"]) < 1e-9
    # useMex = 0 (lbfgsProd.m instead of lbfgsProdC.c) is the same optimiser
    x2, f2, flag2, out2 = M.minFunc(fun, z["x0"], maxIter=int(z["max_iter"]), corrections=5 if name.endswith("_c5") else 100,
                                    useMex=0)
    assert flag2 == flag and out2["funcCount"] == out["funcCount"] and rel(x2, x) < 1e-6


def test_minfunc_options_paths():
    """LS_init 1-3, the Armijo line search with a non-monotone reference (Fref), LS_multi and the evaluation cap."""
    x0 = np.full(6, -1.2)
    for ls_init in (1, 2, 3):
        x, f, flag, out = M.minFunc(F.rosenbrock, x0, LS_init=ls_init)
        assert f < 1e-8 and flag in (1, 2)
    x, f, flag, out = M.minFunc(F.rosenbrock, x0, LS_type=0, Fref=5, LS_multi=1, maxIter=2000, maxFunEvals=5000)
    assert f < 1e-6
    x, f, flag, out = M.minFunc(F.rosenbrock, x0, maxFunEvals=10)
    assert flag == 0 and out["message"] == "Reached Maximum Number of Function Evaluations"
    stops = []
    x, f, flag, out = M.minFunc(F.rosenbrock, x0, outputFcn=lambda x, s, i, *a: stops.append(s) or (s == "iter" and i == 3))
    assert flag == -1 and out["iterations"] == 3 and stops == ["init", "iter", "iter", "iter", "done"]
