"""Host-side callers of the path (gpz_amd/host.py): the L-BFGS / strong-Wolfe driver mirrors minFunc's 'lbfgs'
method (minFunc.m:544-582,968-1150; WolfeLineSearch.m; polyinterp.m; lbfgsAdd.m; lbfgsProd.m).  CPU tests use
analytic objectives and the oracle; the GPU tests run init -> train -> predict through the HIP path."""
import math

import numpy as np
import pytest

from gpz_amd import host
from oracle import gpz_oracle as O
from helpers import make_problem, rel


def rosenbrock(x):
    f = np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
    g = np.zeros_like(x)
    g[:-1] = -400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
    g[1:] += 200.0 * (x[1:] - x[:-1] ** 2)
    return float(f), g


def test_polyinterp_cubic_closed_form():
    # f(x) = x^3 - 3x  -> minimum at x = 1; points at 0 and 2
    t = host._polyinterp([(0.0, 0.0, -3.0), (2.0, 2.0, 9.0)])
    assert abs(t - 1.0) < 1e-14
    # quadratic from one value+slope and one value: f = (x-0.3)^2
    t = host._polyinterp([(0.0, 0.09, -0.6), (1.0, 0.49, None)], 0.0, 1.0)
    assert abs(t - 0.3) < 1e-12


def test_lbfgs_converges_on_rosenbrock():
    x0 = np.full(10, -1.2)
    x, f, flag, evals, msg = host.minfunc_lbfgs(rosenbrock, x0, max_iter=500)
    assert flag in (1, 2) and f < 1e-8 and np.max(np.abs(x - 1.0)) < 1e-3   # stops on progTol/optTol like minFunc


def test_wolfe_conditions_hold():
    rng = np.random.default_rng(0)
    A = rng.standard_normal((8, 8)); A = A @ A.T + np.eye(8)
    b = rng.standard_normal(8)
    fun = lambda x: (0.5 * x @ A @ x - b @ x, A @ x - b)
    x = rng.standard_normal(8)
    f, g = fun(x)
    d = -g
    gtd = g @ d
    t, fn, gn, ev = host._wolfe(fun, x, 1.0, d, f, g, gtd, 1e-4, 0.9, 25, 1e-9)
    assert fn <= f + 1e-4 * t * gtd and abs(gn @ d) <= -0.9 * gtd


def test_lbfgs_memory_is_circular_and_skips_bad_pairs():
    mem = host._LBFGS(3, 2)
    assert not mem.add(np.array([1.0, 0, 0]), np.array([-1.0, 0, 0]))         # y's <= 1e-10: skipped (lbfgsAdd.m:5)
    for q in range(3):
        assert mem.add(np.eye(3)[q] * (q + 1.0), np.eye(3)[q])
    assert mem.count == 2 and mem.hdiag == pytest.approx(1.0 / 3.0)
    d = mem.direction(np.ones(3))
    assert np.all(np.isfinite(d)) and d @ np.ones(3) < 0                      # a descent direction


def test_nonfinite_objective_backs_off():
    def fun(x):
        if abs(x[0]) > 2.0:
            return float("nan"), np.full_like(x, np.nan)
        return float((x[0] - 1.5) ** 2), np.array([2 * (x[0] - 1.5)])
    x, f, flag, evals, msg = host.minfunc_lbfgs(fun, np.array([-1.9]), max_iter=50)
    assert abs(x[0] - 1.5) < 1e-5


def test_lbfgs_on_oracle_objective_decreases():
    model, theta, X, Y, _, rng = make_problem(120, 2, 6, 1, "VD", True, seed=4)
    fun = lambda t: (lambda r: (r.nlogML, r.grad))(O.GPz(t, model, X, Y))
    f0, g0 = fun(theta)
    x, f, flag, evals, msg = host.minfunc_lbfgs(fun, theta, max_iter=15)
    assert f < f0 - 1e-3


def test_input_helpers_match_oracle():
    sd = np.array([2.0, 4.0])
    for psi in (np.array([1.0, 2.0, 3.0]), np.abs(np.random.default_rng(0).standard_normal((3, 2)))):
        for method in ("VD", "VC"):
            assert rel(host.fixPsi(psi, 3, sd, method), O.fixPsi(psi, 3, sd, method)) < 1e-15
    tr, va, te = host.sample(100, 0.7, 0.15, 0.15, np.random.default_rng(1))
    assert tr.sum() == 70 and va.sum() == 15 and te.sum() == 15 and not (tr & va).any() and (tr | va | te).all()


# ---- through the HIP path -------------------------------------------------------------------------
def _sinc_data(n=3000, seed=1):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-10, 10, (n, 1))
    noise = 0.05 + 0.2 * (1 + np.sin(X[:, 0] / 3)) / 2
    Y = (np.sinc(X[:, 0] / np.pi) + noise * rng.standard_normal(n))[:, None]
    return X, Y


@pytest.mark.gpu
def test_init_train_predict_sinc():
    import gpz_amd
    X, Y = _sinc_data()
    rng = np.random.default_rng(0)
    tr, va, te = gpz_amd.sample(X.shape[0], 0.7, 0.15, 0.15, rng)
    model = gpz_amd.init(X, Y, "VL", 30, training=tr, rng=rng)
    assert model.method == "VL" and model.sets["last"]["w"].shape == (30, 1)
    model = gpz_amd.train(model, X, Y, maxIter=150, maxAttempts=50, training=tr, validation=va, verbose=False)
    mu, sigma, nu, beta_i, gamma, PHI, w, iS = gpz_amd.predict(X, model, selection=te)
    rmse = math.sqrt(np.mean((mu[:, 0] - Y[te, 0]) ** 2))
    assert rmse < 0.2 and np.all(sigma > 0)
    # heteroscedastic noise model picked up the input-dependent noise: predicted variance correlates with it
    noise = 0.05 + 0.2 * (1 + np.sin(X[te, 0] / 3)) / 2
    assert np.corrcoef(np.sqrt(sigma[:, 0]), noise)[0, 1] > 0.5


@pytest.mark.gpu
def test_training_trajectory_matches_oracle_objective():
    """Same optimiser, objective from the HIP path vs the oracle: the first iterations must coincide."""
    import gpz_amd
    model, theta, X, Y, _, rng = make_problem(400, 2, 8, 1, "VC", True, seed=6)
    ctx = gpz_amd.GPzContext(model, X, Y)
    fs_gpu, fs_cpu = [], []
    rec = lambda store: (lambda x, kind, i, ev, f, *a: store.append(f) or False)
    host.minfunc_lbfgs(ctx.eval, theta, max_iter=6, output_fcn=rec(fs_gpu))
    ctx.close()
    host.minfunc_lbfgs(lambda t: (lambda r: (r.nlogML, r.grad))(O.GPz(t, model, X, Y)), theta, max_iter=6,
                       output_fcn=rec(fs_cpu))
    assert len(fs_gpu) == len(fs_cpu) and rel(np.array(fs_gpu), np.array(fs_cpu)) < 1e-7


# ---- device-resident optimiser vectors and L-BFGS memory (gpz_eval_dev, gpz_lbfgs_*, k_lbfgs.hip) ----
@pytest.mark.gpu
@pytest.mark.parametrize("p,corr,steps", [(1000, 5, 12), (113001, 7, 10), (4097, 100, 30)])
def test_device_lbfgs_direction_matches_two_loop(p, corr, steps):
    """lbfgsAdd.m / lbfgsProd.m on the device (Gram-matrix form of the two-loop recursion) against the host two-loop,
    through a wrapping ring and with rejected (y's <= 1e-10) pairs in between."""
    rng = np.random.default_rng(p)
    hostmem = host._LBFGS(p, corr)
    devmem = host._LBFGSDevice(p, corr)
    g_old = rng.standard_normal(p)
    for it in range(steps):
        d = rng.standard_normal(p)
        t = float(rng.random() + 0.1)
        g = g_old + (0.3 * t) * d + 0.05 * rng.standard_normal(p)        # y's > 0 mostly
        if it % 5 == 3:
            g = g_old - 0.2 * t * d                                       # y's < 0: the pair must be skipped
        a_h = hostmem.add_step(g, g_old, t, d)
        a_d = devmem.add_step(host.DevVec.from_host(g), host.DevVec.from_host(g_old), t, host.DevVec.from_host(d))
        assert a_h == a_d
        dh = hostmem.direction(g)
        dd = devmem.direction(host.DevVec.from_host(g)).host()
        assert rel(dd, dh) < 1e-10, it
        g_old = g
    devmem.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mf_mem_p64_c3", "mf_mem_p513_c7", "mf_mem_p900_c100",
                                  "ref_lbfgs_mem"])   # the last one: lbfgsAdd.m / lbfgsProd.m themselves, executed (oracle/run_reference.py)
def test_device_lbfgs_memory_matches_the_restated_reference(name):
    """gpz_lbfgs_add / gpz_lbfgs_direction (k_lbfgs.hip) replay the fixtures of oracle/minfunc_oracle.py — lbfgsAdd.m's ring
    (wrapping, rejected pairs) and lbfgsProd.m / mex/lbfgsProdC.c:46-88's product — not the package's own host two-loop."""
    import os
    from helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    dev = host._LBFGSDevice(int(z["p"]), int(z["corrections"]))
    n_added = 0
    for it in range(z["T"].size):
        g, g_old = z["G"][it + 1], z["G"][it]
        added = dev.add_step(host.DevVec.from_host(g), host.DevVec.from_host(g_old), float(z["T"][it]),
                             host.DevVec.from_host(z["D"][it]))
        assert added == bool(z["added"][it]), it
        n_added += int(added)
        if n_added:
            dd = dev.direction(host.DevVec.from_host(g)).host()
            assert rel(dd, z["directions"][it]) < 1e-10, it
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mf_run_gpz_VD", "mf_run_gpz_VC"])
@pytest.mark.parametrize("device_resident", [False, True])
def test_minfunc_trajectory_through_the_hip_path_matches_the_restated_reference(name, device_resident):
    """minFunc('lbfgs') as restated in oracle/minfunc_oracle.py on the oracle's objective (fixture) against the package's
    driver on the HIP objective: same step lengths, function values and evaluation counts per iteration."""
    import os
    import gpz_amd
    from helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = gpz_amd.Model(m=int(z["m"]), d=int(z["d"]), k=1, method=str(z["method"]), heteroscedastic=True)
    ctx = gpz_amd.GPzContext(model, z["X"], z["Y"])
    its = []

    def out(x, kind, i, evals, f, t, gtd, g, d, opt):
        if kind == "iter":
            its.append((t, f, evals))
        return False
    if device_resident:
        fun = lambda th: (lambda r: (r[0], host.DevVec(r[1])))(ctx.eval_dev(th.t))
        x0 = host.DevVec.from_host(z["x0"])
    else:
        fun, x0 = ctx.eval, z["x0"]
    x, f, flag, evals, msg = host.minfunc_lbfgs(fun, x0, max_iter=int(z["max_iter"]), output_fcn=out)
    ctx.close()
    if device_resident:
        x = x.host()
    assert evals == int(z["funcCount"]) and [e for _, _, e in its] == list(z["funcCounts"][1:])
    assert np.allclose([t for t, _, _ in its], z["steps"], rtol=1e-5)
    assert np.allclose([v for _, v, _ in its], z["fval"][1:], rtol=1e-7, atol=1e-10)
    assert rel(x, z["x"]) < 1e-4


@pytest.mark.gpu
def test_devvec_reductions():
    rng = np.random.default_rng(3)
    a = rng.standard_normal(50001); b = rng.standard_normal(50001)
    A, B = host.DevVec.from_host(a), host.DevVec.from_host(b)
    assert abs((A @ B) - a @ b) < 1e-9 and abs(A.amax() - np.abs(a).max()) == 0 and abs(A.asum() - np.abs(a).sum()) < 1e-8
    assert rel((A + 0.25 * B).host(), a + 0.25 * b) < 1e-15 and A.legal()
    a[77] = np.nan
    assert not host.DevVec.from_host(a).legal()


@pytest.mark.gpu
def test_device_resident_training_matches_host_training():
    """train(..., device_resident=True): theta, g, d and the L-BFGS memory stay on the GPU; same optimiser, same
    objective, so the iterates agree with the host-vector run to rounding."""
    import gpz_amd
    X, Y = _sinc_data()
    rng = np.random.default_rng(0)
    tr, va, te = gpz_amd.sample(X.shape[0], 0.7, 0.15, 0.15, rng)
    m0 = gpz_amd.init(X, Y, "VL", 20, training=tr, rng=np.random.default_rng(5))
    import copy
    # the two memories sum in different orders (two-loop vs Gram-matrix form), so the iterates drift apart at the
    # rounding level and the drift grows with the iteration count: compare a short run tightly, a long one by outcome
    ma = gpz_amd.train(copy.deepcopy(m0), X, Y, maxIter=12, maxAttempts=50, training=tr, validation=va, verbose=False,
                       device_resident=False)
    mb = gpz_amd.train(copy.deepcopy(m0), X, Y, maxIter=12, maxAttempts=50, training=tr, validation=va, verbose=False,
                       device_resident=True)
    assert ma.train_info["funEvals"] == mb.train_info["funEvals"]
    assert abs(ma.train_info["f"] - mb.train_info["f"]) <= 1e-7 * abs(ma.train_info["f"])
    assert rel(mb.sets["last"]["theta"], ma.sets["last"]["theta"]) < 1e-5
    ma = gpz_amd.train(copy.deepcopy(m0), X, Y, maxIter=150, maxAttempts=50, training=tr, validation=va, verbose=False,
                       device_resident=False)
    mb = gpz_amd.train(copy.deepcopy(m0), X, Y, maxIter=150, maxAttempts=50, training=tr, validation=va, verbose=False,
                       device_resident=True)
    assert abs(ma.train_info["f"] - mb.train_info["f"]) <= 2e-2 * abs(ma.train_info["f"])
    mu_a = gpz_amd.predict(X, ma, selection=te)[0]; mu_b = gpz_amd.predict(X, mb, selection=te)[0]
    ra = math.sqrt(np.mean((mu_a[:, 0] - Y[te, 0]) ** 2)); rb = math.sqrt(np.mean((mu_b[:, 0] - Y[te, 0]) ** 2))
    assert rb < 0.2 and abs(ra - rb) < 0.02


def test_sample_and_metrics_follow_the_reference_semantics():
    """sample.m:3-17: fractions are NOT renormalised and counts are accepted (demo_photoz.m:49); metrics.m: fun(y,mu,sigma)
    element-wise on the variance-sorted vectors, running mean."""
    tr, va, te = host.sample(1000, 0.2, 0.2, 0.2, np.random.default_rng(0))
    assert (tr.sum(), va.sum(), te.sum()) == (200, 200, 200) and not (tr & va).any() and not (tr & te).any() and not (va & te).any()
    tr, va, te = host.sample(1000, 0.9, 0.2, 0.2, np.random.default_rng(0))          # training capped by what is left
    assert (tr.sum(), va.sum(), te.sum()) == (600, 200, 200)
    tr, va, te = host.sample(50000, 10000, 10000, 10000, np.random.default_rng(0))   # row counts
    assert (tr.sum(), va.sum(), te.sum()) == (10000, 10000, 10000)
    rng = np.random.default_rng(1)
    y, mu, sg = rng.standard_normal(200), rng.standard_normal(200), rng.random(200)
    sc = host.metrics(y, mu, sg, lambda y_, mu_, s_: (y_ - mu_) ** 2 / (1 + s_))    # a three-argument lambda like demo_photoz.m:86-88
    o = np.argsort(sg, kind="stable")
    want = np.cumsum((y[o] - mu[o]) ** 2 / (1 + sg[o])) / np.arange(1, 201)
    assert sc.shape == (200,) and np.allclose(sc, want, rtol=1e-14, atol=0)


@pytest.mark.gpu
def test_train_over_three_loopback_shards_reproduces_the_single_context_run():
    """train(..., n_gpus=3, reducer="loopback"): init -> L-BFGS -> re-solve with the closure evaluated by gpz_mgpu_* (rows
    sharded by the library, partials reduced at the two exchange points).  The same run on one context must take the same
    steps: identical evaluation count, objective and parameters to rounding of the regrouped sums."""
    import gpz_amd
    X, Y = _sinc_data(1500)
    rng = np.random.default_rng(3)
    tr, va, te = gpz_amd.sample(X.shape[0], 0.7, 0.15, 0.15, rng)
    base = gpz_amd.init(X, Y, "VL", 12, training=tr, rng=np.random.default_rng(4))
    import copy
    m1 = gpz_amd.train(copy.deepcopy(base), X, Y, maxIter=25, training=tr, validation=va, verbose=False, device_resident=False)
    m3 = gpz_amd.train(copy.deepcopy(base), X, Y, maxIter=25, training=tr, validation=va, verbose=False, n_gpus=3,
                       reducer="loopback")
    assert m1.train_info["funEvals"] == m3.train_info["funEvals"]
    assert abs(m1.train_info["f"] - m3.train_info["f"]) <= 1e-9 * abs(m1.train_info["f"])
    for name in ("last", "best"):
        assert rel(m3.sets[name]["theta"], m1.sets[name]["theta"]) <= 1e-7
        assert rel(m3.sets[name]["w"], m1.sets[name]["w"]) <= 1e-6


def test_bench_refuses_rank_records_that_do_not_describe_n_devices():
    """bench.py --gpus N gathers what every rank's RCCL communicator reports about itself (gpz_*_comm_info) and must fail unless that
    is N ranks, 0 .. N-1 once each, on N different PCI devices (the first real multi-GPU run has to prove itself)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    good = [{"nccl_count": 4, "nccl_rank": r, "pci_bus_id": "0000:%02x:00.0" % (5 + r)} for r in range(4)]
    assert bench.ranks_describe_n_devices(good, 4) is None
    assert "ncclCommCount" in bench.ranks_describe_n_devices([dict(q, nccl_count=2) for q in good], 4)           # a smaller communicator
    assert bench.ranks_describe_n_devices([dict(q, nccl_rank=0) for q in good], 4)                                 # a rank twice
    assert bench.ranks_describe_n_devices([dict(q, pci_bus_id=good[0]["pci_bus_id"]) for q in good], 4)            # ranks sharing a device
    assert bench.ranks_describe_n_devices(good[:3], 4)                                                             # a rank missing
    assert bench.ranks_describe_n_devices([dict(q, nccl_count=-1, nccl_rank=-1) for q in good], 4)                 # no in-library communicator
