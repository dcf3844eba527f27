"""Generate the optimiser fixtures under tests/golden/ (mf_*.npz) from oracle/minfunc_oracle.py.

    python tests/golden/make_golden_minfunc.py

These vectors are outputs of the restatement, frozen at generation time (they carry per-trial traces and phases the reference does
not return).  What the reference's own minFunc files return on the same inputs is in ref_minfunc.npz (oracle/run_reference.py executes
them; tests/test_reference_run.py compares the two):

    mf_mem_*    a sequence of lbfgsAdd(g - g_old, t*d, ...) calls through a wrapping ring with rejected pairs, and the
                lbfgsProd direction after every call (inputs: G, D, T; outputs: added flags, ring state, directions)
    mf_ls_*     WolfeLineSearch / ArmijoBacktrack on analytic objectives: every trial step and function value, the
                accepted step, the number of evaluations
    mf_run_*    whole minFunc('lbfgs') runs: step length, function value and evaluation count per iteration, final x

The objectives are identified by name; tests/minfunc_objectives.py holds their definitions.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import minfunc_oracle as M  # noqa: E402
from oracle import gpz_oracle as O  # noqa: E402
import minfunc_objectives as F  # noqa: E402


def mem_case(name, p, corr, steps, seed):
    rng = np.random.default_rng(seed)
    S = np.zeros((p, corr)); Y = np.zeros((p, corr)); YS = np.zeros(corr)
    start, end, hd = 1, 0, 1.0
    G = np.zeros((steps + 1, p)); D = np.zeros((steps, p)); T = np.zeros(steps)
    added = np.zeros(steps, dtype=np.int32); dirs = np.zeros((steps, p))
    starts = np.zeros(steps, dtype=np.int32); ends = np.zeros(steps, dtype=np.int32); hds = np.zeros(steps)
    G[0] = rng.standard_normal(p)
    for it in range(steps):
        d = rng.standard_normal(p)
        t = float(rng.random() + 0.1)
        g = G[it] + (0.3 * t) * d + 0.05 * rng.standard_normal(p)
        if it % 5 == 3:
            g = G[it] - 0.2 * t * d                                  # y's < 0: lbfgsAdd.m:5 rejects the pair
        start, end, hd, skipped = M.lbfgsAdd(g - G[it], t * d, S, Y, YS, start, end, hd)
        G[it + 1], D[it], T[it] = g, d, t
        added[it] = 0 if skipped else 1
        starts[it], ends[it], hds[it] = start, end, hd
        dirs[it] = M.lbfgsProd(g, S, Y, YS, start, end, hd) if end > 0 else -g
    np.savez(os.path.join(HERE, name), p=p, corrections=corr, G=G, D=D, T=T, added=added, lbfgs_start=starts,
             lbfgs_end=ends, Hdiag=hds, directions=dirs, YS=YS)


def ls_case(name, obj, x, d, t0, kind="wolfe", ls_interp=2, c1=1e-4, c2=0.9):
    fun = F.OBJECTIVES[obj]
    f, g = fun(x)
    gtd = float(g @ d)
    trace = []
    if kind == "wolfe":
        t, fn, gn, ev = M.WolfeLineSearch(x, t0, d, f, g, gtd, c1, c2, ls_interp, 0, 25, 1e-9, fun, trace)
    else:
        t, _, fn, gn, ev = M.ArmijoBacktrack(x, t0, d, f, f, g, gtd, c1, ls_interp, 0, 1e-9, fun, trace)
    np.savez(os.path.join(HERE, name), objective=obj, kind=kind, x=x, d=d, t0=t0, ls_interp=ls_interp, c1=c1, c2=c2,
             trial_t=np.array([r[1] for r in trace]), trial_f=np.array([r[2] for r in trace]),
             phase=np.array([r[0] for r in trace]), t=t, f_new=fn, g_new=gn, funEvals=ev)


def run_case(name, obj, x0, max_iter, extra=None, **opts):
    fun = F.OBJECTIVES[obj] if isinstance(obj, str) else obj
    x, f, flag, out = M.minFunc(fun, x0, maxIter=max_iter, **opts)
    tr = out["trace"]
    data = dict(objective=obj if isinstance(obj, str) else "gpz", x0=x0, max_iter=max_iter, x=x, f=f, exitflag=flag,
                iterations=out["iterations"], funcCount=out["funcCount"], message=out["message"],
                fval=np.array(tr["fval"]), funcCounts=np.array(tr["funcCount"]), optCond=np.array(tr["optCond"]),
                steps=np.array(tr["t"]), X_iter=np.array(tr["x"]))
    data.update(extra or {})
    np.savez(os.path.join(HERE, name), **data)


def main():
    mem_case("mf_mem_p64_c3", 64, 3, 10, 11)
    mem_case("mf_mem_p513_c7", 513, 7, 25, 12)
    mem_case("mf_mem_p900_c100", 900, 100, 16, 13)

    rng = np.random.default_rng(21)
    x = np.full(6, -1.2)
    f, g = F.rosenbrock(x)
    ls_case("mf_ls_rosen_sd", "rosenbrock", x, -g, min(1.0, 1.0 / np.sum(np.abs(g))))
    ls_case("mf_ls_rosen_long", "rosenbrock", x, -g / np.linalg.norm(g), 3.0)          # overshoots: bracket + zoom
    x = rng.standard_normal(8)
    f, g = F.quadratic8(x)
    ls_case("mf_ls_quad_unit", "quadratic8", x, -g, 1.0)
    ls_case("mf_ls_quad_tiny", "quadratic8", x, -g, 1e-4)                             # extrapolation phase
    ls_case("mf_ls_quad_bisect", "quadratic8", x, -g, 2.0, ls_interp=0)
    ls_case("mf_ls_quad_tiny_c2", "quadratic8", x, -g, 1e-6, c2=0.1)                  # several extrapolations, then zoom
    ls_case("mf_ls_quad_unit_c2", "quadratic8", x, -g, 1.0, c2=0.01)
    ls_case("mf_ls_quad_extend_i1", "quadratic8", x, -g, 1e-6, ls_interp=1, c2=0.1)   # t*10 extension + bisection
    xr = np.full(6, -1.2)
    fr_, gr_ = F.rosenbrock(xr)
    ls_case("mf_ls_rosen_c2", "rosenbrock", xr, -gr_ / np.linalg.norm(gr_), 1e-3, c2=0.1)
    ls_case("mf_ls_rosen_far_c2", "rosenbrock", xr, -gr_ / np.linalg.norm(gr_), 1.5, c2=0.1)
    x = np.array([-1.9])
    f, g = F.nan_wall(x)
    ls_case("mf_ls_nanwall", "nan_wall", x, -g, 1.0)                                  # illegal region -> Armijo
    x = rng.standard_normal(8)
    f, g = F.quadratic8(x)
    ls_case("mf_ls_armijo_quad", "quadratic8", x, -g, 5.0, kind="armijo")
    ls_case("mf_ls_armijo_quad_i1", "quadratic8", x, -g, 5.0, kind="armijo", ls_interp=1)

    run_case("mf_run_rosen10", "rosenbrock", np.full(10, -1.2), 500)
    run_case("mf_run_quad8", "quadratic8", np.random.default_rng(22).standard_normal(8), 100)
    run_case("mf_run_nanwall", "nan_wall", np.array([-1.9]), 50)
    run_case("mf_run_rosen10_c5", "rosenbrock", np.full(10, -1.2), 60, corrections=5)  # the ring wraps

    # minFunc on the oracle's GPz objective: the trajectory train() has to reproduce (train.m:42-48)
    n, d, m = 300, 2, 8
    r = np.random.default_rng(6)
    X = r.standard_normal((n, d))
    A = r.standard_normal((d, 1)) / np.sqrt(d)
    Y = np.sin(X @ A) + 0.1 * r.standard_normal((n, 1))
    Y -= Y.mean(0)
    for method in ("VD", "VC"):
        model, theta = O.init_theta(X, Y, method, m, True, r)
        theta = theta + 0.05 * r.standard_normal(theta.size)
        fun = lambda th: (lambda q: (q.nlogML, q.grad))(O.GPz(th, model, X, Y))
        run_case(f"mf_run_gpz_{method}", fun, theta, 12,
                 extra=dict(X=X, Y=Y, method=method, m=m, d=d, n=n))


if __name__ == "__main__":
    main()
