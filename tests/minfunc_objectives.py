"""Analytic objectives of the optimiser fixtures (tests/golden/mf_*.npz): name -> fun(x) -> (f, g)."""
import numpy as np


def rosenbrock(x):
    f = np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
    g = np.zeros_like(x)
    g[:-1] = -400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
    g[1:] += 200.0 * (x[1:] - x[:-1] ** 2)
    return float(f), g


_rng = np.random.default_rng(20)
_Q = _rng.standard_normal((8, 8))
_A8 = _Q @ np.diag(10.0 ** np.linspace(0, 3, 8)) @ _Q.T / 8.0
_A8 = 0.5 * (_A8 + _A8.T) + np.eye(8)
_b8 = _rng.standard_normal(8)


def quadratic8(x):
    """Ill-conditioned convex quadratic (cond ~ 1e3)."""
    return float(0.5 * x @ _A8 @ x - _b8 @ x), _A8 @ x - _b8


def nan_wall(x):
    """(x - 1.5)^2 with a non-finite region |x| > 2: the line search must back off (WolfeLineSearch.m:53-70)."""
    if abs(x[0]) > 2.0:
        return float("nan"), np.full_like(x, np.nan)
    return float((x[0] - 1.5) ** 2), np.array([2 * (x[0] - 1.5)])


OBJECTIVES = {"rosenbrock": rosenbrock, "quadratic8": quadratic8, "nan_wall": nan_wall}
