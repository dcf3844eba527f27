import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, gpz_amd
from helpers import make_problem, rel
from test_gpu_parity import _well_conditioned_gamma
from oracle import gpz_oracle as O
for (n,d,m,k) in [(400,10,10,1),(350,12,9,1),(350,16,9,1),(260,20,24,1)]:
    model, theta, X, Y, Psi, rng = make_problem(n, d, m, k, "VC", True, seed=101 + d, psi=True)
    theta = _well_conditioned_gamma(model, theta, rng)
    ref = O.GPz(theta, model, X, Y, Psi)
    c64 = gpz_amd.GPzContext(model, X, Y, Psi); f64, g64 = c64.eval(theta); c64.close()
    c32 = gpz_amd.GPzContext(model, X, Y, Psi, dtype="f32"); f32, g32 = c32.eval(theta); P32 = c32.phi(); c32.close()
    c64 = gpz_amd.GPzContext(model, X, Y, Psi); c64.eval(theta); P64 = c64.phi(); c64.close()
    md = m*d; gd = model.g_dim
    mx = np.abs(ref.grad).max()
    print(f"d={d}: f rel {abs(f32-ref.nlogML)/abs(ref.nlogML):.2e} (f64 path {abs(f64-ref.nlogML)/abs(ref.nlogML):.2e}); PHI rel {rel(P32,P64):.2e};"
          f" g err/max: dP {np.abs(g32[:md]-ref.grad[:md]).max()/mx:.2e} dG {np.abs(g32[md:md+gd]-ref.grad[md:md+gd]).max()/mx:.2e} rest {np.abs(g32[md+gd:]-ref.grad[md+gd:]).max()/mx:.2e}"
          f" | block maxima dP {np.abs(ref.grad[:md]).max():.2e} dG {np.abs(ref.grad[md:md+gd]).max():.2e} rest {np.abs(ref.grad[md+gd:]).max():.2e}")
