"""mex/gpz_mex.cpp — the MATLAB side of the drop-in boundary — compiled and EXECUTED without MATLAB.

tests/stubs/mex.h declares the part of the MEX C API the gateway uses (documented MATLAB signatures);
tests/stubs/mex_runtime.cpp implements it on heap arrays and adds a small driver.  The gateway is compiled unchanged,
linked against libgpz_hip.so, and called the way mex/GPz.m calls it; on the GPU box its results are compared with the
oracle.  Gateway convention checked: minFunc_2012/minFunc/mex/lbfgsProdC.c:7 (mexFunction signature), :24-25 (argument
checks reported with mexErrMsg*), :43 (outputs from mxCreate*)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import make_problem, rel, grad_tol
from oracle import gpz_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(ROOT, "tests", "stubs")
TWIN = os.path.join(ROOT, "build", "libgpz_mex_twin.so")
SRCS = [os.path.join(ROOT, "mex", "gpz_mex.cpp"), os.path.join(STUBS, "mex_runtime.cpp")]


def test_gateway_compiles_against_the_mex_api_declarations():
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + STUBS,
                        "-I" + os.path.join(ROOT, "include"), SRCS[0]], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _twin():
    from gpz_amd import _lib
    _lib.load()          # the package's loading order (torch's bundled HIP runtime first, see _lib.load) before the twin pulls libgpz_hip.so in
    libdir = os.path.join(ROOT, "gpz_amd", "lib")
    deps = SRCS + [os.path.join(STUBS, "mex.h"), os.path.join(ROOT, "include", "gpz_hip.h")]
    if not os.path.exists(TWIN) or any(os.path.getmtime(s) > os.path.getmtime(TWIN) for s in deps):
        os.makedirs(os.path.dirname(TWIN), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-Wall", "-I" + STUBS,
                               "-I" + os.path.join(ROOT, "include"), *SRCS, "-L" + libdir, "-lgpz_hip",
                               "-Wl,-rpath," + libdir, "-o", TWIN])
    lib = C.CDLL(TWIN)
    vp, sz = C.c_void_p, C.c_size_t
    for name, res, args in [("mexrt_double", vp, [vp, C.c_int, C.POINTER(sz)]), ("mexrt_logical", vp, [vp, sz]),
                            ("mexrt_string", vp, [C.c_char_p]), ("mexrt_struct", vp, []),
                            ("mexrt_set_field", None, [vp, C.c_char_p, vp]), ("mexrt_ndim", C.c_int, [vp]),
                            ("mexrt_dim", sz, [vp, C.c_int]), ("mexrt_call", C.c_int, [C.c_int, C.POINTER(vp), C.c_int, C.POINTER(vp)]),
                            ("mexrt_last_error", C.c_char_p, []), ("mexrt_last_id", C.c_char_p, []),
                            ("mexrt_locks", C.c_int, []), ("mexrt_unload", None, []), ("mxDestroyArray", None, [vp]),
                            ("mxGetPr", vp, [vp]), ("mxGetData", vp, [vp]), ("mxGetNumberOfElements", sz, [vp])]:
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    return lib


class MexError(Exception):
    def __init__(self, ident, msg):
        super().__init__(f"{ident}: {msg}")
        self.ident = ident


class Mex:
    """Marshals NumPy values the way MATLAB hands them to a MEX file and calls the gateway."""

    def __init__(self):
        self.lib = _twin()
        self.held = {}      # id(ndarray) -> (mxArray, ndarray): a MATLAB variable keeps its data pointer between calls

    def hold(self, *arrays):
        for a in arrays:
            if a is not None:
                self.held[id(a)] = (self._in(a), a)

    def poke(self, a, flat_index, value):
        """Write into the held mxArray of ``a`` in place (what MATLAB may do to an unshared variable)."""
        p = self.held[id(a)][0]
        C.cast(self.lib.mxGetPr(p), C.POINTER(C.c_double))[flat_index] = value

    def poke_logical(self, a, flat_index, value):
        """In-place write into a held logical mxArray (one byte per element)."""
        p = self.held[id(a)][0]
        C.cast(self.lib.mxGetData(p), C.POINTER(C.c_uint8))[flat_index] = 1 if value else 0

    def release(self):
        for p, _ in self.held.values():
            self.lib.mxDestroyArray(p)
        self.held = {}

    def _in(self, v):
        L = self.lib
        if v is None:
            return L.mexrt_double(None, 2, (C.c_size_t * 2)(0, 0))
        if isinstance(v, str):
            return L.mexrt_string(v.encode())
        if isinstance(v, dict):
            s = L.mexrt_struct()
            for key, val in v.items():
                L.mexrt_set_field(s, key.encode(), self._in(val))
            return s
        a = np.asarray(v)
        if a.dtype == bool:
            a = np.ascontiguousarray(a.ravel().astype(np.uint8))
            return L.mexrt_logical(a.ctypes.data, a.size)
        a = np.asarray(a, dtype=np.float64)
        if a.ndim < 2:
            a = a.reshape(-1, 1) if a.ndim == 1 else a.reshape(1, 1)
        f = np.asfortranarray(a)
        return L.mexrt_double(f.ctypes.data, f.ndim, (C.c_size_t * f.ndim)(*f.shape))

    def _out(self, p):
        L = self.lib
        shape = tuple(L.mexrt_dim(p, q) for q in range(L.mexrt_ndim(p)))
        n = L.mxGetNumberOfElements(p)
        buf = np.ctypeslib.as_array(C.cast(L.mxGetPr(p), C.POINTER(C.c_double)), shape=(n,)).copy() if n else np.zeros(0)
        return buf.reshape(shape, order="F")

    def __call__(self, nlhs, *args):
        L = self.lib
        owned = [None if id(a) in self.held else self._in(a) for a in args]
        prhs = (C.c_void_p * len(args))(*[self.held[id(a)][0] if o is None else o for a, o in zip(args, owned)])
        plhs = (C.c_void_p * max(nlhs, 1))()
        rc = L.mexrt_call(nlhs, plhs, len(args), prhs)
        for p in owned:
            if p is not None:
                L.mxDestroyArray(p)
        if rc:
            raise MexError(L.mexrt_last_id().decode(), L.mexrt_last_error().decode())
        outs = [self._out(plhs[q]) for q in range(max(nlhs, 1)) if plhs[q]]
        for q in range(max(nlhs, 1)):
            if plhs[q]:
                L.mxDestroyArray(plhs[q])
        return outs[0] if nlhs <= 1 and outs else outs


def model_struct(model, **extra):
    s = {"m": float(model.m), "d": float(model.d), "k": float(model.k), "method": model.method,
         "heteroscedastic": bool(model.heteroscedastic)}
    s.update(extra)
    return s


def test_gateway_links_and_rejects_bad_calls_without_a_gpu():
    mex = Mex()
    assert mex(1, "gpus") == 0.0
    mex(0, "reset")
    for args, ident in [(("nonsense",), "gpz:usage"), (("eval", np.zeros(3)), "gpz:usage"), (("dxy", np.zeros((2, 2))), "gpz:usage"),
                        (("inv_logdet", np.zeros((2, 3))), "gpz:size"), (("phi",), "gpz:state"), (("pinv_mode", 5.0), "gpz:usage"),
                        (("dxy", np.zeros((2, 2)), np.zeros((2, 3))), "gpz:size"),
                        (("eval", np.zeros(3), {"m": 2.0}, None, None, None, None, None, None), "gpz:model")]:
        with pytest.raises(MexError) as e:
            mex(1, *args)
        assert e.value.ident == ident, (args[0], e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("method,psi,nanfrac,k", [("VC", False, 0.0, 1), ("VD", True, 0.3, 2), ("GC", True, 0.0, 1)])
def test_gateway_eval_solve_phi_match_the_oracle(method, psi, nanfrac, k):
    """What mex/GPz.m does for [f,g] = GPz(theta,...) (globals from stats), [~,~,w,iSigma_w,PHI] = GPz(...) and the
    closure-change detection: new omega values in a NEW array of the same size must rebuild the device context."""
    n = 1500
    model, theta, X, Y, Psi, rng = make_problem(n, 5, 24, k, method, True, seed=51, psi=psi, nanfrac=nanfrac)
    r2 = np.random.default_rng(4)
    om = r2.random((n, 1)) + 0.5
    tr = r2.random(n) < 0.75
    ms = model_struct(model)
    mex = Mex()
    ref = O.GPz(theta, model, X, Y, Psi, om, tr, ~tr)
    tol = grad_tol(ref.cond)
    f, g, st = mex(3, "eval", theta, ms, X, Y, Psi, om, tr, ~tr)
    assert mex(1, "gpus") == 1.0 and mex.lib.mexrt_locks() == 1
    assert f.shape == (1, 1) and g.shape == (theta.size, 1) and st.shape == (4, 1)
    assert abs(f[0, 0] - ref.nlogML) <= 1e-8 * abs(ref.nlogML) and rel(g.ravel(), ref.grad) <= tol
    for q, key in enumerate(["trainRMSE", "trainLL", "validRMSE", "validLL"]):
        assert abs(st[q, 0] - ref.stats[key]) <= 1e-10 * max(1.0, abs(ref.stats[key]))
    f_again = mex(1, "eval", theta, ms, X, Y, Psi, om, tr, ~tr)          # nlhs = 1: only f comes back
    assert f_again[0, 0] == f[0, 0]
    r4 = O.GPz(theta, model, X, Y, Psi, om, tr, ~tr, nargout=5)
    w, iS, part = mex(3, "solve", theta, ms, X, Y, Psi, om, tr, ~tr)
    assert w.shape == (model.m, k) and iS.shape == (model.m, model.m, k) and part.shape == (1, k)
    assert rel(w, r4.w) <= tol and rel(iS, r4.iSigma_w) <= tol and rel(part.ravel(), r4.nlogML) <= 1e-8
    PHI = mex(1, "phi")
    assert PHI.shape == (int(tr.sum()), model.m) and rel(PHI, r4.PHI) <= 1e-10
    # closure change: other weights, other masks, other model -> the live context must not be reused
    om2 = om * (1.0 + r2.random((n, 1)))
    ref2 = O.GPz(theta, model, X, Y, Psi, om2, tr, ~tr)
    f2 = mex(1, "eval", theta, ms, X, Y, Psi, om2, tr, ~tr)
    assert abs(f2[0, 0] - ref2.nlogML) <= 1e-8 * abs(ref2.nlogML) and f2[0, 0] != f[0, 0]
    tr3 = r2.random(n) < 0.75
    ref3 = O.GPz(theta, model, X, Y, Psi, om2, tr3, None)
    f3, g3, st3 = mex(3, "eval", theta, ms, X, Y, Psi, om2, tr3, None)
    assert abs(f3[0, 0] - ref3.nlogML) <= 1e-8 * abs(ref3.nlogML) and np.isnan(st3[2:, 0]).all()
    with pytest.raises(MexError) as e:
        mex(1, "eval", theta[:-1], ms, X, Y, Psi, om2, tr3, None)
    assert e.value.ident == "gpz:theta"
    with pytest.raises(MexError) as e:
        mex(1, "solve", theta[:-1], ms, X, Y, Psi, om2, tr3, None)
    assert e.value.ident == "gpz:theta"
    with pytest.raises(MexError) as e:
        mex(1, "eval", theta, ms, X.astype(np.float64)[:, :-1], Y, None, None, None, None)
    assert e.value.ident == "gpz:size"
    mex(0, "reset")
    assert mex(1, "gpus") == 0.0
    # a closure whose arrays keep their data pointers (MATLAB variables between minFunc iterations): ONE build, however
    # many evaluations; an in-place edit of a sampled element, or another model field, is seen and rebuilds
    mex.hold(X, Y, Psi, om, tr)
    b0 = mex(1, "builds")[0, 0]
    fa = mex(1, "eval", theta, ms, X, Y, Psi, om, tr, None)
    fb = mex(1, "eval", theta + 1e-3, ms, X, Y, Psi, om, tr, None)
    mex(3, "solve", theta, ms, X, Y, Psi, om, tr, None)
    assert mex(1, "builds")[0, 0] == b0 + 1 and fa[0, 0] != fb[0, 0]
    hit = next(i for i in np.flatnonzero(tr) if i % (n // 256) == 0)       # a training row among the 256 sampled elements
    mex.poke(om, int(hit), om[hit, 0] * 3.0)
    fc = mex(1, "eval", theta, ms, X, Y, Psi, om, tr, None)
    assert mex(1, "builds")[0, 0] == b0 + 2 and fc[0, 0] != fa[0, 0]
    # the n-sized arrays are hashed COMPLETELY - an in-place edit of ONE element that a strided sample would
    # miss (MATLAB: training(bad) = false on an unshared variable keeps the pointer) must rebuild the context
    miss = next(i for i in np.flatnonzero(tr) if i % (n // 256) != 0 and i != n - 1)
    mex.poke_logical(tr, int(miss), False)
    tr_edit = tr.copy(); tr_edit[miss] = False
    om_now = om.copy(); om_now[hit, 0] *= 3.0
    fd = mex(1, "eval", theta, ms, X, Y, Psi, om, tr, None)
    ref_d = O.GPz(theta, model, X, Y, Psi, om_now, tr_edit, None)
    assert mex(1, "builds")[0, 0] == b0 + 3 and abs(fd[0, 0] - ref_d.nlogML) <= 1e-8 * abs(ref_d.nlogML) and fd[0, 0] != fc[0, 0]
    miss_y = next(i for i in np.flatnonzero(tr_edit) if i % (Y.size // 256 or 1) != 0 and i != Y.size - 1)
    mex.poke(Y, int(miss_y), 0.123)
    fe = mex(1, "eval", theta, ms, X, Y, Psi, om, tr, None)
    assert mex(1, "builds")[0, 0] == b0 + 4 and fe[0, 0] != fd[0, 0]
    b0 += 2
    if model.heteroscedastic:
        ms2 = dict(ms, heteroscedastic=False)
        with pytest.raises(MexError) as e:          # theta is now too long for the model: the context was rebuilt for it
            mex(1, "eval", theta, ms2, X, Y, Psi, om, tr, None)
        assert e.value.ident == "gpz:theta" and mex(1, "builds")[0, 0] == b0 + 3
    mex.release()
    mex(0, "reset")
    mex.lib.mexrt_unload()


@pytest.mark.gpu
def test_gateway_optional_model_fields():
    """model.n_gpus and model.dtype (optional fields of the model struct): n_gpus = 1 is honoured, more GPUs than the box has is
    an error from the library reported through mexErrMsgIdAndTxt, dtype = 'f32' selects the fp32 pair kernels."""
    import gpz_amd
    n, d, m = 600, 6, 10
    model, theta, X, Y, Psi, rng = make_problem(n, d, m, 1, "VC", True, seed=61, psi=True)
    from test_gpu_parity import _well_conditioned_gamma
    theta = _well_conditioned_gamma(model, theta, rng)
    mex = Mex()
    f64 = mex(1, "eval", theta, model_struct(model, n_gpus=1.0), X, Y, Psi, None, None, None)
    assert mex(1, "gpus") == 1.0
    f32 = mex(1, "eval", theta, model_struct(model, n_gpus=1.0, dtype="f32"), X, Y, Psi, None, None, None)
    ref = O.GPz(theta, model, X, Y, Psi)
    assert abs(f64[0, 0] - ref.nlogML) <= 1e-8 * abs(ref.nlogML)
    assert abs(f32[0, 0] - ref.nlogML) <= 1e-4 * abs(ref.nlogML) and f32[0, 0] != f64[0, 0]
    with pytest.raises(MexError) as e:
        mex(1, "eval", theta, model_struct(model, n_gpus=float(gpz_amd.device_count() + 2)), X, Y, Psi, None, None, None)
    assert e.value.ident == "gpz:create"
    mex(0, "reset")
    mex.lib.mexrt_unload()


@pytest.mark.gpu
def test_gateway_per_output_weights_and_exact_identity_of_the_big_arrays():
    """(1) omega n x k through the gateway (GPz.m:48 omega(training,:); getOmega.m:19 on a k-column Y), any other shape refused.
    (2) X and Psi are sampled on every call and hashed completely every K-th: an in-place edit of an element NO sample looks at
    (MATLAB: X(i, c) = v on an unshared variable keeps the pointer) is caught at the latest K - 1 evaluations later — with
    model.verify_every = 1 on the very next call — rebuilds the device context and evaluates on the new data (the closure
    semantics of train.m:40: f sees the arrays as they are when it is called)."""
    n, d, m, k = 1500, 5, 16, 2
    model, theta, X, Y, Psi, rng = make_problem(n, d, m, k, "VD", True, seed=71, psi=True)
    om = rng.random((n, k)) + 0.5
    om[:, 1] *= 2.0
    tr = rng.random(n) < 0.8
    mex = Mex()
    ms = model_struct(model, verify_every=1.0)
    ref = O.GPz(theta, model, X, Y, Psi, om, tr, ~tr)
    f, g, st = mex(3, "eval", theta, ms, X, Y, Psi, om, tr, ~tr)
    assert abs(f[0, 0] - ref.nlogML) <= 1e-8 * abs(ref.nlogML) and rel(g.ravel(), ref.grad) <= grad_tol(ref.cond)
    for q, key in enumerate(["trainRMSE", "trainLL", "validRMSE", "validLL"]):
        assert abs(st[q, 0] - ref.stats[key]) <= 1e-10 * max(1.0, abs(ref.stats[key]))
    for bad in (np.ones((n, 3)), np.ones((n - 1, 1)), np.ones((n - 1, 2))):
        with pytest.raises(MexError) as e:
            mex(1, "eval", theta, ms, X, Y, Psi, bad, tr, ~tr)
        assert e.value.ident == "gpz:size"
    mex(0, "reset")
    # held arrays: one build for many evaluations; an edit where the 256 samples do not look
    mex.hold(X, Y, Psi, om, tr)
    b0 = mex(1, "builds")[0, 0]
    stale0 = mex(1, "verify_info")[0, 3]
    fa = mex(1, "eval", theta, ms, X, Y, Psi, om, tr, None)
    fb = mex(1, "eval", theta, ms, X, Y, Psi, om, tr, None)
    assert mex(1, "builds")[0, 0] == b0 + 1 and fa[0, 0] == fb[0, 0]
    step = X.size // 256
    row = next(i for i in np.flatnonzero(tr) if i % step != 0 and i != X.size - 1)      # flat index i = row i of column 0
    mex.poke(X, int(row), X[row, 0] + 0.75)
    X2 = X.copy(); X2[row, 0] += 0.75
    fc = mex(1, "eval", theta, ms, X, Y, Psi, om, tr, None)
    ref_c = O.GPz(theta, model, X2, Y, Psi, om, tr, None)
    info = mex(1, "verify_info")
    assert mex(1, "builds")[0, 0] == b0 + 2 and info[0, 0] == 1.0 and info[0, 3] == stale0 + 1 and info[0, 1] > 0.0
    assert fc[0, 0] != fa[0, 0] and abs(fc[0, 0] - ref_c.nlogML) <= 1e-8 * abs(ref_c.nlogML)
    prow = next(i for i in np.flatnonzero(tr) if i % (Psi.size // 256) != 0)
    mex.poke(Psi, int(prow), Psi[prow, 0] + 0.5)
    Psi2 = Psi.copy(); Psi2[prow, 0] += 0.5
    fd = mex(1, "eval", theta, ms, X, Y, Psi, om, tr, None)
    ref_d = O.GPz(theta, model, X2, Y, Psi2, om, tr, None)
    assert mex(1, "builds")[0, 0] == b0 + 3 and abs(fd[0, 0] - ref_d.nlogML) <= 1e-8 * abs(ref_d.nlogML)
    # verify_every = 4: the edit is seen on the 4th call after the last check, not before; never later
    ms4 = model_struct(model, verify_every=4.0)
    f0 = mex(1, "eval", theta, ms4, X, Y, Psi, om, tr, None)            # same closure, same context: call 1 of 4
    assert mex(1, "builds")[0, 0] == b0 + 3 and f0[0, 0] == fd[0, 0]
    mex.poke(X, int(row), X2[row, 0] - 0.75)                             # back to the original value
    seen = []
    for _ in range(4):
        seen.append(mex(1, "eval", theta, ms4, X, Y, Psi, om, tr, None)[0, 0])
    ref_e = O.GPz(theta, model, X, Y, Psi2, om, tr, None)
    assert mex(1, "builds")[0, 0] == b0 + 4 and seen[0] == fd[0, 0]     # stale for at most K - 1 = 3 calls ...
    assert abs(seen[-1] - ref_e.nlogML) <= 1e-8 * abs(ref_e.nlogML)     # ... and right from the K-th on
    # no verify_every field: K follows the measured times (hash <= 2 % of the evaluations), between 1 and 64
    mex(1, "eval", theta, model_struct(model), X, Y, Psi, om, tr, None)
    mex(1, "eval", theta, model_struct(model), X, Y, Psi, om, tr, None)
    info = mex(1, "verify_info")
    assert 1.0 <= info[0, 0] <= 64.0 and info[0, 2] > 0.0
    with pytest.raises(MexError) as e:
        mex(1, "eval", theta, model_struct(model, verify_every=0.0), X, Y, Psi, om, tr, None)
    assert e.value.ident == "gpz:model"
    mex.release()
    mex(0, "reset")
    mex.lib.mexrt_unload()


@pytest.mark.gpu
def test_gateway_reducer_field_and_loopback_shards():
    """model.reducer = 'loopback': model.n_gpus shards on one device behind the same gateway (single-GPU MATLAB hosts)."""
    model, theta, X, Y, Psi, rng = make_problem(900, 4, 8, 1, "VD", True, seed=8)
    mex = Mex()
    one = mex(2, "eval", theta, model_struct(model, n_gpus=1.0), X, Y, None, None, None, None)
    three = mex(2, "eval", theta, model_struct(model, n_gpus=3.0, reducer="loopback"), X, Y, None, None, None, None)
    assert mex(1, "gpus") == 3.0
    comm = mex(1, "comm")      # per rank [ncclCommCount ncclCommUserRank ncclCommCuDevice hipDevice]: no RCCL communicator behind loopback shards
    assert comm.shape == (3, 4) and np.all(comm[:, :3] == -1.0) and np.all(comm[:, 3] == 0.0)
    assert abs(one[0][0, 0] - three[0][0, 0]) <= 1e-12 * abs(one[0][0, 0]) and rel(three[1], one[1]) <= 1e-9
    with pytest.raises(MexError) as e:
        mex(1, "eval", theta, model_struct(model, reducer="carrier-pigeon"), X, Y, None, None, None, None)
    assert e.value.ident == "gpz:model"
    mex(0, "reset")
    mex.lib.mexrt_unload()


@pytest.mark.gpu
def test_gateway_standalone_entries_validate_their_arguments():
    """'getphi' / 'predict' / 'prior' must check numel(theta), w, iSigma_w, priors and the shapes of X / Psi
    against the model before the library reads them (a short theta would be an out-of-bounds host read)."""
    model, theta, X, Y, _, rng = make_problem(120, 3, 5, 1, "VC", True, seed=12)
    ms = model_struct(model)
    m, d = model.m, model.d
    w, iS, pri = np.zeros((m, 1)), np.eye(m)[:, :, None], np.full(m, 1.0 / m)
    cube = np.zeros((d, d, X.shape[0]))
    mex = Mex()
    bad = [(("getphi", ms, theta[:-1], X, None), "gpz:theta"),
           (("getphi", ms, theta, X[:, :-1], None), "gpz:size"),
           (("getphi", ms, theta, X, cube[:, :, :-1]), "gpz:size"),
           (("getphi", ms, theta, X, np.zeros((X.shape[0], d + 1))), "gpz:size"),
           (("getphi", dict(ms, method="XX"), theta, X, None), "gpz:model"),
           (("prior", ms, theta[:-2], X, None), "gpz:theta"),
           (("prior", ms, theta, X[:, :-1], None), "gpz:size"),
           (("predict", ms, theta[:-1], w, iS, pri, X, None), "gpz:theta"),
           (("predict", ms, theta, w[:-1], iS, pri, X, None), "gpz:size"),
           (("predict", ms, theta, w, iS[:, :-1], pri, X, None), "gpz:size"),
           (("predict", ms, theta, w, iS, pri[:-1], X, None), "gpz:size"),
           (("predict", ms, theta, w, iS, pri, X, cube[:-1]), "gpz:size")]
    for args, ident in bad:
        with pytest.raises(MexError) as e:
            mex(1, *args)
        assert e.value.ident == ident, (args[0], ident, str(e.value))
    mex(5, "predict", ms, theta, w, iS, None, X, None)        # empty priors are allowed without missing values
    mex.lib.mexrt_unload()


@pytest.mark.gpu
def test_gateway_standalone_entries():
    rng = np.random.default_rng(2)
    A = rng.standard_normal((30, 30))
    A = A @ A.T + 30 * np.eye(30)
    mex = Mex()
    Xi, ld = mex(2, "inv_logdet", A)
    Xr, lr = O.inv_logdet(A)
    assert rel(Xi, Xr) <= 1e-10 and abs(ld[0, 0] - lr) <= 1e-10 * abs(lr)
    P, Q = rng.standard_normal((40, 4)), rng.standard_normal((9, 4))
    assert rel(mex(1, "dxy", P, Q), O.Dxy(P, Q)) <= 1e-12
    model, theta, X, Y, Psi, _ = make_problem(200, 4, 12, 1, "VC", True, seed=3)
    PHI, lnb, N = mex(3, "getphi", model_struct(model), theta, X, None)
    out = O.getPHI(X, None, theta, model, None)
    assert rel(PHI, out[0]) <= 1e-10 and rel(lnb, out[2]) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["VD", "GC"])
def test_gateway_predict_and_prior_entries(method):
    """gpz_mex('predict', ...) per NaN-pattern group as mex/predictDiag.m / predictCov.m call it (all four branches picked from
    what X and Psi contain) and gpz_mex('prior', ...) as mex/getPrior.m calls it, against the Python mirror, which the parity
    tests check against the oracle's predictDiag.m / predictCov.m / getPrior.m restatements."""
    import gpz_amd
    n, d, m = 300, 4, 10
    model, theta, X, Y, _, rng = make_problem(600, d, m, 1, method, True, seed=77)
    ctx = gpz_amd.GPzContext(model, X, Y)
    w, iS, _ = ctx.solve(theta)
    ctx.close()
    priors = gpz_amd.getPrior(X, None, theta, model)
    model.sets = {"best": {"theta": theta, "w": w, "iSigma_w": iS, "priors": priors}}
    Xs = rng.standard_normal((n, d))
    Xn = Xs.copy(); Xn[:, 1] = np.nan                                           # ONE pattern: dimension 1 missing everywhere
    if method[1] == "C":
        Psi = np.zeros((d, d, n)); Psi[np.arange(d), np.arange(d), :] = rng.gamma(1.0, 0.05, (d, n))
    else:
        Psi = rng.gamma(1.0, 0.05, (n, d))
    ms = model_struct(model)
    mex = Mex()
    pr = mex(1, "prior", ms, theta, X, None)
    assert pr.shape == (1, m) and rel(pr.ravel(), priors) <= 1e-12
    for XX, PP in ((Xs, None), (Xs, Psi), (Xn, None), (Xn, Psi)):
        mu, nu, beta_i, gamma, PHI = mex(5, "predict", ms, theta, w, iS, priors, XX, PP)
        ref = gpz_amd.predict(XX, model, Psi=PP)                                # mu, sigma, nu, beta_i, gamma, PHI, ...
        assert mu.shape == (n, 1) and PHI.shape == (n, m)
        assert rel(mu, ref[0]) <= 1e-12 and rel(nu, ref[2]) <= 1e-12 and rel(beta_i, ref[3]) <= 1e-12
        assert np.max(np.abs(gamma - ref[4])) <= 1e-12 * max(1.0, np.max(np.abs(ref[4]))) and rel(PHI, ref[5]) <= 1e-12
