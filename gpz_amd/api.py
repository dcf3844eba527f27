"""Host-side mirror of the reference's interface for the objective/gradient path.

Same names, argument order and meaning as the MATLAB functions they stand in for
(paths relative to the OxfordML/GPz tree):

    GPz(theta, model, X, Y, Psi, omega, training, validation)   GPz/GPz.m:1
    getPHI(X, Psi, theta, model, selection)                      GPz/getPHI.m:1
    inv_logdet(X)                                                GPz/inv_logdet.m:1
    Dxy(X, Y)                                                    GPz/Dxy.m:1
    predict(X, model, whichSet=..., selection=...)               GPz/predict.m:1 (no-Psi / no-NaN branch)

Everything numeric happens in libgpz_hip.so on the GPU; this file only marshals numpy arrays
(column-major, like MATLAB) across the C ABI.  ``model`` is any object with the reference's struct fields
``m, d, k, method, heteroscedastic`` (+ ``muX, sdX, muY`` and ``sets`` for predict).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _lib

# the reference's globals (GPz.m:3-7), refreshed by every 2-output GPz() call and left alone by solve-only calls
globals_ = {"trainRMSE": None, "trainLL": None, "validRMSE": None, "validLL": None}


@dataclass
class Model:
    """model struct of init.m:16-20,41-43,86."""
    m: int
    d: int
    k: int = 1
    method: str = "VD"
    heteroscedastic: bool = True
    g_dim: int = 0
    muX: Optional[np.ndarray] = None
    sdX: Optional[np.ndarray] = None
    muY: Optional[np.ndarray] = None
    sets: dict = field(default_factory=dict)

    def __post_init__(self):
        if self.g_dim == 0:
            self.g_dim = {"GL": 1, "VL": self.m, "GD": self.d, "VD": self.m * self.d, "GC": self.d ** 2,
                          "VC": self.d ** 2 * self.m}[self.method]
        if self.muX is None:
            self.muX = np.zeros(self.d)
        if self.sdX is None:
            self.sdX = np.ones(self.d)
        if self.muY is None:
            self.muY = np.zeros(self.k)


def _desc(model, device=0, stream=None, rank=0, world=1, dtype="f64"):
    ds = _lib.gpz_desc()
    if dtype not in ("f64", "f32"):
        raise ValueError("dtype must be 'f64' or 'f32'")
    ds.dtype = 1 if dtype == "f32" else 0
    ds.d, ds.m, ds.k = int(model.d), int(model.m), int(model.k)
    ds.method = str(model.method).encode()
    ds.heteroscedastic = 1 if model.heteroscedastic else 0
    ds.device = int(device)
    ds.stream = stream
    ds.rank, ds.world = int(rank), int(world)
    return ds


def _psi_kind(model, psi):
    """psi_kind of the C ABI from the array's shape: n x d for the diagonal kinds (1), d x d x n cubes for GC/VC (2), and
    n x d per-dimension variances given to GC/VC (3: the diagonal cubes of fixPsi.m:27-31, expanded by the library)."""
    if psi is None:
        return 0
    if psi.ndim == 3:
        return 2
    return 3 if str(model.method)[1] == "C" else 1


def _f64(a, ndim=None):
    if a is None:
        return None
    a = np.asarray(a, dtype=np.float64)
    if ndim == 2 and a.ndim == 1:
        a = a[:, None]
    return np.asfortranarray(a)


def _omega(omega, n_tot, k):
    """omega as the reference takes it: n x 1 (one weight per row for every output) or n x k (per-output weights,
    GPz.m:48 ``omega(training,:)``; getOmega.m:19 returns ``(1+Y).^-2``, n x k for a k-column Y)."""
    om = _f64(omega, 2)
    if om is not None and (om.ndim != 2 or om.shape[0] != n_tot or om.shape[1] not in (1, k)):
        raise ValueError("omega must be n x 1 or n x k")
    return om


def _mask(a, n):
    if a is None:
        return None
    a = np.asarray(a)
    if a.size == 0:
        return None
    a = np.ascontiguousarray(a.astype(bool).ravel().astype(np.uint8))
    if a.size != n:
        raise ValueError("mask length must equal the number of rows of X")
    return a


class GPzContext:
    """The closure ``f = @(theta) GPz(theta,model,X,Y,Psi,omega,training,validation)`` (train.m:40) with the data
    resident on the GPU.  ``X`` must already be normalised and ``Y`` centred, as train.m:30-33 does before
    building the closure."""

    def __init__(self, model, X, Y, Psi=None, omega=None, training=None, validation=None, device=0, stream=None,
                 rank=0, world=1, allreduce=None, dtype="f64", patterns=None):
        """patterns: NaN-pattern table of the whole data set (G x d bool, True = missing, first-occurrence order; see
        gpz_amd.dist.nan_patterns) — needed by row-sharded GC/VC runs with missing values."""
        lib = _lib.load()
        X = _f64(X, 2)
        Y = _f64(Y, 2)
        n_tot = X.shape[0]
        if X.shape[1] != model.d or Y.shape != (n_tot, model.k):
            raise ValueError("X must be n x d and Y n x k")
        om = _omega(omega, n_tot, model.k)
        psi_kind = 0
        psi = None
        if Psi is not None:
            psi = _f64(Psi)
            psi_kind = _psi_kind(model, psi)
        self._tr = _mask(training, n_tot)
        self._va = _mask(validation, n_tot)
        self.model = model
        self._desc = _desc(model, device, stream, rank, world, dtype)
        self._desc.omega_cols = 0 if om is None else om.shape[1]
        h = C.c_void_p()
        pat = None
        if patterns is not None:
            pat = np.ascontiguousarray(np.asarray(patterns, dtype=bool).reshape(-1, model.d).astype(np.uint8))
        _lib.check(lib.gpz_ctx_create_sharded(
            C.byref(self._desc), n_tot, _lib.dptr(X), _lib.dptr(Y), _lib.dptr(psi), psi_kind, _lib.dptr(om),
            None if self._tr is None else self._tr.ctypes.data_as(_lib.c_uint8_p),
            None if self._va is None else self._va.ctypes.data_as(_lib.c_uint8_p),
            None if pat is None else pat.ctypes.data_as(_lib.c_uint8_p), 0 if pat is None else pat.shape[0], C.byref(h)))
        self._h = h
        self._lib = lib
        self.p = int(lib.gpz_theta_len(h))
        self.n_train = int(lib.gpz_n_train(h))
        self.n_valid = int(lib.gpz_n_valid(h))
        self.stats = {}
        self.info = 0
        self.n_global = self.n_train
        self._cb = None
        if allreduce is not None:
            self._cb = _lib.ALLREDUCE_FN(allreduce)
            _lib.check(lib.gpz_ctx_set_allreduce(h, self._cb, None))

    def set_allreduce(self, allreduce):
        """Install (or replace) the caller-supplied all-reduce hook of a sharded context (gpz_ctx_set_allreduce)."""
        self._cb = _lib.ALLREDUCE_FN(allreduce)
        _lib.check(self._lib.gpz_ctx_set_allreduce(self._h, self._cb, None))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpz_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval(self, theta):
        """[nlogML, grad] = GPz(theta, ...) plus the four global statistics (self.stats)."""
        theta = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).ravel())
        if theta.size != self.p:
            raise ValueError(f"theta must have {self.p} elements")
        f = C.c_double()
        g = np.empty(self.p)
        st = (C.c_double * 4)(float("nan"), float("nan"), float("nan"), float("nan"))
        dg = (C.c_double * 2)()
        _lib.check(self._lib.gpz_eval(self._h, _lib.dptr(theta), C.byref(f), _lib.dptr(g), st, dg))
        self.stats = {"trainRMSE": st[0], "trainLL": st[1]}
        if self._va is not None:     # a mask that selects no row gives NaN, as GPz.m:258-259 does (0/0)
            self.stats.update(validRMSE=st[2], validLL=st[3])
        self.info = int(dg[0])
        self.n_global = int(dg[1])
        return f.value, g

    def eval_dev(self, theta_t):
        """gpz_eval_dev: theta_t is a float64 CUDA tensor on the context's device; returns (f, g) with g a new CUDA
        tensor — theta and the gradient never visit the host."""
        import torch
        if theta_t.numel() != self.p or theta_t.dtype != torch.float64 or not theta_t.is_cuda:
            raise ValueError(f"theta must be a float64 CUDA tensor of {self.p} elements")
        theta_t = theta_t.contiguous()
        g = torch.empty_like(theta_t)
        torch.cuda.current_stream(theta_t.device).synchronize()     # the context runs on its own stream
        f = C.c_double()
        st = (C.c_double * 4)(float("nan"), float("nan"), float("nan"), float("nan"))
        dg = (C.c_double * 2)()
        _lib.check(self._lib.gpz_eval_dev(self._h, theta_t.data_ptr(), C.byref(f), g.data_ptr(), st, dg))
        self.stats = {"trainRMSE": st[0], "trainLL": st[1]}
        if self._va is not None:
            self.stats.update(validRMSE=st[2], validLL=st[3])
        self.info = int(dg[0])
        self.n_global = int(dg[1])
        return f.value, g

    def solve(self, theta):
        """[~, ~, w, iSigma_w] = GPz(theta, ...)  (GPz.m:84-87); also returns the 1 x k partial nlogML."""
        theta = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).ravel())
        m, k = self.model.m, self.model.k
        w = np.empty((m, k), order="F")
        iS = np.empty((m, m, k), order="F")
        part = np.empty(k)
        _lib.check(self._lib.gpz_solve(self._h, _lib.dptr(theta), _lib.dptr(w), _lib.dptr(iS), _lib.dptr(part)))
        return w, iS, part

    def set_pinv_mode(self, mode):
        """Branch of inv_logdet.m:7-12: 0 = Cholesky, SVD pseudo-inverse when SIGMA is nearly singular (default);
        1 = always the truncating SVD route; -1 = never."""
        _lib.check(self._lib.gpz_ctx_set_pinv_mode(self._h, int(mode)))

    def last_pinv(self):
        """(route taken, rank kept, largest singular value, Jacobi sweeps) of the last eval/solve."""
        out = (C.c_double * 4)()
        _lib.check(self._lib.gpz_ctx_last_pinv(self._h, out))
        return bool(out[0]), int(out[1]), float(out[2]), int(out[3])

    def phi(self):
        """PHI (n_train x m) of the last eval/solve — the 5th output of GPz.m:1."""
        out = np.empty((self.n_train, self.model.m), order="F")
        _lib.check(self._lib.gpz_get_phi(self._h, _lib.dptr(out)))
        return out

    def enable_timing(self, on=True):
        """on = True / 1: HIP events around every stage (eager launches); 2: around the dominant stages only (PHI build, PHI'W PHI,
        T = PHI [inv(SIGMA) | w], moments), the evaluation still replayed as hipGraph segments; False / 0: off."""
        _lib.check(self._lib.gpz_ctx_enable_timing(self._h, int(on)))

    def reset_timings(self):
        _lib.check(self._lib.gpz_ctx_reset_timings(self._h))

    def timings(self):
        """{stage: (total_ms, calls)} measured with HIP events on the context's stream."""
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        calls = (C.c_int64 * cap)()
        n = self._lib.gpz_ctx_timings(self._h, names, ms, calls, cap)
        return {names[i].decode(): (ms[i], int(calls[i])) for i in range(min(n, cap))}

    def route(self):
        """Which kernels this context runs and the state of its evaluation graph (gpz_ctx_route), as one line of text."""
        buf = C.create_string_buffer(512)
        self._lib.gpz_ctx_route(self._h, buf, 512)
        return buf.value.decode()

    def comm_info(self):
        """What the RCCL communicator behind this context's all-reduce reports about itself (gpz_ctx_comm_info):
        {"nccl_count", "nccl_rank", "nccl_device"} (-1 without an in-library communicator), "hip_device", "pci_bus_id"."""
        return _comm_info(lambda info, bus, cap: self._lib.gpz_ctx_comm_info(self._h, info, bus, cap))


class GPzMulti:
    """The same closure on several GPUs behind ONE synchronous call (gpz_mgpu_* of the C ABI): the library splits the
    training-selected rows into contiguous blocks, one per device, drives every device from its own host thread and
    all-reduces the m x m / m x (d^2+d) partials with RCCL itself — no torch.distributed, no Python in the evaluation.
    What a single MATLAB process reaches through the MEX gateway (minFunc.m:314 calls funObj once and waits).

    n_gpus None/0: every device of the node.  reducer "loopback": all shards on ONE device with the library's own
    rank-ordered reducer — how the sharded path is exercised on single-GPU machines."""

    def __init__(self, model, X, Y, Psi=None, omega=None, training=None, validation=None, n_gpus=None, devices=None,
                 reducer="rccl", dtype="f64"):
        lib = _lib.load()
        X = _f64(X, 2)
        Y = _f64(Y, 2)
        n_tot = X.shape[0]
        if X.shape[1] != model.d or Y.shape != (n_tot, model.k):
            raise ValueError("X must be n x d and Y n x k")
        om = _omega(omega, n_tot, model.k)
        psi, psi_kind = None, 0
        if Psi is not None:
            psi = _f64(Psi)
            psi_kind = _psi_kind(model, psi)
        self._tr = _mask(training, n_tot)
        self._va = _mask(validation, n_tot)
        if reducer not in ("rccl", "loopback"):
            raise ValueError("reducer must be 'rccl' or 'loopback'")
        dev = None
        if devices is not None:
            dev = np.ascontiguousarray(np.asarray(devices, dtype=np.int32))
            n_gpus = dev.size
        self.model = model
        self._desc = _desc(model, 0, None, 0, 1, dtype)
        self._desc.omega_cols = 0 if om is None else om.shape[1]
        h = C.c_void_p()
        _lib.check(lib.gpz_mgpu_create(
            C.byref(self._desc), int(n_gpus or 0), None if dev is None else dev.ctypes.data_as(_lib.c_int32_p),
            1 if reducer == "loopback" else 0, n_tot, _lib.dptr(X), _lib.dptr(Y), _lib.dptr(psi), psi_kind, _lib.dptr(om),
            None if self._tr is None else self._tr.ctypes.data_as(_lib.c_uint8_p),
            None if self._va is None else self._va.ctypes.data_as(_lib.c_uint8_p), C.byref(h)))
        self._h, self._lib = h, lib
        self.n_gpus = int(lib.gpz_mgpu_size(h))
        self.p = int(lib.gpz_mgpu_theta_len(h))
        self.rows_per_gpu = [int(lib.gpz_n_train(lib.gpz_mgpu_ctx(h, r))) for r in range(self.n_gpus)]
        self.n_train = sum(self.rows_per_gpu)
        self.stats, self.info, self.n_global = {}, 0, self.n_train

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpz_mgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval(self, theta):
        theta = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).ravel())
        if theta.size != self.p:
            raise ValueError(f"theta must have {self.p} elements")
        f = C.c_double()
        g = np.empty(self.p)
        st = (C.c_double * 4)(float("nan"), float("nan"), float("nan"), float("nan"))
        dg = (C.c_double * 2)()
        _lib.check(self._lib.gpz_mgpu_eval(self._h, _lib.dptr(theta), C.byref(f), _lib.dptr(g), st, dg))
        self.stats = {"trainRMSE": st[0], "trainLL": st[1]}
        if self._va is not None:
            self.stats.update(validRMSE=st[2], validLL=st[3])
        self.info, self.n_global = int(dg[0]), int(dg[1])
        return f.value, g

    def solve(self, theta):
        theta = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).ravel())
        if theta.size != self.p:
            raise ValueError(f"theta must have {self.p} elements")
        m, k = self.model.m, self.model.k
        w = np.empty((m, k), order="F")
        iS = np.empty((m, m, k), order="F")
        part = np.empty(k)
        _lib.check(self._lib.gpz_mgpu_solve(self._h, _lib.dptr(theta), _lib.dptr(w), _lib.dptr(iS), _lib.dptr(part)))
        return w, iS, part

    def _each(self):
        return [self._lib.gpz_mgpu_ctx(self._h, r) for r in range(self.n_gpus)]

    @property
    def alive(self):
        """False once a rank failed inside a call with the RCCL reducer (communicators aborted): re-create the handle."""
        return bool(self._lib.gpz_mgpu_alive(self._h))

    def debug_fail_at(self, rank, exchange):
        """Test hook (gpz_mgpu_debug_fail_at): the next call fails on `rank` at exchange point 1 or 2."""
        _lib.check(self._lib.gpz_mgpu_debug_fail_at(self._h, int(rank), int(exchange)))

    def comm_info(self, rank=0):
        """gpz_mgpu_comm_info of one rank (see GPzContext.comm_info)."""
        return _comm_info(lambda info, bus, cap: self._lib.gpz_mgpu_comm_info(self._h, int(rank), info, bus, cap))

    def route(self, rank=0):
        buf = C.create_string_buffer(512)
        self._lib.gpz_ctx_route(self._lib.gpz_mgpu_ctx(self._h, int(rank)), buf, 512)
        return buf.value.decode()

    def enable_timing(self, on=True):
        for c in self._each():
            _lib.check(self._lib.gpz_ctx_enable_timing(c, int(on)))

    def reset_timings(self):
        for c in self._each():
            _lib.check(self._lib.gpz_ctx_reset_timings(c))

    def set_pinv_mode(self, mode):
        for c in self._each():
            _lib.check(self._lib.gpz_ctx_set_pinv_mode(c, int(mode)))

    def timings(self, rank=0):
        """{stage: (total_ms, calls)} of one rank (HIP events on that device's stream)."""
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        calls = (C.c_int64 * cap)()
        n = self._lib.gpz_ctx_timings(self._lib.gpz_mgpu_ctx(self._h, rank), names, ms, calls, cap)
        return {names[i].decode(): (ms[i], int(calls[i])) for i in range(min(n, cap))}


def device_count():
    return int(_lib.load().gpz_device_count())


def _comm_info(call):
    info = (C.c_int32 * 4)()
    bus = C.create_string_buffer(64)
    _lib.check(call(info, bus, 64))
    return {"nccl_count": int(info[0]), "nccl_rank": int(info[1]), "nccl_device": int(info[2]), "hip_device": int(info[3]),
            "pci_bus_id": bus.value.decode()}


def rccl_origin():
    """Which RCCL the library bound to (dlopen at first use): '' when none was needed or found."""
    o = _lib.load().gpz_rccl_origin()
    return o.decode() if o else ""


_cache = {}


def _ctx_for(model, X, Y, Psi, omega, training, validation):
    """One context per closure, keyed on the identity of the model and of the data arrays.  The arrays are treated as
    immutable while cached (MATLAB's value semantics: a modified array is a new array there): editing X, Y, omega, the
    masks or the model IN PLACE between calls is not seen — call ``reset()`` after doing that.  The least recently
    created context is evicted once four are alive."""
    def ident(a):
        return None if a is None else (id(a), getattr(a, "shape", None))
    key = (id(model), model.m, model.d, model.k, model.method, bool(model.heteroscedastic),
           ident(X), ident(Y), ident(Psi), ident(omega), ident(training), ident(validation))
    ctx = _cache.get(key)
    if ctx is None:
        if len(_cache) >= 4:
            old = _cache.pop(next(iter(_cache)))        # FIFO: dicts keep insertion order
            old[0].close()
        ctx = (GPzContext(model, X, Y, Psi, omega, training, validation), (X, Y, Psi, omega, training, validation))
        _cache[key] = ctx
    return ctx[0]


def reset():
    """Drop cached contexts (``clear global`` / new data)."""
    for ctx, _ in _cache.values():
        ctx.close()
    _cache.clear()
    _lib.load().gpz_release_cached_memory()          # and the device buffers the library keeps for its next call


def GPz(theta, model, X, Y, Psi=None, omega=None, training=None, validation=None, nargout=2):
    """[nlogML,grad,w,iSigma_w,PHI] = GPz(theta,model,X,Y,Psi,omega,training,validation)   (GPz.m:1).

    ``nargout`` <= 2 returns (nlogML, grad) and refreshes ``globals_``; > 2 returns
    (nlogML_partial, 0, w, iSigma_w[, PHI]) exactly like the early return at GPz.m:84-87."""
    if Y is None:                      # GPz.m:34-40
        return 0.0, 0.0, 0.0, 0.0
    ctx = _ctx_for(model, X, Y, Psi, omega, training, validation)
    if nargout <= 2:
        f, g = ctx.eval(theta)
        globals_.update(ctx.stats)
        return f, g
    w, iS, part = ctx.solve(theta)
    out = (part, 0.0, w, iS)
    if nargout >= 5:
        out = out + (ctx.phi(),)
    return out


def getPHI(X, Psi, theta, model, selection=None, device=0, want_N=False):
    """[PHI,Gamma,lnBeta_i,N] = getPHI(X,Psi,theta,model,selection)   (getPHI.m:1); Gamma is the expanded
    parameter array of getPHI.m:26-40 (pure reshaping of theta, done on the host).  Psi: n x d for the diagonal
    kinds, d x d x n for GC/VC (the layouts fixPsi.m produces); X may contain NaN (missing inputs)."""
    lib = _lib.load()
    X = np.asarray(X, dtype=np.float64)
    psi = None if Psi is None else np.asarray(Psi, dtype=np.float64)
    if selection is not None:
        sel = np.asarray(selection, dtype=bool)
        X = X[sel]                                                     # getPHI.m:14
        if psi is not None:
            psi = psi[:, :, sel] if psi.ndim == 3 else psi[sel]        # getPHI.m:16-22
    X = _f64(X, 2)
    psi_kind = 0
    if psi is not None:
        psi = np.asfortranarray(psi)
        psi_kind = _psi_kind(model, psi)
    theta = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).ravel())
    ns = X.shape[0]
    PHI = np.empty((ns, model.m), order="F")
    lnB = np.empty((ns, model.k), order="F")
    N = np.empty((ns, model.m), order="F") if want_N else None
    ds = _desc(model, device)
    _lib.check(lib.gpz_phi(C.byref(ds), _lib.dptr(theta), _lib.dptr(X), ns, _lib.dptr(psi), psi_kind, _lib.dptr(PHI),
                           _lib.dptr(lnB), _lib.dptr(N)))
    out = (PHI, _expand_gamma(theta, model), lnB)
    return out + (N,) if want_N else out


def _expand_gamma(theta, model):
    m, d = model.m, model.d
    G = theta[m * d:m * d + model.g_dim]
    mt = model.method
    if mt == "GL":
        return np.full((m, d), G[0])
    if mt == "VL":
        return np.tile(G.reshape(m, 1), (1, d))
    if mt == "GD":
        return np.tile(G.reshape(1, d), (m, 1))
    if mt == "VD":
        return G.reshape((m, d), order="F")
    if mt == "GC":
        return np.repeat(G.reshape((d, d), order="F")[:, :, None], m, axis=2)
    return G.reshape((d, d, m), order="F")


def inv_logdet(X, device=0, return_info=False):
    """[Xi,logdet] = inv_logdet(X)   (inv_logdet.m:1-15) for symmetric X; info = singular values dropped by the
    truncation of inv_logdet.m:7-12 (0 = none), -1 = X not finite."""
    lib = _lib.load()
    A = _f64(X)
    m = A.shape[0]
    Xi = np.empty((m, m), order="F")
    ld = C.c_double()
    info = C.c_int32()
    _lib.check(lib.gpz_inv_logdet(_lib.dptr(A), m, device, _lib.dptr(Xi), C.byref(ld), C.byref(info)))
    if return_info:
        return Xi, ld.value, int(info.value)
    return Xi, ld.value


def Dxy(X, Y, device=0):
    """D = Dxy(X,Y)   (Dxy.m:1)."""
    lib = _lib.load()
    X = _f64(X, 2)
    Y = _f64(Y, 2)
    D = np.empty((X.shape[0], Y.shape[0]), order="F")
    _lib.check(lib.gpz_dxy(_lib.dptr(X), X.shape[0], _lib.dptr(Y), Y.shape[0], X.shape[1], device, _lib.dptr(D)))
    return D


def nan_groups(X, device=0):
    """Group id per row by NaN pattern, in first-occurrence order (the loop of getPHI.m:43-54)."""
    lib = _lib.load()
    X = _f64(X, 2)
    n, d = X.shape
    gid = np.empty(n, dtype=np.int32)
    ng = C.c_int32()
    _lib.check(lib.gpz_nan_groups(_lib.dptr(X), n, d, device, gid.ctypes.data_as(_lib.c_int32_p), C.byref(ng)))
    return gid, int(ng.value)


def predict(X, model, whichSet="best", Psi=None, selection=None, device=0, n_gpus=None):
    """[mu,sigma,nu,beta_i,gamma,PHI,w,iSigma_w] = predict(X,model,...)   (predict.m:1).  Rows are grouped by NaN
    pattern as predict.m:45-57 does; a group without missing values runs predictFull / predictNoisy, a group with
    missing values predictMissing / predictNoisyMissing (predictDiag.m:127-297, predictCov.m:134-337).
    n_gpus (0 = every GPU of the node): every group's rows are split into contiguous blocks over the GPUs
    (gpz_mgpu_predict; more blocks than GPUs = several blocks per GPU)."""
    lib = _lib.load()
    X = np.asarray(X, dtype=np.float64)
    psi = None if Psi is None else np.asarray(Psi, dtype=np.float64)
    if selection is not None:
        sel = np.asarray(selection, dtype=bool)
        X = X[sel]                                                   # predict.m:25
        if psi is not None:                                          # predict.m:27-33
            psi = psi[:, :, sel] if (model.method[1] == "C" and psi.ndim == 3) else psi[sel]
    st = model.sets[whichSet]                                        # predict.m:10-14
    Xn = _f64((X - model.muX) / model.sdX, 2)                        # predict.m:35-36
    theta = np.ascontiguousarray(np.asarray(st["theta"], dtype=np.float64).ravel())
    w = _f64(st["w"], 2)
    iS = np.asfortranarray(np.asarray(st["iSigma_w"], dtype=np.float64).reshape(model.m, model.m, model.k))
    ns, k, m = Xn.shape[0], model.k, model.m
    psin = None
    if psi is not None:
        from .host import fixPsi
        psin = np.asfortranarray(fixPsi(psi, ns, model.sdX, model.method))   # predict.m:43
    cube = psin is not None and psin.ndim == 3
    ds = _desc(model, device)
    mu = nu = beta_i = gamma = PHI = None            # allocated below unless the data is one group (then the outputs ARE the results)
    # predict.m:45-57: groups of identical NaN patterns.  The grouping itself is the library's (gpz_nan_groups, ids in first-
    # occurrence order as the reference's loop forms them; np.unique over the rows took a third of a 1e5-row predictFull); complete
    # data is one group and is passed through without gathering.
    if ns and np.isnan(Xn).any():
        gid, n_groups = nan_groups(Xn, device)
        order = np.argsort(gid, kind="stable")
        bounds = np.concatenate(([0], np.cumsum(np.bincount(gid, minlength=n_groups))))
        groups = [order[bounds[g]:bounds[g + 1]] for g in range(n_groups)]
    else:
        groups = [slice(None)] if ns else []
    for idx in groups:
        whole = isinstance(idx, slice)
        Xg = Xn if whole else _f64(Xn[idx], 2)
        ng = Xg.shape[0]
        first_missing = bool(np.isnan(Xg[0]).any())
        Pg = None if psin is None else (psin if whole else np.asfortranarray(psin[:, :, idx] if cube else psin[idx]))
        o_mu = np.empty((ng, k), order="F"); o_nu = np.empty((ng, k), order="F"); o_be = np.empty((ng, k), order="F")
        o_ga = np.zeros((ng, k), order="F"); o_ph = np.empty((ng, m), order="F")
        if n_gpus is not None:
            pri = st.get("priors")
            pri = np.full(m, 1.0 / m) if pri is None else np.ascontiguousarray(np.asarray(pri, dtype=np.float64).ravel())
            _lib.check(lib.gpz_mgpu_predict(C.byref(ds), int(n_gpus), None, _lib.dptr(theta), _lib.dptr(w), _lib.dptr(iS),
                                            _lib.dptr(pri), _lib.dptr(Xg), ng, _lib.dptr(Pg),
                                            0 if Pg is None else (2 if cube else 1), _lib.dptr(o_mu), _lib.dptr(o_nu),
                                            _lib.dptr(o_be), _lib.dptr(o_ga), _lib.dptr(o_ph)))
        elif not first_missing:
            if Pg is None:                                           # predictFull, gamma = 0 (predictDiag.m:74)
                _lib.check(lib.gpz_predict_full(C.byref(ds), _lib.dptr(theta), _lib.dptr(w), _lib.dptr(iS), _lib.dptr(Xg), ng,
                                                _lib.dptr(o_mu), _lib.dptr(o_nu), _lib.dptr(o_be), _lib.dptr(o_ph)))
            else:
                _lib.check(lib.gpz_predict_noisy(C.byref(ds), _lib.dptr(theta), _lib.dptr(w), _lib.dptr(iS), _lib.dptr(Xg), ng,
                                                 _lib.dptr(Pg), 2 if cube else 1, _lib.dptr(o_mu), _lib.dptr(o_nu),
                                                 _lib.dptr(o_be), _lib.dptr(o_ga), _lib.dptr(o_ph)))
        else:
            pri = st.get("priors")
            pri = np.full(m, 1.0 / m) if pri is None else np.ascontiguousarray(np.asarray(pri, dtype=np.float64).ravel())
            _lib.check(lib.gpz_predict_missing(C.byref(ds), _lib.dptr(theta), _lib.dptr(w), _lib.dptr(iS), _lib.dptr(pri),
                                               _lib.dptr(Xg), ng, _lib.dptr(Pg), 0 if Pg is None else (2 if cube else 1),
                                               _lib.dptr(o_mu), _lib.dptr(o_nu), _lib.dptr(o_be), _lib.dptr(o_ga),
                                               _lib.dptr(o_ph)))
        if whole:
            mu, nu, beta_i, gamma, PHI = o_mu, o_nu, o_be, o_ga, o_ph
        else:
            if mu is None:
                mu = np.zeros((ns, k)); nu = np.zeros((ns, k)); beta_i = np.zeros((ns, k)); gamma = np.zeros((ns, k))
                PHI = np.zeros((ns, m))
            mu[idx] = o_mu; nu[idx] = o_nu; beta_i[idx] = o_be; gamma[idx] = o_ga; PHI[idx] = o_ph
    if mu is None:                                                   # no rows
        mu = np.zeros((ns, k)); nu = np.zeros((ns, k)); beta_i = np.zeros((ns, k)); gamma = np.zeros((ns, k)); PHI = np.zeros((ns, m))
    sigma = nu + beta_i + gamma                                      # predict.m:72
    mu = mu + model.muY                                              # predict.m:73
    return mu, sigma, nu, beta_i, gamma, PHI, w, iS


def getPrior(X, Psi, theta, model, selection=None, device=0, return_iterations=False):
    """prior = getPrior(X,Sx,theta,model,set)   (getPrior.m:1); X / Psi already normalised as train.m passes them."""
    lib = _lib.load()
    X = np.asarray(X, dtype=np.float64)
    psi = None if Psi is None else np.asarray(Psi, dtype=np.float64)
    if selection is not None:
        sel = np.asarray(selection, dtype=bool)
        X = X[sel]
        if psi is not None:
            psi = psi[:, :, sel] if psi.ndim == 3 else psi[sel]
    X = _f64(X, 2)
    psi_kind = 0
    if psi is not None:
        psi = np.asfortranarray(psi)
        psi_kind = _psi_kind(model, psi)
    theta = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).ravel())
    prior = np.empty(model.m)
    it = C.c_int32()
    ds = _desc(model, device)
    _lib.check(lib.gpz_prior(C.byref(ds), _lib.dptr(theta), _lib.dptr(X), X.shape[0], _lib.dptr(psi), psi_kind,
                             _lib.dptr(prior), C.byref(it)))
    return (prior, int(it.value)) if return_iterations else prior
