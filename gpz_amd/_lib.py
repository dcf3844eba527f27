"""ctypes binding of libgpz_hip.so (the C ABI declared in include/gpz_hip.h).

The product path has no CPU fallback: if the shared library is missing, or a call needs a GPU that is not
there, this module raises instead of computing anything on the host.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPZ_HIP_LIB") or os.path.join(_HERE, "lib", "libgpz_hip.so")   # override: kernel experiments

c_double_p = C.POINTER(C.c_double)
c_uint8_p = C.POINTER(C.c_uint8)
c_int32_p = C.POINTER(C.c_int32)


class gpz_desc(C.Structure):
    _fields_ = [
        ("d", C.c_int32),
        ("m", C.c_int32),
        ("k", C.c_int32),
        ("method", C.c_char * 4),
        ("heteroscedastic", C.c_int32),
        ("device", C.c_int32),
        ("stream", C.c_void_p),
        ("rank", C.c_int32),
        ("world", C.c_int32),
        ("dtype", C.c_int32),
        ("omega_cols", C.c_int32),
        ("reserved", C.c_int32 * 2),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

# every symbol include/gpz_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "gpz_ctx_create": (C.c_int, [C.POINTER(gpz_desc), C.c_int64, c_double_p, c_double_p, c_double_p, C.c_int32,
                                 c_double_p, c_uint8_p, c_uint8_p, C.POINTER(C.c_void_p)]),
    "gpz_ctx_create_sharded": (C.c_int, [C.POINTER(gpz_desc), C.c_int64, c_double_p, c_double_p, c_double_p, C.c_int32,
                                         c_double_p, c_uint8_p, c_uint8_p, c_uint8_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "gpz_ctx_destroy": (None, [C.c_void_p]),
    "gpz_ctx_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "gpz_theta_len": (C.c_int64, [C.c_void_p]),
    "gpz_theta_len_of": (C.c_int64, [C.POINTER(gpz_desc)]),
    "gpz_n_train": (C.c_int64, [C.c_void_p]),
    "gpz_n_valid": (C.c_int64, [C.c_void_p]),
    "gpz_eval": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "gpz_solve": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "gpz_get_phi": (C.c_int, [C.c_void_p, c_double_p]),
    "gpz_eval_dev": (C.c_int, [C.c_void_p, C.c_void_p, c_double_p, C.c_void_p, c_double_p, c_double_p]),
    "gpz_lbfgs_create": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gpz_lbfgs_destroy": (None, [C.c_void_p]),
    "gpz_lbfgs_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, c_int32_p]),
    "gpz_lbfgs_direction": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpz_lbfgs_last_error": (C.c_char_p, []),
    "gpz_vec_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, c_double_p]),
    "gpz_vec_axpy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "gpz_ctx_set_pinv_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "gpz_ctx_last_pinv": (C.c_int, [C.c_void_p, c_double_p]),
    "gpz_ctx_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "gpz_ctx_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), c_double_p, C.POINTER(C.c_int64), C.c_int]),
    "gpz_ctx_route": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "gpz_ctx_reset_timings": (C.c_int, [C.c_void_p]),
    "gpz_phi": (C.c_int, [C.POINTER(gpz_desc), c_double_p, c_double_p, C.c_int64, c_double_p, C.c_int32, c_double_p,
                          c_double_p, c_double_p]),
    "gpz_predict_full": (C.c_int, [C.POINTER(gpz_desc), c_double_p, c_double_p, c_double_p, c_double_p, C.c_int64,
                                   c_double_p, c_double_p, c_double_p, c_double_p]),
    "gpz_predict_noisy": (C.c_int, [C.POINTER(gpz_desc), c_double_p, c_double_p, c_double_p, c_double_p, C.c_int64,
                                    c_double_p, C.c_int32, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "gpz_predict_missing": (C.c_int, [C.POINTER(gpz_desc), c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                      C.c_int64, c_double_p, C.c_int32, c_double_p, c_double_p, c_double_p, c_double_p,
                                      c_double_p]),
    "gpz_prior": (C.c_int, [C.POINTER(gpz_desc), c_double_p, c_double_p, C.c_int64, c_double_p, C.c_int32, c_double_p,
                            c_int32_p]),
    "gpz_inv_logdet": (C.c_int, [c_double_p, C.c_int32, C.c_int32, c_double_p, c_double_p, c_int32_p]),
    "gpz_dxy": (C.c_int, [c_double_p, C.c_int64, c_double_p, C.c_int64, C.c_int32, C.c_int32, c_double_p]),
    "gpz_nan_groups": (C.c_int, [c_double_p, C.c_int64, C.c_int32, C.c_int32, c_int32_p, c_int32_p]),
    "gpz_mgpu_create": (C.c_int, [C.POINTER(gpz_desc), C.c_int32, c_int32_p, C.c_int32, C.c_int64, c_double_p, c_double_p,
                                  c_double_p, C.c_int32, c_double_p, c_uint8_p, c_uint8_p, C.POINTER(C.c_void_p)]),
    "gpz_mgpu_destroy": (None, [C.c_void_p]),
    "gpz_mgpu_eval": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "gpz_mgpu_solve": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "gpz_mgpu_size": (C.c_int32, [C.c_void_p]),
    "gpz_mgpu_alive": (C.c_int32, [C.c_void_p]),
    "gpz_mgpu_debug_fail_at": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "gpz_mgpu_theta_len": (C.c_int64, [C.c_void_p]),
    "gpz_mgpu_ctx": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "gpz_device_count": (C.c_int, []),
    "gpz_mgpu_predict": (C.c_int, [C.POINTER(gpz_desc), C.c_int32, c_int32_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                  C.c_int64, c_double_p, C.c_int32, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "gpz_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "gpz_ctx_init_rccl": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "gpz_rccl_origin": (C.c_char_p, []),
    "gpz_last_error": (C.c_char_p, []),
    "gpz_version": (C.c_int, []),
    "gpz_release_cached_memory": (None, []),
    "gpz_debug_fail_alloc": (None, [C.c_int64]),
    "gpz_ctx_comm_info": (C.c_int, [C.c_void_p, c_int32_p, C.c_char_p, C.c_int32]),
    "gpz_mgpu_comm_info": (C.c_int, [C.c_void_p, C.c_int32, c_int32_p, C.c_char_p, C.c_int32]),
}

_lib = None


class GpzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgpz_hip error {code}: {msg}")
        self.code = code


def load():
    """Load libgpz_hip.so and bind every declared symbol.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own HIP runtime.  If this library pulls in /opt/rocm's runtime first, a later
    # `import torch` finds "No HIP GPUs"; loaded in the other order both share one runtime.  torch is only plumbing
    # here (device buffers for the sharded / device-resident callers), so it is optional.
    if os.environ.get("GPZ_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with ./build.sh or __graft_entry__.build(); "
            "gpz_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().gpz_last_error()
        raise GpzError(rc, msg.decode() if msg else "")


def check_plain(rc):
    """Status check for the entry points that keep their own error string (gpz_lbfgs_*, gpz_vec_*)."""
    if rc != 0:
        msg = load().gpz_lbfgs_last_error()
        raise GpzError(rc, msg.decode() if msg else "")


def dptr(a):
    """double* of a contiguous float64 numpy array (None -> NULL)."""
    if a is None:
        return None
    return a.ctypes.data_as(c_double_p)
