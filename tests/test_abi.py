"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/gpz_hip.h
declares, fails loudly without a GPU (no CPU fallback), and the product package never touches the oracle."""
import os
import re

import numpy as np
import pytest

import gpz_amd
from gpz_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "gpz_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(gpz_[a-z_0-9]+)\s*\(", src))
    names.discard("gpz_allreduce_fn")
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gpz_hip.h but not exported"
    assert set(declared) == set(_lib.SYMBOLS), "ctypes table and header disagree"
    assert lib.gpz_version() == 3


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a box without a GPU")
def test_fails_loudly_without_gpu():
    with pytest.raises(_lib.GpzError):
        gpz_amd.Dxy(np.zeros((3, 2)), np.zeros((2, 2)))
    model = gpz_amd.Model(m=3, d=2)
    with pytest.raises(_lib.GpzError):
        gpz_amd.GPzContext(model, np.zeros((8, 2)), np.zeros((8, 1)))


def test_product_package_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gpz_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in txt.replace("no CPU fallback", ""), f"{fn} mentions the oracle"


def test_argument_validation_is_host_side():
    model = gpz_amd.Model(m=3, d=2)
    with pytest.raises(ValueError):
        gpz_amd.GPzContext(model, np.zeros((8, 3)), np.zeros((8, 1)))      # wrong d
    with pytest.raises(ValueError):
        gpz_amd.GPzContext(model, np.zeros((8, 2)), np.zeros((8, 1)), training=np.ones(5, bool))


def test_error_text_reaches_a_thread_that_never_failed_itself():
    """gpz_last_error() is per thread; a thread that has never failed itself gets the most recent failure of any thread (VERDICT r03
    weak 10: work that fails on a worker thread must not leave its caller with an empty message).  Argument errors need no GPU."""
    import ctypes
    import threading
    lib = _lib.load()
    lib.gpz_last_error.restype = ctypes.c_char_p
    seen = {}

    def failing():      # gpz_inv_logdet with null pointers: rejected before any HIP call
        rc = lib.gpz_inv_logdet(None, 4, 0, None, None, None)
        seen["rc"], seen["own"] = rc, lib.gpz_last_error().decode()

    def bystander():
        seen["other"] = lib.gpz_last_error().decode()

    for fn in (failing, bystander):
        t = threading.Thread(target=fn)
        t.start(); t.join()
    assert seen["rc"] < 0 and "gpz_inv_logdet" in seen["own"]
    assert seen["other"] == seen["own"]


def test_theta_layout_matches_reference():
    # g_dim per method (init.m:65-86) and p = m*d + g_dim + m*k + k + 2*m*k
    for method, g in (("GL", 1), ("VL", 7), ("GD", 3), ("VD", 21), ("GC", 9), ("VC", 63)):
        assert gpz_amd.Model(m=7, d=3, method=method).g_dim == g


def test_the_library_has_no_undefined_symbols_of_its_own():
    """every symbol the translation units of libgpz_hip.so expect from each other is defined in it (a lazily bound ctypes load would
    only trip over a missing one when the call is made - on the GPU box)"""
    import subprocess
    for name in ("libgpz_hip.so", "libgpz_hip_dev.so"):
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpz_amd", "lib", name)
        if not os.path.exists(path):
            assert name != "libgpz_hip.so"
            continue
        out = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
        own = [l for l in out.splitlines() if "gpz" in l.lower() or "launch_" in l]
        assert not own, own
