"""Developer check: RCCL bound inside libgpz_hip.so (gpz_rccl_unique_id / gpz_ctx_init_rccl) on a one-rank communicator."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import gpz_amd
from gpz_amd import _lib
from helpers import make_problem
model, theta, X, Y, _, rng = make_problem(5000, 10, 96, 1, "VC", True, seed=41)
a = gpz_amd.GPzContext(model, X, Y, rank=0, world=2)
lib = _lib.load()
idbuf = C.create_string_buffer(128)
_lib.check(lib.gpz_rccl_unique_id(idbuf))
print("origin", lib.gpz_rccl_origin().decode(), flush=True)
os.system("grep -E 'rccl|amdhip' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
_lib.check(lib.gpz_ctx_init_rccl(a._h, idbuf, 0, 1, 0))
f, g = a.eval(theta)
b = gpz_amd.GPzContext(model, X, Y)
fb, gb = b.eval(theta)
print("identical", f == fb, np.array_equal(g, gb))
