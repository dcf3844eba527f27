"""gpz_amd — MI355X (gfx950) implementation of GPz's marginal-likelihood objective/gradient path.

The numeric work lives in ``lib/libgpz_hip.so`` (hand-written HIP, built by ``./build.sh`` or
``__graft_entry__.build()``); ``api`` mirrors the reference's MATLAB interface on top of its C ABI and
``dist`` shards rows across GPUs with an RCCL all-reduce of the m x m / m x d partials.
"""
from .api import (GPz, GPzContext, GPzMulti, device_count, rccl_origin, Model, getPHI, getPrior, inv_logdet, Dxy, nan_groups, predict, reset, globals_)  # noqa: F401
from . import dist  # noqa: F401
from .host import init, train, fixPsi, getOmega, sample, metrics, minfunc_lbfgs  # noqa: F401
