"""Row sharding across GPUs (one process per GPU) and the all-reduce hook of the C ABI.

The evaluation is data-parallel over samples: every n-indexed quantity of GPz.m is row-local and rows couple
only through sums.  Each rank owns a contiguous block of the training-selected rows; per
evaluation the library calls the hook twice, on [PHI'W PHI | PHI'W y | scalar sums] (m x m) and on
[dP/dGamma moments | PHI'(..) vectors | scalar sums] (m x (d^2+d)), and every rank finishes the m-sized work
redundantly so no broadcast is needed.  The hook is ``torch.distributed.all_reduce`` — RCCL over xGMI for
device buffers ("nccl" backend), gloo for the host-buffer tests.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


def shard_bounds(n, rank, world):
    """Contiguous block [lo, hi) of rank ``rank`` out of ``world`` over n rows: balanced blocks (sizes differ by at
    most one row), so no rank is left empty as long as n >= world."""
    return (rank * n) // world, ((rank + 1) * n) // world


def shard_index(rank, world, n, training=None, validation=None):
    """Row indices (into the unsharded arrays) owned by ``rank``: its contiguous block of the training-selected
    rows followed by its block of the validation rows; also returns how many of them are training rows."""
    tr = np.ones(n, dtype=bool) if training is None else np.asarray(training, dtype=bool).ravel()
    idx_t = np.flatnonzero(tr)
    if idx_t.size < world:
        raise ValueError(f"{idx_t.size} training rows cannot be sharded over {world} ranks: every rank needs at least one")
    lo, hi = shard_bounds(idx_t.size, rank, world)
    keep = [idx_t[lo:hi]]
    if validation is not None:
        idx_v = np.flatnonzero(np.asarray(validation, dtype=bool).ravel())
        lv, hv = shard_bounds(idx_v.size, rank, world)
        keep.append(idx_v[lv:hv])
    return np.concatenate(keep), hi - lo


def shard_rows(rank, world, X, Y, omega=None, training=None, validation=None):
    """Split the TRAINING-selected rows (and, independently, the validation rows) into contiguous blocks and
    return this rank's (X, Y, omega, training, validation) slices.  Rows selected by neither mask are dropped:
    the path never reads them (getPHI.m:14)."""
    X = np.asarray(X)
    Y = np.asarray(Y)
    rows, nt = shard_index(rank, world, X.shape[0], training, validation)
    Xs, Ys = X[rows], Y[rows]
    oms = None if omega is None else np.asarray(omega)[rows]
    trs = np.zeros(rows.size, dtype=bool)
    trs[:nt] = True
    vas = None
    if validation is not None:
        vas = np.zeros(rows.size, dtype=bool)
        vas[nt:] = True
    return Xs, Ys, oms, trs, vas


def shard_psi(rank, world, Psi, training=None, validation=None):
    """This rank's slice of the input-noise array, matching shard_rows: n x d rows for the diagonal kinds,
    d x d x n slices for GC/VC (fixPsi.m:22-53)."""
    if Psi is None:
        return None
    Psi = np.asarray(Psi)
    cube = Psi.ndim == 3
    rows, _ = shard_index(rank, world, Psi.shape[2] if cube else Psi.shape[0], training, validation)
    return np.ascontiguousarray(Psi[:, :, rows]) if cube else Psi[rows]


def nan_patterns(X, training=None, validation=None):
    """NaN-pattern table (G x d bool, True = missing) of the rows the evaluation reads, in first-occurrence order —
    the groups of getPHI.m:43-54 on the unsharded data.  Every rank passes the same table to GPzContext(patterns=...)."""
    X = np.asarray(X)
    n = X.shape[0]
    keep = np.ones(n, dtype=bool) if training is None else np.asarray(training, dtype=bool).ravel().copy()
    if training is not None and validation is not None:
        keep |= np.asarray(validation, dtype=bool).ravel()
    miss = np.isnan(X[keep])
    _, first = np.unique(miss, axis=0, return_index=True)
    return miss[np.sort(first)]


class _CudaBuf:
    """Minimal __cuda_array_interface__ view of a raw device pointer (float64, 1-D)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def init_rccl(ctx, rank, world, device, group=None):
    """Give a sharded context its own RCCL communicator INSIDE the library (gpz_rccl_unique_id / gpz_ctx_init_rccl): the
    two all-reduces of an evaluation are then ncclAllReduce calls issued by the library on its own stream — no Python,
    no GIL, no ctypes trampoline in the evaluation.  torch.distributed is used once, to ship rank 0's 128-byte id."""
    import torch.distributed as dist
    from . import _lib
    lib = _lib.load()
    idbuf = C.create_string_buffer(128)
    box = [None]
    if rank == 0:
        # a failure on rank 0 (no RCCL to bind) must reach every rank through the broadcast, or they wait in it for ever
        try:
            _lib.check(lib.gpz_rccl_unique_id(idbuf))
            box = [idbuf.raw]
        except Exception as e:
            box = [repr(e)]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    if not isinstance(box[0], bytes):
        raise RuntimeError(f"gpz_rccl_unique_id failed on rank 0: {box[0]}")
    idbuf = C.create_string_buffer(box[0], 128)
    _lib.check(lib.gpz_ctx_init_rccl(ctx._h, idbuf, int(rank), int(world), int(device)))
    ctx._cb = None            # the Python hook (if any) is no longer referenced by the library
    return lib.gpz_rccl_origin().decode()


def make_allreduce(group=None, device="cuda", device_index=None):
    """Build the C-ABI all-reduce hook (gpz_allreduce_fn) on top of torch.distributed (the alternative to init_rccl:
    any backend torch.distributed offers, e.g. gloo for the CPU tests and for two ranks that share one GPU).

    device "cuda": ``buf`` is a device pointer; the tensor view is all-reduced with the process group's backend on the
    stream the library passes (the context's stream), so the collective is ordered with the library's kernels whatever
    torch's current stream is.  device "cpu": ``buf`` is a host pointer (used by the gloo tests).
    The hook object owns its tensor views; make one hook per context and drop it with the context."""
    import torch
    import torch.distributed as dist
    views = {}   # the library all-reduces the same two buffers every evaluation: build each tensor view once

    def hook(user, buf, count, stream):
        try:
            t = views.get((buf, count))
            if t is None:
                if device == "cpu":
                    arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(count,))
                    t = torch.from_numpy(arr)
                else:
                    idx = torch.cuda.current_device() if device_index is None else device_index
                    t = torch.as_tensor(_CudaBuf(buf, count), device=torch.device("cuda", idx))
                    if t.data_ptr() != buf:
                        raise RuntimeError("device buffer was copied instead of viewed: wrong device index for the hook")
                views[(buf, count)] = t
            if device != "cpu" and stream:
                with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as e:  # never unwind through the C frame
            print("gpz_amd.dist allreduce hook failed:", repr(e), flush=True)
            return 1

    return hook
