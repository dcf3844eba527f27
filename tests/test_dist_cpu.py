"""N>1 path on CPU: row sharding and the all-reduce hook (gloo, world_size 2).

Each rank computes the row-local partial sums of the first reduction buffer ([PHI'W PHI | PHI'W y | sums],
SURVEY.md §8e) for its shard — with the oracle's getPHI as the checker-side stand-in for the HIP kernels,
which need a GPU — pushes them through gpz_amd.dist's hook exactly as libgpz_hip.so does (raw pointer +
count), and the result must equal the unsharded sums."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpz_amd import dist as gdist
from oracle import gpz_oracle as O
from helpers import make_problem


def test_shard_bounds_partition():
    for n in (0, 1, 7, 64, 1000, 1001):
        for world in (1, 2, 3, 8):
            spans = [gdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans[:-1], spans[1:]):
                assert a[1] == b[0]


def test_shard_rows_masks():
    rng = np.random.default_rng(0)
    n = 101
    X = rng.standard_normal((n, 3)); Y = rng.standard_normal((n, 1)); om = rng.random((n, 1))
    tr = rng.random(n) < 0.6; va = ~tr & (rng.random(n) < 0.5)
    got_t, got_v = [], []
    for r in range(3):
        Xs, Ys, oms, trs, vas = gdist.shard_rows(r, 3, X, Y, om, tr, va)
        assert not (trs & vas).any()
        got_t.append(Xs[trs]); got_v.append(Xs[vas])
        assert np.array_equal(oms[trs], om[tr][gdist.shard_bounds(tr.sum(), r, 3)[0]:gdist.shard_bounds(tr.sum(), r, 3)[1]])
    assert np.array_equal(np.concatenate(got_t), X[tr])
    assert np.array_equal(np.concatenate(got_v), X[va])


def _partials(model, theta, X, Y, om):
    PHI, _, lnB = O.getPHI(X, None, theta, model)
    beta = np.exp(-lnB[:, 0])
    wb = beta * om[:, 0]
    S = (PHI * wb[:, None]).T @ PHI
    rhs = PHI.T @ (wb * Y[:, 0])
    return np.concatenate([S.ravel(), rhs, [om.sum(), (om[:, 0] * lnB[:, 0]).sum(), float(X.shape[0])]])


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, theta, X, Y, _, rng = make_problem(203, 3, 6, 1, "VC", True, seed=21)
    om = np.random.default_rng(5).random((203, 1)) + 0.5
    Xs, Ys, oms, trs, _ = gdist.shard_rows(rank, world, X, Y, om)
    buf = np.ascontiguousarray(_partials(model, theta, Xs[trs], Ys[trs], oms[trs]))
    hook = gdist.make_allreduce(device="cpu")
    rc = hook(None, buf.ctypes.data_as(C.c_void_p).value, buf.size, None)
    full = _partials(model, theta, X, Y, om)
    err = float(np.max(np.abs(buf - full)) / np.max(np.abs(full)))
    q.put((rank, rc, err))
    dist.destroy_process_group()


def test_allreduce_hook_gloo_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, rc, err in res:
        assert rc == 0 and err < 1e-13, (rank, rc, err)
