"""bench.py's launch forms: --gpus N must measure N devices or fail — never fewer, silently."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=e,
                          timeout=timeout, cwd=ROOT)


def line_of(proc):
    assert proc.returncode == 0, proc.stderr[-2000:]
    return json.loads([l for l in proc.stdout.splitlines() if l.startswith("{")][-1])


def test_more_gpus_than_the_box_has_is_an_error_not_a_smaller_run():
    """Runs everywhere: without a GPU bench.py refuses outright; on a GPU box with fewer than 64 devices `--gpus 64` must exit
    non-zero before any work instead of printing an n_gpus: 1 line."""
    p = run_bench("--gpus", "64", "--rows", "4000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert p.returncode != 0
    assert not any(l.startswith("{") for l in p.stdout.splitlines())


@pytest.mark.gpu
def test_gpus_2_on_a_one_gpu_box_exits_non_zero():
    import gpz_amd
    if gpz_amd.device_count() >= 2:
        pytest.skip("needs a box with exactly one GPU")
    p = run_bench("--gpus", "2", "--rows", "4000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout)
    # a launcher whose world size is not --gpus is refused as well
    p = run_bench("--gpus", "1", "--rows", "4000", "--steps", "2", "--no-cpu-baseline",
                  env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_gpus_1_native_mgpu_1_and_plain_are_bitwise_identical():
    common = ["--config", "c2", "--rows", "6000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    plain = line_of(run_bench(*common))
    g1 = line_of(run_bench("--gpus", "1", *common))
    nm1 = line_of(run_bench("--native-mgpu", "1", *common))
    assert plain["n_gpus"] == g1["n_gpus"] == nm1["n_gpus"] == 1
    assert plain["check"] == g1["check"] == nm1["check"]
    # (c2 has m + 1 <= 256 columns: the T-GEMM, the row scalars and the moment sums are the one kernel k_small_tail)
    assert nm1["per_rank"][0]["rows"] == 6000 and "tail_small" in nm1["per_rank"][0]["stage_ms_per_eval"]


@pytest.mark.gpu
def test_gpus_n_runs_the_in_library_rccl_driver_when_the_node_has_the_devices():
    """On a multi-GPU node: `python bench.py --gpus 2` (no launcher) = gpz_mgpu_* with the RCCL reducer, n_gpus from gpz_mgpu_size,
    same f as the one-GPU run up to the regrouped sums."""
    import gpz_amd
    if gpz_amd.device_count() < 2:
        pytest.skip("needs two GPUs")
    common = ["--config", "c2", "--rows", "6000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = line_of(run_bench(*common))
    two = line_of(run_bench("--gpus", "2", *common))
    assert two["n_gpus"] == 2 and len(two["per_rank"]) == 2 and two["config"]["rccl"]
    assert "allreduce1" in two["per_rank"][1]["stage_ms_per_eval"]
    f1, f2 = float.fromhex(one["check"]["f_last"]), float.fromhex(two["check"]["f_last"])
    assert abs(f1 - f2) <= 1e-10 * abs(f1)


@pytest.mark.gpu
def test_injected_rank_failure_is_reported_and_does_not_hang():
    """a rank that fails at an exchange point must not leave the others waiting.  Loopback reducer (one GPU): the
    barrier is poisoned, the call returns the rank's error, the handle stays usable.  RCCL reducer (needs 2 GPUs): the
    communicators are aborted, the call returns, the handle is dead and says so."""
    import gpz_amd
    from gpz_amd import _lib
    from helpers import make_problem
    model, theta, X, Y, _, rng = make_problem(900, 3, 6, 1, "VD", True, seed=5)
    mg = gpz_amd.GPzMulti(model, X, Y, n_gpus=3, reducer="loopback")
    f0, g0 = mg.eval(theta)
    for rank, exch in ((1, 1), (2, 2), (0, 1)):
        mg.debug_fail_at(rank, exch)
        with pytest.raises(_lib.GpzError) as ei:
            mg.eval(theta)
        assert f"rank {rank}" in str(ei.value)
        assert mg.alive
        f1, g1 = mg.eval(theta)                       # fires once; the handle recovers
        assert f1 == f0 and np.array_equal(g0, g1)
    mg.close()
    if gpz_amd.device_count() >= 2:
        mg = gpz_amd.GPzMulti(model, X, Y, n_gpus=2, reducer="rccl")
        f0, _ = mg.eval(theta)
        mg.debug_fail_at(1, 2)
        with pytest.raises(_lib.GpzError):
            mg.eval(theta)
        assert not mg.alive
        with pytest.raises(_lib.GpzError) as ei:
            mg.eval(theta)
        assert ei.value.code == -4 and "dead" in str(ei.value)
        mg.close()


# ---- the first multi-GPU line cannot pass unverified ---------------------------------------------------------
def _verdict_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = bench.rccl_init_verdict(dist, torch, None, rank)                                    # every rank has its communicator
    bad = bench.rccl_init_verdict(dist, torch, "RuntimeError('ncclCommInitRank failed')" if rank == 1 else None, rank)
    q.put((rank, ok, bad))
    dist.destroy_process_group()


def test_a_failed_in_library_rccl_init_is_fatal_on_every_rank():
    """bench.py under a launcher: when gpz_ctx_init_rccl fails on ANY rank, every rank learns it and the run dies with a message — it
    used to continue on the torch.distributed hook with a line on stderr, and the self-verification was then skipped.  World-2 gloo on
    CPU: rank 1 reports the failure, both ranks get the verdict; no failure -> None on both."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_verdict_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, bad in got:
        assert ok is None
        assert bad and "refusing to continue" in bad and "GPZ_BENCH_COMM=torch" in bad
    assert "ncclCommInitRank failed" in got[1][2] and "another rank failed" in got[0][2]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "falling back to the torch.distributed hook" not in src            # the silent route change is gone
    assert 'raise SystemExit("bench.py: " + why)' in src


def test_rank_records_of_either_route_must_describe_n_ranks_on_n_devices():
    """ranks_describe_n_devices on the records bench.py gathers: in-library RCCL communicators and torch.distributed process groups
    alike must show N ranks, ranks 0..N-1 once each, N distinct PCI devices; anything else is the reason the run exits with."""
    sys.path.insert(0, ROOT)
    import bench
    good = [{"nccl_count": 4, "nccl_rank": r, "pci_bus_id": "0000:%02x:00.0" % (5 + r)} for r in range(4)]
    assert bench.ranks_describe_n_devices(good, 4) is None
    hook = [dict(q, comm_record="torch.distributed process group (nccl)") for q in good]
    assert bench.ranks_describe_n_devices(hook, 4) is None
    same_dev = [dict(q, pci_bus_id="0000:05:00.0") for q in good]
    assert "PCI bus ids" in bench.ranks_describe_n_devices(same_dev, 4)
    assert bench.ranks_describe_n_devices(good[:3], 4) and bench.ranks_describe_n_devices(good, 8)
    no_comm = [dict(q, nccl_count=-1, nccl_rank=-1) for q in good]             # a rank without any communicator record
    assert bench.ranks_describe_n_devices(no_comm, 4)
    dup = [dict(q, nccl_rank=0) for q in good]
    assert bench.ranks_describe_n_devices(dup, 4)
