#!/usr/bin/env python
"""Benchmark of the GPz objective+gradient evaluation on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c4|c3|c2]

A "step" is one gpz_eval: theta on the host -> [f, grad, 4 statistics] on the host, data resident in HBM.
Metric (BASELINE.json): objective+gradient evaluations per second, whole job.  N > 1 shards the n rows of the SAME
problem (strong scaling); only the m x m / m x (d^2+d) partials are all-reduced over RCCL.  Two launch forms:
  * `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`: one rank per process, RCCL communicator
    created INSIDE the library from a unique id shipped once by torch.distributed (gpz_ctx_init_rccl);
  * `python bench.py --gpus N` (no launcher, WORLD_SIZE unset): ONE process, the library drives the N devices itself
    (gpz_mgpu_*: a host thread + stream + context per device, ncclCommInitAll) — the form a MATLAB host uses
    (minFunc.m:314 calls funObj once and waits).
Either way the run FAILS (non-zero exit) when the node has fewer than N GPUs, when the launcher's world size is not N, when the
in-library RCCL communicator cannot be created (unless GPZ_BENCH_COMM=torch ASKED for the torch.distributed hook), or when the ranks'
own records do not describe N ranks on N PCI devices; nothing degrades silently to fewer devices or to another route.  `--native-mgpu K` is the separate single-GPU test form: K shards on one
device with the library's loopback reducer (measures the driver's threading, not scaling; reports n_gpus = 1).

Workloads (synthetic data per SURVEY.md §8d):
    c4  n=1e6 d=10 m=1000 VC heteroscedastic fp64   <- the configuration the metric's target is quoted on (default)
    c3  n=1e5 d=10 m=500  VC heteroscedastic + cost-sensitive omega
    c2  n=1e5 d=10 m=200  VD heteroscedastic
    c5  n=2e6 d=20 m=2000 VC heteroscedastic + input noise Psi (diagonal cubes), dtype f32 (fp32 per-pair factorisations)
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c4": dict(n=1_000_000, d=10, m=1000, method="VC", omega=None),
    "c3": dict(n=100_000, d=10, m=500, method="VC", omega="normalized"),
    "c2": dict(n=100_000, d=10, m=200, method="VD", omega=None),
    "c5": dict(n=2_000_000, d=20, m=2000, method="VC", omega=None, psi=True, dtype="f32"),
    # config 5 in the reference's own precision (not a BASELINE line: what the fp32 route is measured against): the fp64 pair
    # kernels of k_cpsi4.hip, f64 MFMA contractions
    "c5_f64": dict(n=2_000_000, d=20, m=2000, method="VC", omega=None, psi=True, dtype="f64"),
}
F64_MFMA_PEAK_TFLOPS = 78.6    # 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz (datasheet f64 matrix = f64 vector rate)
F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
F32_MFMA_UBENCH_TFLOPS = 155.0  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2 microbenchmark ceiling
F64_MFMA_UBENCH_TFLOPS = 77.7  # tools/mfma_f64_bench.hip on this pool's MI355X: back-to-back v_mfma_f64_16x16x4_f64, >=3 waves/SIMD


def pmc_traffic(config):
    """HBM-side bytes per launch of the dominant kernel, from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE).  PMC counters cannot be read inside this process, so
    the figure comes from the profile of the same command; None when no profile exists for the configuration."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_constants.json")) as fh:
            return json.load(fh).get(config)
    except Exception:
        return None


def ranks_describe_n_devices(per_rank, n):
    """A real N-GPU run proves itself from its own output: every rank's communicator must report N ranks (ncclCommCount; with the
    torch.distributed hook: the size of the nccl process group), the ranks 0 .. N-1 must appear once each (ncclCommUserRank / the
    group's rank), on N different PCI devices.  -> None, or the reason as text."""
    counts = sorted({q.get("nccl_count") for q in per_rank})
    ranks_seen = sorted(q.get("nccl_rank") for q in per_rank)
    buses = [q.get("pci_bus_id") for q in per_rank]
    if len(per_rank) != n or counts != [n] or ranks_seen != list(range(n)) or len(set(buses)) != n or "" in buses or None in buses:
        return (f"the communicators do not describe {n} ranks on {n} devices: {len(per_rank)} rank records, ncclCommCount {counts}, "
                f"ncclCommUserRank {ranks_seen}, PCI bus ids {buses}")
    return None


def rccl_init_verdict(dist, torch, err, rank):
    """Every rank reports whether gpz_ctx_init_rccl worked (err = None) and all of them learn the outcome: -> None when every rank has
    its communicator, else the text the run dies with (on every rank: nobody is left waiting in a collective)."""
    flag = torch.tensor([0 if err else 1], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return None
    return (f"rank {rank}: the RCCL communicator inside the library could not be created on every rank"
            + (f" (this rank: {err})" if err else " (another rank failed)")
            + "; refusing to continue on another route - GPZ_BENCH_COMM=torch selects the torch.distributed hook explicitly")


def synth(cfg, n=None):
    """Synthetic problem of SURVEY.md §8d: default_rng(1) data, default_rng(2) theta perturbation."""
    from gpz_amd.api import Model
    n = n or cfg["n"]
    d, m, method = cfg["d"], cfg["m"], cfg["method"]
    rng = np.random.default_rng(1)
    X = rng.standard_normal((n, d))
    a = rng.standard_normal(d) / math.sqrt(d)
    c = rng.standard_normal(d) / math.sqrt(d)
    y = np.sin(X @ a) + 0.1 * (1.0 + np.abs(X @ c)) * rng.standard_normal(n)
    y = (y - y.mean())[:, None]
    omega = None
    if cfg["omega"] == "normalized":
        omega = ((1.0 + y - y.min()) ** -2.0)           # getOmega.m:19 on a shifted target
    model = Model(m=m, d=d, k=1, method=method, heteroscedastic=True)
    P = X[rng.choice(n, m, replace=False)] + 0.1 * rng.standard_normal((m, d))
    sub = X[rng.choice(n, min(n, 20000), replace=False)]  # init.m:62 heuristic on a row subsample
    D = ((sub ** 2).sum(1)[:, None] + (P ** 2).sum(1)[None, :] - 2.0 * sub @ P.T)
    gam = np.sqrt(0.5 * m ** (1.0 / d) / np.mean(np.abs(D), axis=0))
    if method == "VC":
        G = np.zeros((d, d, m))
        for j in range(m):
            G[:, :, j] = np.eye(d) * gam[j]
    else:
        G = np.tile(gam[:, None], (1, d))
    vy = float(np.var(y, ddof=1))
    theta = np.concatenate([P.ravel(order="F"), G.ravel(order="F"), np.full(m, -math.log(vy)), [math.log(vy)],
                            0.01 * rng.standard_normal(m), np.zeros(m)])
    theta = theta + 0.05 * np.random.default_rng(2).standard_normal(theta.size)
    if cfg.get("psi") and method == "VC" and cfg.get("gamma", "relative") == "relative":
        # With input noise the precision matrices are drawn as gamma_j (I + 0.3 N(0,1)/sqrt(d)): cond(Gamma_j'Gamma_j) of a few units.
        # The absolute 0.05 N(0,1) perturbation above is larger than gamma_j itself at d = 20 (cond up to 7e9), and there the REFERENCE's
        # dGamma chain through inv(Gamma_j'Gamma_j) (GPz.m:174-180) is rounding noise, so no parity statement about those blocks can be
        # made against it (cfg["gamma"] = "absolute" keeps that theta for the robustness tests).
        r3 = np.random.default_rng(3)
        g0 = m * d
        for j in range(m):
            Gj = gam[j] * (np.eye(d) + 0.3 * r3.standard_normal((d, d)) / math.sqrt(d))
            theta[g0 + j * d * d:g0 + (j + 1) * d * d] = Gj.ravel(order="F")
    return model, theta, X, y, omega


def synth_psi(cfg, rows, cube=False):
    """Input noise of config 5: Gamma(1, 0.5) variances per dimension for the given row indices.  The library takes
    the n x d variances directly (psi_kind 3 of the C ABI: it builds the diagonal d x d x n cubes of fixPsi.m:27-31 on its side, so the
    6.4 GB cube of n = 2e6 rows is never materialised on the host); cube=True returns the d x d x n cubes themselves, the layout
    the oracle takes."""
    d = cfg["d"]
    var = np.random.default_rng(4).gamma(1.0, 0.5, (cfg["n"], d))[rows]
    if not cube:
        return np.asfortranarray(var)
    Psi = np.zeros((d, d, var.shape[0]))
    Psi[np.arange(d), np.arange(d), :] = var.T
    return Psi


def cpu_baseline(cfg, model, theta, X, y, omega, rows):
    """The oracle (as-written NumPy restatement of the reference path) timed on a bounded row sample."""
    from oracle import gpz_oracle as O
    om = Omodel = None
    Omodel = O.Model(m=model.m, d=model.d, k=1, method=model.method, heteroscedastic=True)
    Xs, ys = X[:rows], y[:rows]
    oms = None if omega is None else omega[:rows]
    psis = synth_psi(cfg, np.arange(rows), cube=True) if cfg.get("psi") else None
    # BLAS threads = the cores this process may actually use (affinity mask and cgroup quota): the default pool is sized
    # by the machine's core count, and on a quota-limited box that oversubscription halves the dgemm rate
    # (GPU box: 256 logical CPUs, quota 16 -> 714 GFLOP/s with the default 64 threads, 1406 with 16)
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        import threadpoolctl
        limiter = threadpoolctl.threadpool_limits(limits=usable, user_api="blas")
    except Exception:
        import contextlib
        limiter = contextlib.nullcontext()
    with limiter:
        t0 = time.perf_counter()
        ref = O.GPz(theta, Omodel, Xs, ys, psis, oms)
        t_all = time.perf_counter() - t0
        # the m^3 part does not scale with n: time it alone
        S = np.eye(model.m) + np.ones((model.m, model.m)) * 1e-3
        t0 = time.perf_counter()
        O.inv_logdet(S)
        t_svd = time.perf_counter() - t0
        try:
            import threadpoolctl
            nthreads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info() if p.get("user_api") == "blas"] + [1])
        except Exception:
            nthreads = usable
        vec = None
        if not cfg.get("psi"):
            # mode (ii) of BASELINE.md section 5: the same mathematics written the BLAS-3 way (ln PHI and the dP / dGamma sums as
            # GEMMs over the monomials of a row, dsyrk + one n m^2 product, Cholesky inverse) - oracle/gpz_vectorised.py
            from oracle import gpz_vectorised as V
            t0 = time.perf_counter()
            fv, gv = V.GPz(theta, Omodel, Xs, ys, oms)
            t_vec = time.perf_counter() - t0
            Sv = S.copy()
            t0 = time.perf_counter()
            import scipy.linalg as sla
            sla.cho_solve(sla.cho_factor(Sv, lower=True), np.eye(model.m))
            t_chol = time.perf_counter() - t0
            vec = dict(value=1.0 / ((t_vec - t_chol) * cfg["n"] / rows + t_chol), unit="evals/s",
                       agrees_with_as_written=bool(abs(fv - ref.nlogML) <= 1e-9 * abs(ref.nlogML)
                                                   and np.max(np.abs(gv - ref.grad)) <= 1e-6 * np.max(np.abs(ref.grad))),
                       sample=f"oracle/gpz_vectorised.py on the same {rows} rows: {t_vec:.1f} s measured (of which "
                              f"{t_chol:.2f} s is the m^3 Cholesky inverse), row-dependent part scaled x{cfg['n'] / rows:.1f}")
    scale = cfg["n"] / rows
    t_full = (t_all - t_svd) * scale + t_svd
    blas = "unknown"
    try:
        import threadpoolctl
        for p_ in threadpoolctl.threadpool_info():
            if p_.get("user_api") == "blas":
                blas = f"{p_.get('internal_api')} {p_.get('version')} ({p_.get('threading_layer', '')}, {p_.get('architecture', '')})"
    except Exception:
        pass
    out = dict(value=1.0 / t_full, unit="evals/s", cores=int(nthreads), kind="port", mode="as-written",
               host_cpu_count=os.cpu_count(), usable_cores=usable, blas=blas,
               sample=f"oracle GPz() as-written on the first {rows} of {cfg['n']} rows: {t_all:.1f} s measured "
                      f"(of which {t_svd:.2f} s is the m^3 SVD inverse), row-dependent part scaled x{scale:.1f}")
    if vec:
        out["vectorised"] = vec
    return ref, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)       # >= 20 calls after 3 warm-ups
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--n", "--rows", dest="n", type=int, default=None, help="override the row count (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--native-mgpu", type=int, default=0, metavar="K",
                    help="single process, K shards behind gpz_mgpu_* (one host thread per shard, reduction inside the library): RCCL over "
                         "K GPUs when the node has them, else all K shards on GPU 0 with the loopback reducer (measures the driver's "
                         "threading, not scaling)")
    ap.add_argument("--timed-events", choices=["dominant", "none"], default="dominant",
                    help="HIP events in the timed region: around the dominant stages (graph segments with events between them; default) or "
                         "none (the whole evaluation one graph: the latency-bound configurations, where eight extra graph launches show)")
    ap.add_argument("--validation", type=float, default=0.0,
                    help="fraction of the rows turned into validation rows (GPz.m:239-261 priced inside the step; SURVEY 8d: c2 with 0.15)")
    args = ap.parse_args()

    import torch
    import gpz_amd
    from gpz_amd import dist as gdist

    launched = "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) >= 1 and "RANK" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # GPZ_DIST_BACKEND=gloo lets two ranks share one GPU (single-GPU boxes: RCCL refuses duplicate devices) - a test form
    backend = os.environ.get("GPZ_DIST_BACKEND", "nccl")
    ndev = gpz_amd.device_count()
    if launched and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a "
                         f"{world}-rank run as {args.gpus} GPUs")
    if args.gpus > ndev and not (launched and backend != "nccl"):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node has {ndev} GPU(s); refusing to run on fewer devices "
                         "(use --native-mgpu K for the single-GPU loopback form)")
    if args.native_mgpu > 0 and (launched and world > 1 or args.gpus > 1):
        raise SystemExit("--native-mgpu is the single-GPU loopback form: do not combine it with --gpus N or a launcher")
    native = (not launched or world == 1) and args.gpus > 1          # one process, N devices, RCCL inside the library
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    use_dist = launched and world > 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = dict(CONFIGS[args.config])
    if args.n:
        cfg["n"] = args.n
    model, theta0, X, y, omega = synth(cfg)
    n = cfg["n"]
    stream = torch.cuda.current_stream().cuda_stream
    dtype = cfg.get("dtype", "f64")
    comm = "none"
    rccl_origin = None
    tr_mask = va_mask = None
    if args.validation > 0.0:
        if use_dist or native:
            raise SystemExit("--validation is a single-GPU measurement")
        va_mask = np.random.default_rng(6).random(n) < args.validation
        tr_mask = ~va_mask
    if use_dist:
        Xs, ys, oms, trs, _ = gdist.shard_rows(rank, world, X, y, omega)
        rows, _ = gdist.shard_index(rank, world, n)
        psi = synth_psi(cfg, rows) if cfg.get("psi") else None
        # The two all-reduces of an evaluation: RCCL called INSIDE the library on its own communicator (gpz_ctx_init_rccl; no
        # Python between the kernels) - GPZ_BENCH_COMM=torch selects the torch.distributed hook instead (any backend, e.g.
        # gloo for two ranks sharing one GPU).
        comm = os.environ.get("GPZ_BENCH_COMM", "rccl" if backend == "nccl" else "torch")
        if comm not in ("rccl", "torch"):
            raise SystemExit("GPZ_BENCH_COMM must be rccl or torch")
        ctx = gpz_amd.GPzContext(model, Xs, ys, psi, oms, trs, None, device=local_rank, stream=stream or None,
                                 rank=rank, world=world, allreduce=gdist.make_allreduce() if comm == "torch" else None, dtype=dtype)
        if comm != "torch":
            # FATAL when it fails: a line that silently took another route than the one it names is worth nothing.  The torch.distributed
            # hook is a measured route of its own, selected by GPZ_BENCH_COMM=torch, never a fallback.
            err = None
            try:
                rccl_origin = gdist.init_rccl(ctx, rank, world, local_rank)
            except Exception as e:
                err = repr(e)
            why = rccl_init_verdict(dist, torch, err, rank)
            if why:
                ctx.close()
                dist.destroy_process_group()
                raise SystemExit("bench.py: " + why)
    elif native or args.native_mgpu > 0:
        psi = synth_psi(cfg, np.arange(n)) if cfg.get("psi") else None
        K = args.gpus if native else args.native_mgpu
        reducer = "rccl" if native else "loopback"
        ctx = gpz_amd.GPzMulti(model, X, y, psi, omega, tr_mask, va_mask, n_gpus=K, reducer=reducer, dtype=dtype)
        if native and ctx.n_gpus != args.gpus:
            raise SystemExit(f"bench.py: asked for {args.gpus} GPUs, the library reports {ctx.n_gpus}")
        comm = f"gpz_mgpu x{K} ({reducer})"
        if native:
            rccl_origin = gpz_amd.rccl_origin()
    else:
        psi = synth_psi(cfg, np.arange(n)) if cfg.get("psi") else None
        ctx = gpz_amd.GPzContext(model, X, y, psi, omega, tr_mask, va_mask, device=local_rank, stream=stream or None, dtype=dtype)
    del psi
    multi = native or args.native_mgpu > 0
    n_local = ctx.rows_per_gpu[0] if multi else ctx.n_train
    n_gpus_used = ctx.n_gpus if native else world

    prng = np.random.default_rng(3)
    thetas = [theta0 + 1e-3 * prng.standard_normal(theta0.size) for _ in range(args.steps + args.warmup)]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region runs what a caller gets: from the third evaluation on, hipGraph replays (cut into segments at the two exchange
    # points of a sharded run).  Timing level 2 keeps that and records HIP events around the dominant stages only, BETWEEN the graph
    # launches (events inside a graph cannot be timed on ROCm 7: tools/graph_event_probe.hip); the all-stage breakdown comes from a
    # second pass of the same K steps with events around every stage, which runs as eager launches ("stage_pass").
    ctx.enable_timing(2 if args.timed_events == "dominant" else 0)
    for i in range(args.warmup):      # (the graph is recorded on the second evaluation: W >= 2 keeps the recording out of the timed region)
        ctx.eval(thetas[i])
    ctx.reset_timings()
    barrier()
    t0 = time.perf_counter()
    fs, step_s = [], []
    for i in range(args.steps):
        ts = time.perf_counter()
        f, g = ctx.eval(thetas[args.warmup + i])     # synchronous: returns with f, g on the host
        step_s.append(time.perf_counter() - ts)
        fs.append(f)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tim = ctx.timings()
    route = ctx.route() if hasattr(ctx, "route") else None
    rank_routes = [ctx.route(r) for r in range(ctx.n_gpus)] if multi else None
    # Second pass, events around EVERY stage (eager launches): the per-stage breakdown (allreduce1 / allreduce2 are the two exchange points)
    ctx.enable_timing(1)
    ctx.reset_timings()
    barrier()
    ts0 = time.perf_counter()
    for i in range(args.steps):
        ctx.eval(thetas[args.warmup + i])
    barrier()
    ts1 = time.perf_counter() - ts0
    tim_full = ctx.timings()
    if multi:
        rank_stage_t = {r: ctx.timings(r) for r in range(ctx.n_gpus)}   # (the gap pass below resets the timers)
    if not tim:            # --timed-events none: the kernel times of the roofline come from the stage pass
        tim = tim_full
    stage_pass = {"ms_per_step": ts1 / args.steps * 1e3, "evals_per_s": args.steps / ts1,
                  "note": "same K steps with HIP events around every stage: eager launches, not replay",
                  "stage_ms_per_eval": {k: v[0] / args.steps for k, v in tim_full.items()}}
    # The same K steps once more WITHOUT any event: the whole evaluation one graph launch - what a default caller gets (the
    # timed region above is cut into segments with events between them, which shows on the latency-bound configurations).  Reported
    # beside the headline, never instead of it: the roofline's kernel durations must come from the timed region itself.
    single_graph = None
    if args.timed_events == "dominant":
        ctx.enable_timing(0)
        for i in range(max(2, args.warmup)):
            ctx.eval(thetas[i % len(thetas)])
        barrier()
        tq0 = time.perf_counter()
        for i in range(args.steps):
            ctx.eval(thetas[args.warmup + i])
        barrier()
        tq1 = time.perf_counter() - tq0
        single_graph = {"ms_per_step": tq1 / args.steps * 1e3, "evals_per_s": args.steps / tq1,
                        "note": "same K steps, no HIP events: the evaluation as ONE hipGraph launch (the default caller's path)"}
    # Third pass (<= 5 steps), timing level 3: the replay of the timed region with events around EVERY graph segment and around the
    # all-reduce hooks.  Wall time of a call minus the sum of all of them = what the device spent between segments (launch-to-launch gaps,
    # waiting for the host, the result copy's latency): the figure a real N-GPU run is read against the single-device loopback line with.
    gsteps = min(args.steps, 5)
    ctx.enable_timing(3)
    for i in range(2):
        ctx.eval(thetas[i])
    ctx.reset_timings()
    barrier()
    tg0 = time.perf_counter()
    for i in range(gsteps):
        ctx.eval(thetas[args.warmup + i])
    wall_gap = (time.perf_counter() - tg0) / gsteps * 1e3
    barrier()

    def gap_record(t):
        seg = sum(v[0] for v in t.values()) / gsteps
        return {"wall_ms_per_eval": wall_gap, "segments_ms_per_eval": seg - t.get("exchange", (0.0, 0))[0] / gsteps,
                "exchange_ms_per_eval": t.get("exchange", (0.0, 0))[0] / gsteps, "gap_ms_per_eval": wall_gap - seg,
                "note": f"{gsteps} replayed evaluations, HIP events around every graph segment and every all-reduce; gap = wall - segments - exchange"}
    gaps = gap_record(ctx.timings()) if not multi else None
    per_rank = None
    if multi:
        per_rank = [dict({"rank": r, "rows": ctx.rows_per_gpu[r], "rccl": rccl_origin, "route": rank_routes[r],
                          "stage_ms_per_eval": {k: v[0] / args.steps for k, v in rank_stage_t[r].items()},
                          "gaps": gap_record(ctx.timings(r))}, **ctx.comm_info(r))
                    for r in range(ctx.n_gpus)]
    elif use_dist:
        # per rank: its rows, every stage and what the communicator itself reports (ncclCommCount / ncclCommUserRank / device + PCI bus
        # id) - the first real multi-GPU line proves N ranks on N devices from its own output and can be read stage by stage against
        # the single-GPU shard line (profiles/*_shard125k.json)
        mine = dict({"rank": rank, "rows": n_local, "comm": comm, "rccl": rccl_origin, "route": route,
                     "stage_ms_per_eval": stage_pass["stage_ms_per_eval"], "gaps": gaps}, **ctx.comm_info())
        if comm == "torch":
            # no communicator inside the library: what the process group behind the hook says about itself, and this rank's device
            mine.update({"nccl_count": dist.get_world_size(), "nccl_rank": dist.get_rank(), "comm_record": f"torch.distributed process group ({backend})"})
        per_rank = [None] * world if rank == 0 else None
        dist.gather_object(mine, per_rank, dst=0)
    else:
        per_rank_single = ctx.comm_info()
    if per_rank is not None and rank == 0 and (native or (use_dist and backend == "nccl")):
        # self-verification of EVERY real multi-GPU run, whichever route carries the all-reduce (the loopback / gloo test forms share one
        # device on purpose: not checked)
        why = ranks_describe_n_devices(per_rank, n_gpus_used)
        if why:
            raise SystemExit("bench.py: " + why)
    graph_pass = {"ms_per_step": elapsed / args.steps * 1e3, "evals_per_s": args.steps / elapsed, "route": route,
                  "note": "the timed region itself: hipGraph replay" + (", events around the dominant stages only" if args.timed_events == "dominant" else ", no events")}
    finite = bool(np.isfinite(fs).all() and np.isfinite(g).all())

    out = None
    if rank == 0:
        m = cfg["m"]
        ms_per_step = elapsed / args.steps * 1e3
        # few basis functions (m + k <= 256): the T-GEMM, the row scalars and the moment sums are ONE kernel (k_small_tail); its time is
        # priced against the same algorithmic flops 2 n m^2 of the product it contains
        small_tail = "tgemm" not in tim and "tail_small" in tim
        tg_ms, tg_calls = tim.get("tail_small" if small_tail else "tgemm", (0.0, 0))
        tg_avg = tg_ms / max(1, tg_calls)
        # rows one launch covers: all of the rank's rows, or one row tile when the context streams them (GPZ_ROW_TILE / PHI + T beyond the HBM)
        import re as _re
        _mt = _re.search(r"streamed, (\d+) tiles of (\d+)", route or "")
        ntiles = int(_mt.group(1)) if _mt else 1
        n_launch = n_local / ntiles
        flops_tg = 2.0 * n_launch * m * m             # algorithmic flops of PHI*inv(SIGMA) per launch (SURVEY §8d F_T)
        ach = flops_tg / (tg_avg * 1e-3) / 1e12 if tg_avg > 0 else 0.0
        sy_ms, sy_calls = tim.get("syrk", (0.0, 0))
        sy_avg = sy_ms / max(1, sy_calls)
        ach_sy = (n_launch * m * (m + 1.0)) / (sy_avg * 1e-3) / 1e12 if sy_avg > 0 else 0.0
        ph_ms, ph_calls = tim.get("phi_build", (0.0, 0))
        ph_avg = ph_ms / max(1, ph_calls)
        phi_gbs = 8.0 * (n_launch * cfg["d"] + n_launch * m) / (ph_avg * 1e-3) / 1e9 if ph_avg > 0 else 0.0
        f32_route = bool(cfg.get("psi")) and dtype != "f64"
        out = {
            "metric": "objective+gradient evals/sec", "value": args.steps / elapsed, "unit": "evals/s",
            "n_gpus": n_gpus_used, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "median_ms_per_step": float(np.median(step_s) * 1e3), "median_evals_per_s": float(1.0 / np.median(step_s)),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if dtype == "f64" else "f64 + f32 per-pair factorisations + f32-operand MFMA contractions (fp64 master sums)",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: n={n} d={cfg['d']} m={m} method={cfg['method']} heteroscedastic k=1"
                                   + (" omega=(1+y-min y)^-2" if cfg["omega"] else "")
                                   + (f" Psi=diag cubes (Gamma(1, 0.5) variances) dtype={dtype} Gamma_j=gamma_j(I+0.3 N(0,1)/sqrt(d))" if cfg.get("psi") else "")
                                   + (f" validation={int(va_mask.sum())} rows (training {n_local})" if va_mask is not None else ""),
                       "rows_per_gpu": n_local, "sharding": (f"rows/{world} + all-reduce of the m x m and m x (d^2+d) partials: "
                                                              + ("RCCL inside the library (gpz_ctx_init_rccl)" if comm != "torch"
                                                                 else f"torch.distributed hook ({backend})"))
                       if world > 1 else (f"one process, {comm}: rows/{ctx.n_gpus} per shard, one host thread per shard, "
                                          "reduction inside the library" if multi else "single GPU"),
                       "launch": ("torch.distributed.run, one rank per GPU" if use_dist else
                                  "one process, gpz_mgpu_* drives every device" if native else
                                  "one process, loopback shards on one device" if multi else "one process, one device"),
                       "rccl": rccl_origin},
            "roofline": {"bound": "mfma", "kernel": ("k_small_tail (T = PHI*[inv(SIGMA)|w] + row scalars + moment sums in one kernel, T in registers; "
                                                     "2*n*m^2 flops/launch; the stage time includes k_small_finish, the launch that sums its per-workgroup records)" if small_tail else
                                                     "k_tgemm (T = PHI*[inv(SIGMA)|w], 2*n*m^2 flops/launch)")
                                                    + (" on fp32-operand MFMAs" if f32_route else ""),
                         "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS if f32_route else F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / (F32_MFMA_PEAK_TFLOPS if f32_route else F64_MFMA_PEAK_TFLOPS),
                         "traffic": (pmc_traffic(args.config) or {}).get("tgemm_bytes_per_launch") if world == 1 and not multi and not args.n else None,
                         "traffic_source": (pmc_traffic(args.config) or {}).get("source") if world == 1 and not multi and not args.n else None,
                         "traffic_note": (pmc_traffic(args.config) or {}).get("note"),
                         "mfma_util_counters": (pmc_traffic(args.config) or {}).get("tgemm_mfma_util") if world == 1 and not multi and not args.n else None,
                         # the clock the kernel ran at in the committed PMC pass (GRBM cycles / kernel-trace duration) and the MFMA peak at THAT
                         # clock: flops-based frac x (2.4 / clock) is what the counter-based utilisation should be read against
                         "kernel_clock_ghz_counters": (pmc_traffic(args.config) or {}).get("tgemm_kernel_clock_ghz") if world == 1 and not multi and not args.n else None,
                         "avg_ms": tg_avg,
                         "ubench_ceiling": F32_MFMA_UBENCH_TFLOPS if f32_route else F64_MFMA_UBENCH_TFLOPS,
                         "frac_of_ubench": ach / (F32_MFMA_UBENCH_TFLOPS if f32_route else F64_MFMA_UBENCH_TFLOPS)},
            "kernels": {"syrk_tflops_algorithmic": ach_sy, "syrk_avg_ms": sy_avg,
                        "phi_build_GBs_algorithmic": phi_gbs, "phi_build_avg_ms": ph_avg,
                        "stage_ms_per_eval": stage_pass["stage_ms_per_eval"],
                        "dominant_stage_ms_per_eval_in_the_timed_region": {k: v[0] / args.steps for k, v in tim.items()}},
            # BASELINE.json's metric names it: "MFMA util on PHI'W PHI".  Flops-based from this run (n m (m+1) algorithmic flops per launch
            # over the kernel's HIP-event time in the timed region); counter-based from the committed rocprofv3 PMC passes of the same
            # command (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / kernel cycles - counters cannot be read inside this process).
            "roofline_syrk": {"bound": "mfma", "kernel": ("k_syrk_small (PHI' W PHI, n*m*(m+1) flops/launch: the whole triangle of 16 x 16 blocks in one workgroup; "
                                                          "the stage time includes the sum of the per-workgroup records)" if "k_syrk_small" in (route or "") else
                                                          "k_syrk (PHI' W PHI, n*m*(m+1) flops/launch: the symmetric half)") + (" on fp32-operand MFMAs" if f32_route else ""),
                              "achieved": ach_sy, "peak": F32_MFMA_PEAK_TFLOPS if f32_route else F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": ach_sy / (F32_MFMA_PEAK_TFLOPS if f32_route else F64_MFMA_PEAK_TFLOPS), "avg_ms": sy_avg,
                              "mfma_util_counters": (pmc_traffic(args.config) or {}).get("syrk_mfma_util") if world == 1 and not multi and not args.n else None,
                              "kernel_clock_ghz_counters": (pmc_traffic(args.config) or {}).get("syrk_kernel_clock_ghz") if world == 1 and not multi and not args.n else None,
                              "mfma_instructions_counters": (pmc_traffic(args.config) or {}).get("syrk_mfma_instructions") if world == 1 and not multi and not args.n else None,
                              "counters_source": (pmc_traffic(args.config) or {}).get("source") if world == 1 and not multi and not args.n else None,
                              "traffic": (pmc_traffic(args.config) or {}).get("syrk_bytes_per_launch") if world == 1 and not multi and not args.n else None},
            "phi_build": {"bound": "hbm (diagonal kinds) / fp64 vector ALU (covariance kinds, SURVEY.md 8d)", "avg_ms": ph_avg,
                          "algorithmic_GBs": phi_gbs, "peak_GBs": 8000.0, "frac": phi_gbs / 8000.0,
                          "fabric_GBs_counters": ((pmc_traffic(args.config) or {}).get("phi_bytes_per_launch", 0.0) / (ph_avg * 1e-3) / 1e9
                                                  if ph_avg > 0 and world == 1 and not multi and not args.n and (pmc_traffic(args.config) or {}).get("phi_bytes_per_launch") else None)},
            "finite": finite,
            # bitwise fingerprint of the last step's result (same theta sequence for every launch form)
            "check": {"f_last": float(fs[-1]).hex(), "g_sum": float(np.sum(g)).hex(), "g_absmax": float(np.max(np.abs(g))).hex()},
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        else:
            out["device"] = per_rank_single
        if route:
            out["route"] = route
        if gaps:
            out["gaps"] = gaps
        if graph_pass:
            out["graph_pass"] = graph_pass
        if single_graph:
            out["single_graph_pass"] = single_graph
        out["stage_pass"] = {k: v for k, v in stage_pass.items() if k != "stage_ms_per_eval"}
        if cfg.get("psi") and not f32_route:
            # config 5 in fp64: the per-pair sweeps of k_cpsi4.hip (four pairs per wave on v_mfma_f64_4x4x4).  Algorithmic work per
            # (sample, basis) pair at d = 20 in the M = Psi + Sigma form: PHI d^3/6 + d^2 = 1733 FMA, moments d^3/2 + 2 d^2 = 4800 FMA.
            mo_ms, mo_calls = tim.get("moments", (0.0, 0))
            mo_avg = mo_ms / max(1, mo_calls)
            pairs = float(n_local) * m
            ach_mo = pairs * 4800 * 2 / (mo_avg * 1e-3) / 1e12 if mo_avg > 0 else 0.0
            ach_ph = pairs * 1733 * 2 / (ph_avg * 1e-3) / 1e12 if ph_avg > 0 else 0.0
            out["roofline_gemm"] = out["roofline"]
            out["roofline"] = {"bound": "mfma", "kernel": "k_cpsi4_moments (per-(sample, basis) d x d sweeps on 4 x 4 f64 MFMA tiles, 9600 flops/pair; "
                                                          "latency-bound small-matrix work: the pivot-block chain, DESIGN.md section 3 row 9f)",
                               "achieved": ach_mo, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_mo / F64_MFMA_PEAK_TFLOPS,
                               "traffic": None, "avg_ms": mo_avg}
            out["roofline_f64_pair_kernels"] = {"k_cpsi4_moments": {"achieved": ach_mo, "avg_ms": mo_avg, "flops_per_pair": 9600},
                                                "k_cpsi4_phi (+ fill, row dots)": {"achieved": ach_ph, "avg_ms": ph_avg, "flops_per_pair": 3466}}
        if f32_route:
            # config 5: the dominant kernels are the fp32 per-pair factorisations (VALU), not the GEMMs.  Algorithmic work per
            # (sample, basis) pair at D = 20, whitened form: PHI 3480 FMA, moments 6370 FMA, 2 flops each.
            mo_ms, mo_calls = tim.get("moments", (0.0, 0))
            mo_avg = mo_ms / max(1, mo_calls)
            pairs = float(n_local) * m
            ach_mo = pairs * 6370 * 2 / (mo_avg * 1e-3) / 1e12 if mo_avg > 0 else 0.0
            ach_ph = pairs * 3480 * 2 / (ph_avg * 1e-3) / 1e12 if ph_avg > 0 else 0.0
            pair = {
                "bound": "fp32 VALU (157.3 TFLOP/s spec = the fp32 MFMA rate)",
                "peak": 157.3, "unit": "TFLOP/s",
                # measured issue ceilings of fp32 instruction streams on this chip (tools/pk_fma_rate.hip, profiles/r04_ubench_pk_fma_rate.txt;
                # DESIGN.md section 8): the kernels' 2:1 mix of v_pk_fma_f32 and v_fma_f32 at four waves per SIMD, and at the ONE wave per
                # SIMD a 220-register triangle per lane allows
                "measured_issue_ceiling": {"mix_4_waves_per_simd": 123.2, "mix_1_wave_per_simd": 86.6, "unit": "TFLOP/s issued",
                                           "note": "issued multiply-adds; the kernels issue 1.4x the algorithmic ones (junk slots of the row-pair layout, non-FMA instructions)"},
                "k_psi32_moments": {"achieved": ach_mo, "frac": ach_mo / 157.3, "avg_ms": mo_avg, "flops_per_pair": 12740},
                "k_psi32_phi (+ fill, row dots)": {"achieved": ach_ph, "frac": ach_ph / 157.3, "avg_ms": ph_avg, "flops_per_pair": 6960}}
            # the dominant kernel of this configuration is the moment kernel, not a GEMM: it is the top-level roofline
            out["roofline_gemm"] = out["roofline"]
            out["roofline"] = {"bound": "valu", "kernel": "k_psi32_moments (per-(sample, basis) d x d factorisations, 12740 flops/pair)",
                               "achieved": ach_mo, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_mo / F32_MFMA_PEAK_TFLOPS,
                               "traffic": (pmc_traffic(args.config) or {}).get("moments_bytes_per_launch") if world == 1 and not multi and not args.n else None,
                               "avg_ms": mo_avg}
            out["roofline_f32_pair_kernels"] = pair
        if world == 1 and not args.no_cpu_baseline and va_mask is None and not multi:
            rows = max(2000, min(n, n // 8 if n >= 200000 else n))        # ~13 s of CPU work at c4 (16 BLAS threads), the whole problem at c2 / c3
            if cfg.get("psi"):
                rows = 60                                                  # per-pair d x d loops in NumPy: ~1e5 pairs
            ref, cb = cpu_baseline(cfg, model, theta0, X, y, omega, rows)
            out["cpu_baseline"] = cb
            # parity gate on the same sample through the HIP path
            c2 = gpz_amd.GPzContext(model, X[:rows], y[:rows], synth_psi(cfg, np.arange(rows)) if cfg.get("psi") else None,
                                    None if omega is None else omega[:rows], device=local_rank, dtype=dtype)
            f2, g2 = c2.eval(theta0)
            c2.close()
            # north_star: "NLL/gradient matching the reference to 1e-8 relative".  The fp64 configurations are gated at exactly that;
            # 50 cond(SIGMA) eps - what rounding in the m x m solve alone may cost either side - is reported, not used (c4: 1.5e-6).
            out["parity"] = {"rows": rows, "rel_f": abs(f2 - ref.nlogML) / abs(ref.nlogML),
                             "rel_g_max": float(np.max(np.abs(g2 - ref.grad)) / np.max(np.abs(ref.grad))),
                             "cond_sigma": ref.cond, "tol_g": 1e-8, "cond_scaled_bound_g": max(1e-8, 50 * ref.cond * 2.2e-16)}
            if cfg.get("psi"):
                # conditioning of the precision matrices of this theta (synth draws them with cond of a few units: the reference's
                # dGamma chain through inv(Gamma_j'Gamma_j), GPz.m:174-180, keeps its digits and rel_g_max is gated as it stands)
                m_, d_ = cfg["m"], cfg["d"]
                Gm = theta0[m_ * d_:m_ * d_ + d_ * d_ * m_].reshape((d_, d_, m_), order="F")
                cg = np.array([np.linalg.cond(Gm[:, :, j].T @ Gm[:, :, j]) for j in range(m_)])
                out["parity"].update({"dtype": dtype, "max_cond_gamma": float(cg.max()),
                                      "tol_f": 1e-4 if f32_route else 1e-8,
                                      "tol_g": 1e-3 if f32_route else max(1e-8, 50 * max(ref.cond, float(cg.max()) ** 1.5) * 2.2e-16)})
            out["parity"]["pass"] = bool(out["parity"]["rel_f"] <= out["parity"].get("tol_f", 1e-8) and
                                         out["parity"]["rel_g_max"] <= out["parity"]["tol_g"])
            out["parity"]["meets_1e-8"] = bool(out["parity"]["rel_f"] <= 1e-8 and out["parity"]["rel_g_max"] <= 1e-8)
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
