"""Developer tool: randomised sweep of ROW-SHARDED evaluations (2 or 3 ranks on one GPU, gloo moving the device buffers)
against the oracle on the unsharded data: methods, input noise, missing values (shared NaN-pattern table), weights,
masks, fp32 pair path.  usage: fuzz_sharded.py [cases] [seed] [world]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def draw(rng):
    method = str(rng.choice(["GL", "VL", "GD", "VD", "GC", "VC"]))
    cov = method[1] == "C"
    d = int(rng.integers(2, 8)); m = int(rng.choice([3, 7, 16, 33, 64])); k = int(rng.choice([1, 1, 2]))
    n = int(rng.choice([9, 40, 130, 515, 1100]))
    psi = bool(rng.random() < 0.4); nanfrac = float(rng.choice([0.0, 0.0, 0.3]))
    if cov and (psi or nanfrac > 0) and n * m > 30000: n = max(9, 30000 // m)
    return dict(method=method, d=d, m=m, k=k, n=n, hetero=bool(rng.random() < 0.7), psi=psi, nanfrac=nanfrac,
                seed=int(rng.integers(1 << 30)), om=bool(rng.random() < 0.4), masks=int(rng.choice([0, 1, 2])),
                f32=bool(cov and psi and rng.random() < 0.5))   # with missing values every rank must fall back to fp64 together


def build(cfg):
    from helpers import make_problem
    from test_gpu_parity import _well_conditioned_gamma
    model, theta, X, Y, Psi, r2 = make_problem(cfg["n"], cfg["d"], cfg["m"], cfg["k"], cfg["method"], cfg["hetero"],
                                               seed=cfg["seed"], psi=cfg["psi"], nanfrac=cfg["nanfrac"])
    if cfg["f32"]:
        theta = _well_conditioned_gamma(model, theta, r2)
    n = cfg["n"]
    if cfg["nanfrac"] > 0 and cfg["d"] > 2 and cfg["seed"] % 2:   # several missing dimensions per row, many patterns
        miss = r2.random((n, cfg["d"])) < 0.15
        miss[:, int(r2.integers(cfg["d"]))] = False
        X = X.copy(); X[miss] = np.nan
    om = (r2.random((n, 1)) + 0.5) if cfg["om"] else None
    tr = va = None
    if cfg["masks"] >= 1:
        tr = r2.random(n) < 0.75
        tr[:3] = True
        if cfg["masks"] == 2:
            va = ~tr
    return model, theta, X, Y, Psi, om, tr, va


def worker(rank, world, port, cfgs, q):
    import torch, torch.distributed as dist
    import gpz_amd
    from gpz_amd import dist as gdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    out = []
    for ci, cfg in enumerate(cfgs):
        if os.environ.get("FUZZ_VERBOSE"): print(f"[rank {rank}] case {ci}: {cfg}", flush=True)
        model, theta, X, Y, Psi, om, tr, va = build(cfg)
        try:
            Xs, Ys, oms, trs, vas = gdist.shard_rows(rank, world, X, Y, om, tr, va)
            cov = model.method[1] == "C"
            pats = gdist.nan_patterns(X, tr, va) if (cov and np.isnan(X).any()) else None
            ctx, err = None, None
            try:
                ctx = gpz_amd.GPzContext(model, Xs, Ys, gdist.shard_psi(rank, world, Psi, tr, va), oms, trs, vas, rank=rank,
                                         world=world, allreduce=gdist.make_allreduce(), patterns=pats,
                                         dtype="f32" if cfg["f32"] else "f64")
            except Exception as e:
                err = repr(e)[:300]
            # a rank that failed to build its context must not leave the others waiting in the all-reduce
            flag = torch.tensor([0.0 if ctx is not None else 1.0])
            dist.all_reduce(flag)
            if flag.item() > 0:
                if ctx is not None: ctx.close()
                out.append((None, None, None, err or "another rank failed to create its context"))
                continue
            f, g = ctx.eval(theta); st = dict(ctx.stats); ctx.close()
            out.append((f, g, st, None))
        except Exception as e:
            out.append((None, None, None, repr(e)[:300]))
    q.put((rank, out))
    dist.destroy_process_group()


def main():
    import socket
    import torch.multiprocessing as mp
    from oracle import gpz_oracle as O
    from helpers import grad_tol, rel
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    cfgs = [draw(rng) for _ in range(cases)]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn"); q = mpc.Queue()
    procs = [mpc.Process(target=worker, args=(r, world, port, cfgs, q)) for r in range(world)]
    t0 = time.time()
    for p in procs: p.start()
    res = dict(q.get(timeout=1500) for _ in procs)
    for p in procs: p.join(timeout=60)
    bad = refused = 0
    for c, cfg in enumerate(cfgs):
        model, theta, X, Y, Psi, om, tr, va = build(cfg)
        ref = O.GPz(theta, model, X, Y, Psi, om, tr, va)
        tol_f, tol_g = (1e-4, 1e-3) if (cfg["f32"] and cfg["nanfrac"] == 0.0) else (1e-8, grad_tol(ref.cond))
        if model.method[1] == "C" and not (cfg["f32"] and cfg["nanfrac"] == 0.0):
            P, G, *_ = O.unpack_theta(theta, model); Gm = O.expand_gamma(G, model)
            cg = max(np.linalg.cond(Gm[:, :, j].T @ Gm[:, :, j]) for j in range(Gm.shape[2]))
            tol_g = max(tol_g, 50 * cg * 2.2e-16)
            if (cfg["psi"] or cfg["nanfrac"] > 0) and cg > 1e4: tol_g = max(tol_g, 1e-2)
        for r in range(world):
            f, g, st, err = res[r][c]
            if err is not None:
                if "cannot be sharded over" in err:   # fewer training rows than ranks: the host refuses, as designed
                    refused += 1; break
                bad += 1; print("ERROR", cfg, "rank", r, err); break
            es = max((0.0 if (np.isnan(v) and np.isnan(st.get(kk, np.nan))) else abs(st.get(kk, np.nan) - v) / max(1.0, abs(v)))
                     for kk, v in ref.stats.items())
            if not (abs(f - ref.nlogML) <= max(tol_f, tol_g) * abs(ref.nlogML) and rel(g, ref.grad) <= tol_g and es <= max(1e-9, tol_f)):
                bad += 1
                print("FAIL", cfg, "rank", r, f"ef={abs(f - ref.nlogML) / abs(ref.nlogML):.1e} eg={rel(g, ref.grad):.1e} es={es:.1e} tol_g={tol_g:.1e}")
                break
        else:
            if not all(res[r][c][0] == res[0][c][0] and np.array_equal(res[r][c][1], res[0][c][1]) for r in range(world)):
                bad += 1; print("FAIL ranks differ", cfg)
    print(f"{cases} sharded cases (world={world}), {bad} failures, {refused} refused (rows < ranks), {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
