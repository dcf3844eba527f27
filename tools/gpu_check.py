"""Developer check on a GPU box: component-by-component comparison with the oracle (verbose)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpz_amd
from oracle import gpz_oracle as O

rng = np.random.default_rng(1)

def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b))))

def data(n, d, k=1):
    X = rng.standard_normal((n, d))
    Y = np.sin(X @ rng.standard_normal((d, k)) / np.sqrt(d)) + 0.1 * rng.standard_normal((n, k))
    Y -= Y.mean(0)
    return X, Y

print("== Dxy"); X, _ = data(300, 5); P = rng.standard_normal((17, 5))
print("  rel", rel(gpz_amd.Dxy(X, P), O.Dxy(X, P)))

print("== nan_groups"); Xn = rng.standard_normal((5000, 6)); msk = rng.random((5000, 6)) < 0.15; Xn[msk] = np.nan
gid, ng = gpz_amd.nan_groups(Xn); rg, pats = O.nan_groups(Xn)
print("  groups", ng, pats.shape[0], "exact", bool(np.array_equal(gid, rg)))

print("== inv_logdet")
for m in (5, 32, 33, 100, 257, 1000):
    A = rng.standard_normal((m, 3 * m)); S = A @ A.T + np.eye(m)
    t = time.time(); Xi, ld, info = gpz_amd.inv_logdet(S, return_info=True); t = time.time() - t
    Ri, rl = O.inv_logdet(S)
    print(f"  m={m} info={info} rel_inv={rel(Xi, Ri):.2e} rel_logdet={abs(ld-rl)/abs(rl):.2e} cond={O.cond_of(S):.1e} t={t*1e3:.1f}ms")

print("== getPHI")
for method in ("GL", "VL", "GD", "VD", "GC", "VC"):
    for (n, d, m, k) in ((300, 3, 7, 1), (1000, 10, 50, 2), (257, 7, 33, 1)):
        X, Y = data(n, d, k)
        model, th = O.init_theta(X, Y, method, m, True, rng)
        th = th + 0.05 * rng.standard_normal(th.size)
        o = th.size - 2 * m * k; th[o:o + m * k] = 0.05 * rng.standard_normal(m * k)
        PHI, G, lnB = gpz_amd.getPHI(X, None, th, model)
        rP, rG, rB = O.getPHI(X, None, th, model)
        print(f"  {method} n={n} d={d} m={m} k={k}: PHI {rel(PHI, rP):.2e} lnBeta {rel(lnB, rB):.2e} Gamma {rel(G, rG):.1e}")

print("== GPz eval")
for method in ("GL", "VL", "GD", "VD", "GC", "VC"):
    for het in (True, False):
        for (n, d, m, k) in ((300, 3, 7, 1), (2000, 10, 64, 1), (600, 5, 20, 2)):
            X, Y = data(n, d, k)
            model, th = O.init_theta(X, Y, method, m, het, rng)
            th = th + 0.05 * rng.standard_normal(th.size)
            if het:
                o = th.size - 2 * m * k; th[o:o + m * k] = 0.05 * rng.standard_normal(m * k)
            om = rng.random((n, 1)) + 0.5
            tr = rng.random(n) < 0.8
            ref = O.GPz(th, model, X, Y, None, om, tr, ~tr)
            ctx = gpz_amd.GPzContext(model, X, Y, None, om, tr, ~tr)
            f, g = ctx.eval(th)
            w, iS, part = ctx.solve(th)
            rs = O.GPz(th, model, X, Y, None, om, tr, ~tr, nargout=3)
            st = ctx.stats
            es = max(abs(st[k_] - ref.stats[k_]) for k_ in ref.stats)
            print(f"  {method} het={int(het)} n={n} d={d} m={m} k={k}: f {abs(f-ref.nlogML)/abs(ref.nlogML):.2e} "
                  f"g {rel(g, ref.grad):.2e} stats {es:.1e} w {rel(w, rs.w):.2e} iS {rel(iS, rs.iSigma_w):.2e} "
                  f"part {rel(part, rs.nlogML):.2e} info={ctx.info} cond={ref.cond:.1e}")
            if rel(g, ref.grad) > 1e-6:
                # per-block gradient error
                md = m * model.d; gd = model.g_dim
                blocks = {"dP": (0, md), "dG": (md, md + gd), "dlnA": (md + gd, md + gd + m * k), "db": (md + gd + m * k, md + gd + m * k + k)}
                if het:
                    o = md + gd + m * k + k
                    blocks["dv"] = (o, o + m * k); blocks["dlnT"] = (o + m * k, o + 2 * m * k)
                print("     ", {b: f"{rel(g[a:c], ref.grad[a:c]):.1e}" for b, (a, c) in blocks.items()})
            ctx.close()

print("== predict")
X, Y = data(500, 4, 1)
model, th = O.init_theta(X, Y, "VC", 16, True, rng)
th = th + 0.05 * rng.standard_normal(th.size)
rs = O.GPz(th, model, X, Y, nargout=3)
model.sets["best"] = {"theta": th, "w": rs.w, "iSigma_w": rs.iSigma_w}
out = gpz_amd.predict(X[:200], model); ref = O.predict(X[:200], model)
print("  mu %.2e sigma %.2e nu %.2e beta_i %.2e PHI %.2e" % tuple(rel(out[i], ref[i]) for i in (0, 1, 2, 3, 5)))
print("done")
