"""Developer tool: per-iteration cost of the L-BFGS memory (add + direction) at c4's size, host NumPy vs device."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpz_amd import host
p, corr = (int(sys.argv[1]) if len(sys.argv) > 1 else 113001), 100
rng = np.random.default_rng(0)
hm, dm = host._LBFGS(p, corr), host._LBFGSDevice(p, corr)
g_old = rng.standard_normal(p)
th, td = [], []
for it in range(130):
    d = rng.standard_normal(p); t = 0.7
    g = g_old + 0.3 * t * d + 0.05 * rng.standard_normal(p)
    G, GO, D = host.DevVec.from_host(g), host.DevVec.from_host(g_old), host.DevVec.from_host(d)
    t0 = time.perf_counter(); hm.add_step(g, g_old, t, d); dh = hm.direction(g); th.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); dm.add_step(G, GO, t, D); dd = dm.direction(G); x = dd.amax(); td.append(time.perf_counter() - t0)
    g_old = g
print("p=%d corrections=%d: host two-loop %.2f ms/iter, device %.3f ms/iter (full memory, last 20 iterations); max rel diff %.1e"
      % (p, corr, 1e3 * np.mean(th[-20:]), 1e3 * np.mean(td[-20:]), np.max(np.abs(dd.host() - dh)) / np.max(np.abs(dh))))
