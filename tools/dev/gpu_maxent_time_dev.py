import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mlmc_amd import _lib, Legendre
from mlmc_amd.tool import simple_distribution as sd
import cProfile, pstats
_lib.init(0)
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
g6 = np.load(os.path.join(GOLDEN, "G6_maxent.npz"))
keys = sorted(set(k.rsplit("_moment_data", 1)[0] for k in g6.files if k.endswith("_moment_data") and "_old_" not in k))
print(keys[:6])
for key in keys[:4]:
    dom = tuple(g6[key + "_domain"]); md = g6[key + "_moment_data"]
    R = md.shape[0]
    fn = Legendre(R, dom)
    d = sd.SimpleDistribution(fn, md.copy(), domain=dom)
    d.estimate_density_minimize(tol=1e-8)
    t0 = time.perf_counter()
    for _ in range(20):
        d = sd.SimpleDistribution(fn, md.copy(), domain=dom)
        res = d.estimate_density_minimize(tol=1e-8)
    dt = (time.perf_counter() - t0) / 20
    print(key, "R", R, "nit", res.nit, "ms", round(dt * 1e3, 3))
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    d = sd.SimpleDistribution(fn, md.copy(), domain=dom)
    res = d.estimate_density_minimize(tol=1e-8)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
