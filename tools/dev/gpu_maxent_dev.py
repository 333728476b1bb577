import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mlmc_amd import _lib, Legendre, TransformedMoments
from mlmc_amd.tool import simple_distribution as sd
_lib.init(0)
g6 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "G6_maxent.npz"))
for key in ("norm12_R7", "norm12_R21", "norm12_R41", "lognorm_R41"):
    R = int(key.split("_R")[1])
    dom = tuple(g6[key + "_domain"])
    base = Legendre(R, dom)
    ortho = TransformedMoments(base, g6[key + "_L"])
    d = sd.SimpleDistribution(ortho, g6[key + "_moment_data"].copy(), domain=dom)
    print("solving", key, flush=True)
    t0 = time.perf_counter()
    res = d.estimate_density_minimize(tol=1e-8)
    dt = time.perf_counter() - t0
    ref = g6[key + "_sd_multipliers"]
    print(key, "nit", res.nit, "fun_norm", res.fun_norm, "success", res.success, "ms", dt * 1e3,
          "mult err", np.max(np.abs(d.multipliers - ref)), "dens err", np.max(np.abs(d.density(g6[key + "_xgrid"]) - g6[key + "_sd_density"])), flush=True)
