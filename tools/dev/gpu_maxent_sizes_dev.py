"""Dev: max-entropy solve time by number of orthogonal moments (cooperative launch vs step-by-step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.stats as stats
from mlmc_amd import _lib, Legendre, TransformedMoments
from mlmc_amd.tool import simple_distribution as sd
_lib.init(0)
dom = tuple(stats.norm().ppf([1e-4, 1 - 1e-4]))
pdf = lambda x: stats.norm().pdf(x) / (1 - 2e-4)
for R in (32, 64, 100):
    base = Legendre(R, dom)
    cov = sd.compute_semiexact_cov(base, pdf)
    ortho, info = sd.construct_ortogonal_moments(base, cov, tol=1e-4)
    means = sd.compute_semiexact_moments(ortho, pdf)
    md = np.stack([means, np.ones_like(means)], axis=1)
    for stepwise in ("0", "1"):
        os.environ.pop("MLMC_MAXENT_STEPWISE", None)          # the library tests for presence, not value
        if stepwise == "1":
            os.environ["MLMC_MAXENT_STEPWISE"] = "1"
        ts = []
        for rep in range(10):
            d = sd.SimpleDistribution(ortho, md.copy(), domain=dom)
            t0 = time.perf_counter()
            res = d.estimate_density_minimize(tol=1e-8)
            ts.append(1e3 * (time.perf_counter() - t0))
        print(f"R {R} -> {ortho.size} orthogonal, stepwise={stepwise}: nit {res.nit} success {res.success} ms {[round(t, 3) for t in ts]}", flush=True)
