import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, cProfile, pstats
from mlmc_amd import _lib, Legendre
from mlmc_amd.estimator import Estimate
from mlmc_amd.quantity.quantity import make_root_quantity
from mlmc_amd.sim.synth_device import SynthDeviceStorage
_lib.init(0)
n = int(os.environ.get("N", 10_000_000))
st = SynthDeviceStorage([[0.5], [0.07], [0.01]], [n, n, n])
root = make_root_quantity(st, st.load_result_format())
q = root['length'][1]['10'][0]
est = Estimate(q, st, Legendre(32, (-3.719, 3.719)))
for it in range(3):
    t0 = time.perf_counter(); d, info, res, mom = est.construct_density(tol=1e-8); t1 = time.perf_counter()
    print("construct_density ms", round(1e3 * (t1 - t0), 3), "nit", res.nit, "moments", mom.size)
t0 = time.perf_counter(); m, v = est.estimate_moments(); print("estimate_moments ms", round(1e3 * (time.perf_counter() - t0), 3))
t0 = time.perf_counter(); c, cv = est.estimate_covariance(); print("estimate_covariance ms", round(1e3 * (time.perf_counter() - t0), 3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    est.construct_density(tol=1e-8)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(16)
