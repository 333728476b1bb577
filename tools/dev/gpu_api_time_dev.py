import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
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
def T(name, fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    print(f"{name:42s} {1e3*(time.perf_counter()-t0)/reps:10.3f} ms", flush=True)
    return out
T("estimate_moments", est.estimate_moments)
T("estimate_covariance", est.estimate_covariance)
T("estimate_diff_vars", est.estimate_diff_vars)
T("estimate_diff_vars_regression", lambda: est.estimate_diff_vars_regression(st.get_n_collected()))
T("estimate_domain", lambda: Estimate.estimate_domain(q, st))
T("est_bootstrap(20 x 1e5/level)", lambda: est.est_bootstrap(n_subsamples=20, sample_vector=[100000, 100000, 100000]), reps=1)
T("est_bootstrap(20 x 1e7/level)", lambda: est.est_bootstrap(n_subsamples=20, sample_vector=[n, n, n]), reps=1)
T("construct_density", lambda: est.construct_density(tol=1e-8))
v = root['length'][2]
estv = Estimate(v, st, Legendre(10, (-3.719, 3.719)))
T("vector(4 rows) estimate_moments", estv.estimate_moments)
T("whole root (24 rows) estimate_moments", Estimate(root, st, Legendre(10, (-3.719, 3.719))).estimate_moments, reps=1)
import cProfile, pstats
for name, fn in (("regression", lambda: est.estimate_diff_vars_regression(st.get_n_collected())), ("domain", lambda: Estimate.estimate_domain(q, st)),
                 ("bootstrap", lambda: est.est_bootstrap(n_subsamples=20, sample_vector=[100000, 100000, 100000]))):
    pr = cProfile.Profile(); pr.enable(); fn(); pr.disable()
    print("=====", name)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
