"""Dev: Python-side profile of the adaptive loop (L = 5, R = 32)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.stats as stats
from mlmc_amd import Legendre
from mlmc_amd.estimator import Estimate, estimate_n_samples_for_target_variance, determine_level_parameters
from mlmc_amd.quantity.quantity import make_root_quantity
from mlmc_amd.sampler import DeviceSampler
from mlmc_amd.sim.synth_device import SynthDeviceStorage, result_format

def run():
    L, target_var, n0 = 5, 1e-7, [10000, 100]
    steps = determine_level_parameters(L, [0.5, 0.01])
    st = SynthDeviceStorage(steps, [0] * L)
    sampler = DeviceSampler(st)
    fn = Legendre(32, tuple(stats.norm().ppf([1e-4, 1 - 1e-4])))
    sampler.set_initial_n_samples(n0); sampler.schedule_samples(); sampler.ask_sampling_pool_for_samples()
    value = make_root_quantity(st, result_format())['length'][1]['10'][0]
    est = Estimate(value, st, fn)
    while True:
        variances, n_ops = est.estimate_diff_vars_regression(sampler._n_scheduled_samples)
        n_est = estimate_n_samples_for_target_variance(target_var, variances, n_ops, n_levels=L)
        if sampler.process_adding_samples(n_est, 0, 0.1):
            break
    return est.estimate_moments(fn)
run(); run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
