"""Wall time of the adaptive sampling loop (test/test_run.py:93-105 shape) with every part on the device.  GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scipy.stats as stats
from mlmc_amd import Legendre
from mlmc_amd.estimator import Estimate, estimate_n_samples_for_target_variance, determine_level_parameters
from mlmc_amd.quantity.quantity import make_root_quantity
from mlmc_amd.sampler import DeviceSampler
from mlmc_amd.sim.synth_device import SynthDeviceStorage, result_format

for L, target_var, n0 in ((3, 1e-6, [1000, 100]), (5, 1e-7, [10000, 100]), (5, 2e-8, [10000, 100])):
    steps = determine_level_parameters(L, [0.5, 0.01])
    for rep in range(2):
        t0 = time.perf_counter()
        st = SynthDeviceStorage(steps, [0] * L)
        sampler = DeviceSampler(st)
        fn = Legendre(32, tuple(stats.norm().ppf([1e-4, 1 - 1e-4])))
        sampler.set_initial_n_samples(n0)
        sampler.schedule_samples()
        sampler.ask_sampling_pool_for_samples()
        value = make_root_quantity(st, result_format())['length'][1]['10'][0]
        est = Estimate(value, st, fn)
        rounds = 0
        while True:
            variances, n_ops = est.estimate_diff_vars_regression(sampler._n_scheduled_samples)
            n_est = estimate_n_samples_for_target_variance(target_var, variances, n_ops, n_levels=L)
            rounds += 1
            if sampler.process_adding_samples(n_est, 0, 0.1):
                break
        sampler.ask_sampling_pool_for_samples()
        means, vars_ = est.estimate_moments(fn)
        t1 = time.perf_counter()
        print(f"L={L} target={target_var:g} rounds={rounds} n={st.get_n_collected()} max var={np.max(vars_):.3g}  "
              f"loop {1e3*(t1-t0):.1f} ms ({1e3*(t1-t0)/rounds:.2f} ms/round)", flush=True)
