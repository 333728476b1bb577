"""Dev: Python-side profile of est_bootstrap (100 sub-samples of 1e5 per level)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mlmc_amd import _lib, Legendre
from mlmc_amd.estimator import Estimate
from mlmc_amd.quantity.quantity import make_root_quantity
from mlmc_amd.sim.synth_device import SynthDeviceStorage
_lib.init(0)
n = 1_000_000
st = SynthDeviceStorage([[0.5], [0.07], [0.01]], [n, n, n])
q = make_root_quantity(st, st.load_result_format())['length'][1]['10'][0]
est = Estimate(q, st, Legendre(32, (-3.719, 3.719)))
est.est_bootstrap(n_subsamples=20, sample_vector=[100000] * 3)
t0 = time.perf_counter(); est.est_bootstrap(n_subsamples=100, sample_vector=[100000] * 3); print("100 sub-samples: %.2f ms" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable(); est.est_bootstrap(n_subsamples=100, sample_vector=[100000] * 3); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
