"""Dev: Python-side cost of Estimate.estimate_moments on resident samples (profile of 300 calls)."""
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
for _ in range(400): est.estimate_moments()
t0 = time.perf_counter()
for _ in range(300): est.estimate_moments()
print("estimate_moments: %.1f us" % (1e6 * (time.perf_counter() - t0) / 300))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300): est.estimate_moments()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
