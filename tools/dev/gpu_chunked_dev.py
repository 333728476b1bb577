import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.estimator import Estimate
from mlmc_amd.quantity.quantity import make_root_quantity
from mlmc_amd.sim.synth_device import SynthDeviceStorage
_lib.init(0)
n = 10_000_000
for chunk in [None if c == "None" else int(c) for c in os.environ.get("CHUNKS", "None,2500000,1000000,250000,100000").split(",")]:
    st = SynthDeviceStorage([[0.5], [0.07], [0.01]], [n, n, n], chunk_size=chunk)
    q = make_root_quantity(st, st.load_result_format())['length'][1]['10'][0]
    est = Estimate(q, st, Legendre(32, (-3.719, 3.719)))
    est.estimate_moments(); est.estimate_covariance()
    t0 = time.perf_counter()
    for _ in range(5): m, v = est.estimate_moments()
    t1 = time.perf_counter()
    for _ in range(3): c, cv = est.estimate_covariance()
    t2 = time.perf_counter()
    print(f"chunk {str(chunk):>9s}: moments {1e3*(t1-t0)/5:8.3f} ms   covariance {1e3*(t2-t1)/3:8.3f} ms   mean[1] {m[1]:.12e}", flush=True)
