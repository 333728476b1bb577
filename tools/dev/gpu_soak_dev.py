"""Dev soak: thousands of estimates / tree evaluations / density solves; device memory must stay flat."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.estimator import Estimate
from mlmc_amd.quantity import quantity_estimate as qe
from mlmc_amd.quantity.quantity import make_root_quantity
from mlmc_amd.sim.synth_device import SynthDeviceStorage
_lib.init(0)
st = SynthDeviceStorage([[0.5], [0.1], [0.02]], [200000, 80000, 30000], chunk_size=50000)
root = make_root_quantity(st, st.load_result_format())
x = root['length'][1]['10'][0]
y = root['width'][2]['40'][1]
free0 = None
t0 = time.perf_counter()
for it in range(int(os.environ.get("ITERS", 3000))):
    q = [x, (x - 0.3) * y, x.select(x > -1.0), np.exp(x * 0.1) + y][it % 4]      # new tree objects every time
    fn = Legendre(5 + it % 7, (-6.0, 8.0))                                        # new basis objects every time
    est = Estimate(q, st, fn)
    m, v = est.estimate_moments()
    if it % 10 == 0:
        c, _ = est.estimate_covariance()
    if it % 50 == 0 and q is x:
        est.construct_density(tol=1e-6)
    if it % 25 == 0:
        est.est_bootstrap(n_subsamples=3, sample_vector=[2000, 800, 300])
    if it == 200:
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
    if it % 500 == 0:
        torch.cuda.synchronize()
        print(it, "free GB", round(torch.cuda.mem_get_info()[0] / 2**30, 3), "cache items", len(qe._device_cache._items), flush=True)
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("elapsed s", round(time.perf_counter() - t0, 1), "free delta MB since it 200:", round((free0 - free1) / 2**20, 1))
