"""Dev: where the cold (host-resident storage) first estimate spends its time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.estimator import Estimate
from mlmc_amd.quantity import quantity_estimate as qe
from mlmc_amd.quantity.quantity import make_root_quantity
from mlmc_amd.quantity.quantity_spec import QuantitySpec
from mlmc_amd.sample_storage import Memory
_lib.init(0)
n = int(os.environ.get("N", 10_000_000)); L = 3
spec = [QuantitySpec(name="q", unit="", shape=(1, 1), times=[1], locations=['0'])]
st = Memory(); st.save_global_data(result_format=spec, level_parameters=[[0.5], [0.07], [0.01]])
rng = np.random.default_rng(1)
for l in range(L):
    x = rng.normal(size=n)
    st.set_level_samples(l, x + 0.01 * l, None if l == 0 else x)
st.save_n_ops([(l, (1.0 * n, n)) for l in range(L)])
q = make_root_quantity(st, spec)['q'][1]['0'][0, 0]
est = Estimate(q, st, Legendre(32, (-3.719, 3.719)))
est.estimate_moments()
import cProfile, pstats
for rep in range(2):
    qe.device_cache_clear(); torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter(); est.estimate_moments(); t1 = time.perf_counter()
    pr.disable()
    print("cold ms", 1e3 * (t1 - t0))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
# raw copies of the same arrays, one by one
for l in range(L):
    raw = st._results[l]
    src = torch.from_numpy(np.ascontiguousarray(raw.reshape(raw.shape[0], -1)))
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = src.to("cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"level {l} copy {src.numel()*8/1e6:.0f} MB rep {rep}: {1e3*(t1-t0):.2f} ms  {src.numel()*8/(t1-t0)/1e9:.1f} GB/s")
        del y
