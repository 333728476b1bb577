"""Dev: PCIe-inclusive cost of the first estimate over a host-resident storage (configs[1] shape) vs warm estimates."""
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
for tree in ("1", "0"):
    os.environ["MLMC_HIP_DEVICE_TREE"] = tree
    for rep in range(2):
        qe.device_cache_clear(); torch.cuda.synchronize()
        t0 = time.perf_counter(); m, v = est.estimate_moments(); t1 = time.perf_counter()
        m2, v2 = est.estimate_moments(); t2 = time.perf_counter()
        nbytes = (2 * L - 1) * n * 8
        print(f"device_tree={tree} rep {rep}: cold {1e3*(t1-t0):8.2f} ms ({nbytes/(t1-t0)/1e9:6.2f} GB/s of samples, {L*n*32/(t1-t0):.3e} moment-evals/s)  warm {1e3*(t2-t1):7.3f} ms  equal {np.array_equal(m, m2)}")
# raw H2D rates for reference
x = np.random.default_rng(2).normal(size=(n, 2))
for name, src in (("pageable", torch.from_numpy(x)), ("pinned", torch.from_numpy(x).pin_memory())):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        y = src.to("cuda", non_blocking=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"H2D {name}: {x.nbytes/dt/1e9:.1f} GB/s")
import cProfile, pstats
os.environ["MLMC_HIP_DEVICE_TREE"] = "1"
est.estimate_moments()
t0 = time.perf_counter()
for _ in range(200):
    est.estimate_moments()
print("warm avg ms", 1e3 * (time.perf_counter() - t0) / 200)
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    est.estimate_moments()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
