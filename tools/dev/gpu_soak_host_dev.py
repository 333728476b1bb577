"""Dev soak, host-storage side: many cold / warm estimates over file-like chunked storages through the read-ahead feed, with
cache clears and a second thread; device memory and thread count must stay flat."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.estimator import Estimate
from mlmc_amd.quantity import quantity_estimate as qe
from mlmc_amd.quantity.quantity import make_root_quantity
from mlmc_amd.quantity.quantity_spec import QuantitySpec
from mlmc_amd.sample_storage import Memory
_lib.init(0)
spec = [QuantitySpec(name="q", unit="m", shape=(2, 1), times=[1, 2], locations=['0'])]
steps = [0.5, 0.07, 0.01]
rng = np.random.default_rng(5)
def storage(chunk):
    st = Memory(chunk_size=chunk, copy_chunks=True)
    st.save_global_data(result_format=spec, level_parameters=[[s] for s in steps])
    for l, n in enumerate((60000, 30000, 9000)):
        x = rng.standard_normal((n, 4))
        st.set_level_samples(l, x + steps[l], None if l == 0 else x + steps[l - 1])
    return st
stop = False
def second():
    st2 = storage(7000)
    q2 = make_root_quantity(st2, spec)['q'][2]['0'][1, 0]
    while not stop:
        Estimate(q2, st2, Legendre(9, (-5.0, 5.0))).estimate_moments()
th = threading.Thread(target=second, daemon=True); th.start()
free0 = None
t0 = time.perf_counter()
for it in range(int(os.environ.get("ITERS", 400))):
    st = storage([2000, 5000, 977][it % 3])
    root = make_root_quantity(st, spec)['q']
    q = [root[1]['0'][0, 0], root, (root[2]['0'][1, 0] - 0.25) * root[1]['0'][0, 0]][it % 3]
    est = Estimate(q, st, Legendre(6 + it % 5, (-5.0, 5.0)))
    est.estimate_moments(); est.estimate_covariance(); est.estimate_diff_vars()
    if it % 7 == 0:
        qe.device_cache_clear()
    if it == 50:
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
    if it % 100 == 0:
        print(it, "free GB", round(torch.cuda.mem_get_info()[0] / 2**30, 3), "threads", threading.active_count(), "cache items", len(qe._device_cache._items), flush=True)
stop = True; th.join(timeout=30)
qe.device_cache_clear(); torch.cuda.synchronize()
print("elapsed s", round(time.perf_counter() - t0, 1), "free delta MB since it 50:", round((free0 - torch.cuda.mem_get_info()[0]) / 2**20, 1), "threads", threading.active_count())
