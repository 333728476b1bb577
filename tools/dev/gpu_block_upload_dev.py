"""Cold-path cost of a quantity tree over a Memory storage with M stored values per sample: per-row host gather +
upload vs one block upload + strided LOAD on the device.  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mlmc_amd import Legendre
from mlmc_amd.quantity import quantity_estimate as qe
from mlmc_amd.quantity.quantity import make_root_quantity
from mlmc_amd.quantity.quantity_spec import QuantitySpec
from mlmc_amd.sample_storage import Memory

def storage(ns, m, chunk):
    spec = [QuantitySpec(name="q", unit="", shape=(m, 1), times=[1], locations=['0'])]
    st = Memory(chunk_size=chunk) if chunk else Memory()
    st.save_global_data(result_format=spec, level_parameters=[[0.5 ** l] for l in range(len(ns))])
    rng = np.random.default_rng(3)
    for l, n in enumerate(ns):
        f = rng.normal(1.0, 0.5, size=(n, m))
        st.set_level_samples(l, f, None if l == 0 else f + 0.01 * rng.normal(size=(n, m)))
    st.save_n_ops([(l, (1.0, 1)) for l in range(len(ns))])
    return st, spec

for m, ns in ((24, (2000000, 1000000, 500000)), (4, (4000000, 2000000)), (100, (400000, 200000))):
    st, spec = storage(ns, m, None)
    root = make_root_quantity(st, spec)['q'][1]['0']
    fn = Legendre(8, (-2.0, 4.0))
    for label, q in (("all rows", root), ("m/4 rows", root[: max(m // 4, 1)])):
        for mode in ("0", "1", "0", "1"):
            os.environ["MLMC_HIP_BLOCK_UPLOAD"] = mode
            qe.device_cache_clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = qe.estimate_mean(qe.moments(q, fn))
            t1 = time.perf_counter()
            r2 = qe.estimate_mean(qe.moments(q, fn))
            t2 = time.perf_counter()
            print(f"M={m:4d} {label:9s} block={mode}  cold {1e3*(t1-t0):9.2f} ms   warm {1e3*(t2-t1):7.3f} ms   mean[1]={r.mean.ravel()[1]:.15g}", flush=True)
