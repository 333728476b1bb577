import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.util import level_arrays
from mlmc_amd import _lib, Legendre
from mlmc_amd.estimator import Estimate
from mlmc_amd.quantity import quantity_estimate as qe
from mlmc_amd.quantity.quantity import make_root_quantity
from mlmc_amd.quantity.quantity_spec import QuantitySpec
from mlmc_amd.sample_storage import Memory
_lib.init(0)
spec = [QuantitySpec(name="q", unit="m", shape=(2, 1), times=[1, 2], locations=['0'])]
steps = [0.5, 0.07, 0.01]
levels = level_arrays([40013, 25001, 9000], steps, 4, 17)
st = Memory(chunk_size=3001, copy_chunks=True)
st.save_global_data(result_format=spec, level_parameters=[[s] for s in steps])
for l, (f, c) in enumerate(levels):
    st.set_level_samples(l, np.ascontiguousarray(f.T), None if c is None else np.ascontiguousarray(c.T))
reads = []
inner = st.sample_pairs_level
import traceback, threading
def wrapped(s):
    reads.append((s.level_id, s.chunk_id))
    if len(reads) in (27, 28):
        print("READ", s.level_id, s.chunk_id, threading.current_thread().name)
        traceback.print_stack(limit=8)
    return inner(s)
st.sample_pairs_level = wrapped
root = make_root_quantity(st, spec)['q']
dom = (-3.7, 3.7)
for name, q in (("scalar", root[1]['0'][0, 0]), ("root", root), ("tree", (root[2]['0'][1, 0] - 0.25) * root[1]['0'][0, 0])):
    n0 = len(reads)
    Estimate(q, st, Legendre(5, dom)).estimate_moments()
    kinds = {}
    for k in qe._device_cache._items:
        kinds[str(k[0])[:12] if not isinstance(k[0], tuple) else "result"] = kinds.get(str(k[0])[:12] if not isinstance(k[0], tuple) else "result", 0) + 1
    print(name, "reads", len(reads) - n0, "cache kinds", kinds, "block meta", len(qe._block_meta))
