"""cProfile of one cold streamed estimate (main thread) + wall-clock marks inside _LevelStreamer (run on the GPU box)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlmc_amd import Legendre, _lib
from mlmc_amd.estimator import Estimate, determine_level_parameters
from mlmc_amd.quantity import quantity_estimate as qe
from mlmc_amd.quantity.quantity import make_root_quantity
from mlmc_amd.quantity.quantity_spec import QuantitySpec
from mlmc_amd.sample_storage import Memory
_lib.init(0)
L, n_l, chunk, R = 3, 10_000_000, 100_000, 32
steps_h = [s[0] for s in determine_level_parameters(L, [0.5, 0.01])]
spec = [QuantitySpec(name="q", unit="", shape=(1, 1), times=[1], locations=['0'])]
st = Memory(chunk_size=chunk, copy_chunks=True)
st.save_global_data(result_format=spec, level_parameters=[[h] for h in steps_h])
for l in range(L):
    x = np.random.default_rng(99 + l).standard_normal(n_l)
    st.set_level_samples(l, x, None if l == 0 else x + 0.1)
q = make_root_quantity(st, spec)['q'][1]['0'][0, 0]
est = Estimate(q, st, Legendre(R, (-3.7, 3.7)))
for _ in range(2):
    qe.device_cache_clear()
    est.estimate_moments()
orig = qe._LevelStreamer.stream_level
marks = []
def wrapped(self, *a, **k):
    t0 = time.perf_counter()
    r = orig(self, *a, **k)
    marks.append(time.perf_counter() - t0)
    return r
qe._LevelStreamer.stream_level = wrapped
qe.device_cache_clear()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
est.estimate_moments()
pr.disable()
print("estimate %.2f ms; stream_level per level (ms):" % (1e3 * (time.perf_counter() - t0)), [round(1e3 * m, 2) for m in marks])
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
