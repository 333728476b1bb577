#!/usr/bin/env python3
"""Soak run (GPU box): a few thousand mixed estimates over freshly built storages and trees -- chunked host storages through
the level streamer, resident estimates, covariance (matrix cores, linearised mean, banded spline mean, > 128 moments from
values), construct_density, bootstrap, cache clears -- watching device memory, host RSS and the cache's book-keeping."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psutil
import torch

from mlmc_amd import _lib, Legendre, Spline
from mlmc_amd.estimator import Estimate
from mlmc_amd.quantity import quantity_estimate as qe
from mlmc_amd.quantity.quantity import make_root_quantity
from mlmc_amd.quantity.quantity_spec import QuantitySpec
from mlmc_amd.sample_storage import Memory

_lib.init(0)
ITERS = int(os.environ.get("ITERS", "400"))
rng = np.random.default_rng(5)
spec = [QuantitySpec(name="q", unit="", shape=(2, 1), times=[1], locations=['0'])]
dom = (-3.7, 3.7)
proc = psutil.Process()
t0 = time.time()
marks = []
for it in range(ITERS):
    L = int(rng.integers(1, 4))
    n = [int(rng.choice([2000, 30011, 120000])) for _ in range(L)]
    st = Memory(chunk_size=int(rng.choice([0, 5000, 20000])) or None, copy_chunks=bool(rng.integers(2)))
    st.save_global_data(result_format=spec, level_parameters=[[0.5 / (l + 1)] for l in range(L)])
    for l in range(L):
        x = rng.standard_normal((n[l], 2))
        st.set_level_samples(l, x, None if l == 0 else x + 0.05 * rng.standard_normal((n[l], 2)))
    root = make_root_quantity(st, spec)['q'][1]['0']
    q = root[0, 0] if it % 3 else (root[0, 0] - 0.2) * root[1, 0]
    kind = it % 6
    fn = Legendre(int(rng.choice([5, 21, 33, 64])), dom if it % 3 else (-15.0, 15.0))
    est = Estimate(q, st, fn)
    est.estimate_moments()
    if kind == 0:
        est.estimate_covariance()
        est.estimate_diff_vars()
    elif kind == 1:
        est.construct_density(tol=1e-6)
    elif kind == 2:
        Estimate(q, st, Spline(24, dom if it % 3 else (-15.0, 15.0))).construct_density(tol=1e-6)
    elif kind == 3:
        est.est_bootstrap(n_subsamples=5)
    elif kind == 4:
        qe.estimate_mean(qe.covariance(q, Legendre(130, dom if it % 3 else (-15.0, 15.0))))
    else:
        Estimate.estimate_domain(q, st)
    if it % 37 == 0:
        qe.device_cache_clear()
    if it % 50 == 0 or it == ITERS - 1:
        gc.collect()
        free, total = torch.cuda.mem_get_info()
        c = qe._device_cache
        marks.append((it, round((total - free) / 2 ** 30, 3), round(proc.memory_info().rss / 2 ** 30, 3), len(c._items), round(c._bytes / 2 ** 20, 1)))
        assert c._bytes == sum(item[2] for item in c._items.values())
        print("it %4d  device used %.3f GiB  host rss %.3f GiB  cache items %d (%.1f MiB)  %.0f s" % (marks[-1] + (time.time() - t0,)), flush=True)
qe.device_cache_clear()
gc.collect()
torch.cuda.empty_cache()
dev = [m[1] for m in marks[2:]]
rss = [m[2] for m in marks[2:]]
print("device used min/max GiB", min(dev), max(dev), " host rss min/max GiB", min(rss), max(rss))
assert max(rss) - min(rss) < 1.5, "host memory grows"
print("soak OK", ITERS, "iterations", round(time.time() - t0, 1), "s")
