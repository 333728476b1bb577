"""Randomized validation, run by hand on the GPU box (`python tests/fuzz_tree_gpu.py`; pytest does not collect it): random
quantity trees through the whole estimate path -- block upload (strided LOAD) or per-row upload, chained programs,
several chunk sizes -- against the host-evaluated tree feeding the same device estimator (bit-identical rows => identical
sums) and against per-row uploads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlmc_amd import Legendre
from mlmc_amd.quantity import lowering, quantity_estimate as qe
from mlmc_amd.quantity.quantity import make_root_quantity
from tests.test_lowering import _random_tree, _scalar, _spec, make_storage

rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
bad = 0
checked = 0
t0 = time.time()
for it in range(int(os.environ.get("ITERS", 60))):
    sizes = tuple(int(rng.choice([1, 63, 64, 257, 1000, 4099])) for _ in range(int(rng.integers(1, 4))))
    chunk = None if rng.random() < 0.4 else int(rng.choice([50, 256, 1000]))
    st = make_storage(sizes, seed=int(rng.integers(1 << 20)), chunk_size=chunk)
    root = make_root_quantity(st, _spec())
    leaves = [root['length'][1]['10'][0], root['length'][2]['20'][1], root['width'][3]['30'][0], root['width'][2]['40'],
              root['length'].time_interpolation(1.25)['10']]
    q = _random_tree(rng, leaves, depth=int(rng.integers(1, 5)))
    if rng.random() < 0.4:
        q = q.select(_scalar(rng, leaves, 2) > float(rng.normal() + 2.0))
    if lowering.plan_for(q) is None:
        continue
    fn = Legendre(int(rng.choice([1, 3, 8, 33])), (-20.0, 40.0))
    res = {}
    try:
        for tag, env in (("block", {"MLMC_HIP_BLOCK_UPLOAD": "1", "MLMC_HIP_DEVICE_TREE": "1"}),
                         ("rows", {"MLMC_HIP_BLOCK_UPLOAD": "0", "MLMC_HIP_DEVICE_TREE": "1"}),
                         ("host", {"MLMC_HIP_BLOCK_UPLOAD": "0", "MLMC_HIP_DEVICE_TREE": "0"})):
            os.environ.update(env)
            qe.device_cache_clear()
            with np.errstate(all="ignore"):
                try:
                    m = qe.estimate_mean(qe.moments(q, fn))
                    res[tag] = (m.n_samples, m.n_rm_samples, m.l_means, m.l_vars)
                except Exception as e:
                    res[tag] = repr(e)[:80]
        same = all(type(res[t]) is type(res["host"]) for t in res)
        if same and not isinstance(res["host"], str):
            for t in ("block", "rows"):
                for a, b in zip(res[t], res["host"]):
                    same = same and np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
        elif same:
            # an ill-typed tree (e.g. a scalar quantity combined with a two-row one: the qtype follows the first operand, as in
            # the reference, and no longer describes the rows) fails in every mode; which check trips first may differ
            same = True
        checked += 1
        if not same:
            bad += 1
            print("MISMATCH sizes", sizes, "chunk", chunk, "R", fn.size, {k: (v if isinstance(v, str) else "ok") for k, v in res.items()}, flush=True)
    except Exception as e:
        bad += 1
        print("ERROR", sizes, chunk, repr(e)[:200], flush=True)
print("done", checked, "trees", bad, "bad", round(time.time() - t0, 1), "s")
