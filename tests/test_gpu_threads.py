"""Thread safety of the C ABI and of the Python API on top of it (run with -m gpu): ctypes releases the GIL during a call,
so host threads really do arrive inside the library at the same time; its entry points serialise on one lock
(include/mlmc_hip.h, "Thread safety")."""
import threading

import numpy as np
import pytest

from tests.util import level_arrays

pytestmark = pytest.mark.gpu
DOM = (-3.7190164854556804, 3.7190164854556804)


@pytest.fixture(scope="module")
def hip():
    from mlmc_amd import _lib
    _lib.init(0)
    return _lib


def _run_threads(workers):
    errors = []

    def guard(fn):
        def run():
            try:
                fn()
            except BaseException as e:          # noqa: BLE001 - reported to the test thread
                errors.append(e)
        return run
    threads = [threading.Thread(target=guard(w), daemon=True) for w in workers]      # daemon: a stuck worker cannot keep the process
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    if errors:
        raise errors[0]
    assert not any(t.is_alive() for t in threads), "a worker thread hangs"


def test_concurrent_estimates_through_the_c_abi(hip):
    """Four host threads, each with its own accumulators (moments R = 32 on device chunks, covariance R = 16 on host chunks, a
    64-term one-pass estimate, percentiles), 40 estimates each, all at once: every result equals the single-threaded one
    bit for bit."""
    import torch
    from mlmc_amd import Legendre
    from mlmc_amd.engine import LevelAccumulator, percentiles
    levels = level_arrays([60001, 35000, 9001], [0.5, 0.07, 0.01], 1, 7)
    dev = [(l, torch.from_numpy(f[0].copy()).cuda(), None if c is None else torch.from_numpy(c[0].copy()).cuda())
           for l, (f, c) in enumerate(levels)]
    torch.cuda.synchronize()

    def moments32():
        acc = LevelAccumulator(Legendre(32, DOM), 3)
        return lambda: acc.estimate(dev, reduce=False)

    def moments64():
        acc = LevelAccumulator(Legendre(64, DOM), 3)
        return lambda: acc.estimate(dev, reduce=False)

    def cov16():
        acc = LevelAccumulator(Legendre(16, DOM), 3, LevelAccumulator.COV)

        def run():
            acc.reset()
            for l, (f, c) in enumerate(levels):                 # host chunks: staged through the accumulator's buffers
                acc.push(l, f[0], None if c is None else c[0])
            return acc.finalize(reduce=False)
        return run

    def pct():
        return lambda: (percentiles(dev[1][1], [1.0, 50.0, 99.0]),)

    makers = [moments32, cov16, moments64, pct]
    jobs = [m() for m in makers]
    want = [job() for job in jobs]
    got = [[] for _ in jobs]

    def worker(i):
        def run():
            for _ in range(40):
                got[i].append(jobs[i]())
        return run
    _run_threads([worker(i) for i in range(len(jobs))])
    for i, results in enumerate(got):
        assert len(results) == 40
        for r in results:
            assert all(np.array_equal(a, b) for a, b in zip(r, want[i])), makers[i].__name__


def test_concurrent_estimates_through_the_python_api(hip):
    """Two threads run Estimate.estimate_moments / estimate_covariance / construct_density on different quantities of one
    storage while a third clears the device cache now and then: same numbers as alone."""
    from mlmc_amd import Legendre
    from mlmc_amd.estimator import Estimate
    from mlmc_amd.quantity import quantity_estimate as qe
    from mlmc_amd.quantity.quantity import make_root_quantity
    from mlmc_amd.quantity.quantity_spec import QuantitySpec
    from mlmc_amd.sample_storage import Memory
    spec = [QuantitySpec(name="q", unit="m", shape=(2, 1), times=[1, 2], locations=['0'])]
    steps = [0.5, 0.07, 0.01]
    levels = level_arrays([20001, 12000, 4001], steps, 4, 0)
    st = Memory(chunk_size=5000)
    st.save_global_data(result_format=spec, level_parameters=[[s] for s in steps])
    for l, (f, c) in enumerate(levels):
        st.set_level_samples(l, f.T, None if c is None else c.T)
    root = make_root_quantity(st, spec)['q']
    qa = root[1]['0'][0, 0]
    qb = (root[2]['0'][1, 0] - 0.25) * root[1]['0'][0, 0]
    fa, fb = Legendre(12, DOM), Legendre(9, (-20.0, 20.0))

    def analysis(q, fn):
        est = Estimate(q, st, fn)
        m, v = est.estimate_moments()
        cov, cv = est.estimate_covariance()
        lv, n = est.estimate_diff_vars()
        return m, v, cov, cv, lv, n
    want_a, want_b = analysis(qa, fa), analysis(qb, fb)
    out = {"a": [], "b": []}
    stop = threading.Event()

    def run_a():
        for _ in range(15):
            out["a"].append(analysis(qa, fa))

    def run_b():
        try:
            for _ in range(15):
                out["b"].append(analysis(qb, fb))
        finally:
            stop.set()

    def clearer():
        while not stop.wait(0.01):
            qe.device_cache_clear()
    _run_threads([run_a, run_b, clearer])
    for key, want in (("a", want_a), ("b", want_b)):
        assert len(out[key]) == 15
        for r in out[key]:
            # the level variances come from the covariance sums or, when the cache was cleared in between, from a moments
            # pass: equal to 1e-10, everything else bit for bit
            assert all(np.array_equal(x, y) for i, (x, y) in enumerate(zip(r, want)) if i != 4), key
            assert np.allclose(r[4], want[4], rtol=1e-10, atol=0), key


def test_worker_threads_on_a_rank_bound_to_a_non_zero_device(tmp_path):
    """HIP's current device is per host thread and defaults to 0: on a rank bound to device 1 a worker thread used to
    allocate scratch and launch with device 0 current (round-2 advice).  Every entry point now re-binds the calling thread
    (MLMC_API_GUARD -> bind_thread_to_device).  Needs two visible GPUs; a child process, because a process binds one device."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: a non-zero device index cannot be bound here")
    code = r'''
import threading, sys
import numpy as np
sys.path.insert(0, %r)
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator
_lib.init(1)
x = np.random.default_rng(0).standard_normal(200001)
def job():
    acc = LevelAccumulator(Legendre(24, (-3.7, 3.7)), 1, LevelAccumulator.COV)
    acc.push(0, x, None)
    return acc.finalize(reduce=False)
want = job()
out, err = [], []
def run():
    try:
        out.append(job())
    except BaseException as e:
        err.append(e)
ts = [threading.Thread(target=run) for _ in range(3)]
[t.start() for t in ts]; [t.join(120) for t in ts]
assert not err, err
assert len(out) == 3 and all(all(np.array_equal(a, b) for a, b in zip(r, want)) for r in out)
print("ok")
''' % (str(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))),)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
