import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mlmc_amd import _lib, Legendre, TransformedMoments
from mlmc_amd.engine import LevelAccumulator
_lib.init(0)
dom = (-3.0, 3.0)
rng = np.random.default_rng(0)
x = rng.normal(size=1000)
t0 = time.time()
for it in range(3000):
    R = 3 + it % 40
    fn = Legendre(R, dom)
    v = fn.eval_all(x[: 1 + it % 999])
    if it % 3 == 0:
        tm = TransformedMoments(fn, np.eye(min(R, 5), R))
        v = tm.eval_all(x[:17])
    acc = LevelAccumulator(fn, 2)
    acc.push(0, x)
    acc.push(1, x, x * 0.99)
    r = acc.finalize()
    acc.close()
    if it % 500 == 0:
        print(it, r[0], time.time() - t0, flush=True)
print("stress ok", time.time() - t0)
