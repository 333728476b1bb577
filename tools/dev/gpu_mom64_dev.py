import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator
_lib.init(0, _lib.FLAG_TIMING)
dom = (-3.719, 3.719)
n = 10_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
f = (x + 0.07 * torch.sqrt(1e-4 + x.abs())).contiguous(); c = (x + 0.5 * torch.sqrt(1e-4 + x.abs())).contiguous()
for R in (32, 48, 64):
    for L in (1, 2, 3, 5):
        fn = Legendre(R, dom)
        acc = LevelAccumulator(fn, L)
        for it in range(4):
            acc.reset()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for l in range(L):
                acc.push(l, f, None if l == 0 else c)
            r = acc.finalize()
            dt = time.perf_counter() - t0
        ms, launches, nb = acc.kernel_time()
        print("R", R, "L", L, "wall ms", dt * 1e3, "kernel ms", ms, "launches", launches)
