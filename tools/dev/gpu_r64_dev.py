import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator, level_stats
_lib.init(0, _lib.FLAG_TIMING)
g = torch.Generator(device="cuda"); g.manual_seed(1)
n = 10_000_000; L = 5
data = []
for l in range(L):
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    data.append(((x + 0.07 * torch.sqrt(1e-4 + x.abs())).contiguous(), None if l == 0 else (x + 0.5 * torch.sqrt(1e-4 + x.abs())).contiguous()))
for R in (48, 64, 96, 128):
    acc = LevelAccumulator(Legendre(R, (-3.719, 3.719)), L)
    res = None
    for it in range(25):
        if it == 5:
            acc.kernel_time(); torch.cuda.synchronize(); t0 = time.perf_counter()
        acc.reset()
        for l in range(L):
            acc.push(l, data[l][0], data[l][1])
        res = acc.finalize()
    dt = (time.perf_counter() - t0) / 20
    ms, launches, _ = acc.kernel_time()
    print(f"R {R:3d}: {dt*1e3:7.3f} ms/estimate, kernel {ms/20:7.3f} ms ({launches//20} launches)  s[1,5]={res[2][1,5]:.12e} sp[2,R-1]={res[3][2,R-1]:.12e}")
