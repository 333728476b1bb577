import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator
_lib.init(0, _lib.FLAG_TIMING)
dom = (-3.719, 3.719)
n = 10_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
data=[]
for l in range(5):
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    data.append(((x + 0.07 * torch.sqrt(1e-4 + x.abs())).contiguous(), None if l==0 else (x + 0.5 * torch.sqrt(1e-4 + x.abs())).contiguous()))
for R in (49, 56, 64, 48, 32):
    acc = LevelAccumulator(Legendre(R, dom), 5)
    chunks=[(l,)+data[l] for l in range(5)]
    for it in range(30):
        if it == 10: acc.kernel_time()
        r = acc.estimate(chunks, reduce=False)
    ms, launches, nb = acc.kernel_time()
    print("R", R, "kernel ms per estimate", ms/20, "launches", launches//20, flush=True)
