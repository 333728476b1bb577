import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator
_lib.init(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
L, M, n, R = 3, 4, 1_000_000, 10
data = []
for l in range(L):
    x = torch.randn(M, n, dtype=torch.float64, device="cuda", generator=g)
    data.append((x.contiguous(), None if l == 0 else (x + 0.01).contiguous()))
acc = LevelAccumulator(Legendre(R, (-3.7, 3.7)), L, 0, n_comp=M)
for it in range(20):
    acc.reset()
    for l in range(L):
        acc.push(l, data[l][0], data[l][1])
    acc.finalize()
