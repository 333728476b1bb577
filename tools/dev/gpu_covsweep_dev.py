import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre, Monomial
from mlmc_amd.engine import LevelAccumulator
_lib.init(0, _lib.FLAG_TIMING)
g = torch.Generator(device="cuda"); g.manual_seed(1)
n = 10_000_000; L = 3
data = []
for l in range(L):
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    data.append(((x + 0.07 * torch.sqrt(1e-4 + x.abs())).contiguous(), None if l == 0 else (x + 0.5 * torch.sqrt(1e-4 + x.abs())).contiguous()))
for R in [int(r) for r in os.environ.get("RS", "64,4,5,8,10,16,20,24,32,40,48,64,96,128").split(",")]:
    acc = LevelAccumulator(Legendre(R, (-3.719, 3.719)), L, LevelAccumulator.COV, mean_only=bool(os.environ.get("MEAN_ONLY")))
    for it in range(5):
        if it == 2:
            acc.kernel_time()
        acc.reset()
        for l in range(L):
            acc.push(l, data[l][0], data[l][1])
        acc.finalize()
    ms, launches, nb = acc.kernel_time()
    ms /= 3
    flops = (6 * R * R) * 2 * n + (4 * R * R) * n
    print(f"cov R {R:3d}: kernel {ms:8.3f} ms  {flops/ms/1e9:7.2f} TFLOP/s (algorithmic)  {0.4/ms:6.2f} TB/s  ({launches//3} launches)", flush=True)
