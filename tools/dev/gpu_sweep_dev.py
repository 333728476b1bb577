import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre, Monomial, Fourier
from mlmc_amd.engine import LevelAccumulator
_lib.init(0, _lib.FLAG_TIMING)
g = torch.Generator(device="cuda"); g.manual_seed(1)
n = 10_000_000; L = 3
data = []
for l in range(L):
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    data.append(((x + 0.07 * torch.sqrt(1e-4 + x.abs())).contiguous(), None if l == 0 else (x + 0.5 * torch.sqrt(1e-4 + x.abs())).contiguous()))
for cls in (Legendre, Monomial, Fourier):
    for R in [int(r) for r in os.environ.get("RS", "2,4,8,16,24,32,40,48,56,64,80,100,128").split(",")]:
        fn = cls(R, (-3.719, 3.719))
        acc = LevelAccumulator(fn, L)
        for it in range(13):
            if it == 3:
                acc.kernel_time()
            acc.reset()
            for l in range(L):
                acc.push(l, data[l][0], data[l][1])
            acc.finalize()
        ms, launches, nb = acc.kernel_time()
        ms /= 10
        print(f"{cls.__name__:9s} R {R:3d}: kernel {ms:7.3f} ms  {ms*1e9/(L*n*R):6.3f} ps/eval  {0.4/ms:6.2f} TB/s  ({launches//10} launches)", flush=True)
