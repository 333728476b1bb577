import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator
_lib.init(0, _lib.FLAG_TIMING)
g = torch.Generator(device="cuda"); g.manual_seed(1)
L = 3
for M, n in ((1, 4_000_000), (4, 1_000_000), (24, 1_000_000), (24, 100_000)):
    data = []
    for l in range(L):
        x = torch.randn(M, n, dtype=torch.float64, device="cuda", generator=g)
        data.append((x.contiguous(), None if l == 0 else (x + 0.01).contiguous()))
    for R, mode in ((10, LevelAccumulator.MOMENTS), (32, LevelAccumulator.MOMENTS), (10, LevelAccumulator.COV)):
        acc = LevelAccumulator(Legendre(R, (-3.7, 3.7)), L, mode, n_comp=M)
        for it in range(8):
            if it == 3:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            acc.reset()
            for l in range(L):
                f, c = data[l]
                acc.push(l, f if M > 1 else f[0], c if (c is None or M > 1) else c[0])
            acc.finalize()
        dt = (time.perf_counter() - t0) / 5
        evals = L * n * M * (R if mode == LevelAccumulator.MOMENTS else R * R)
        print(f"M {M:2d} n {n:8d} R {R:2d} {'mom' if mode == 0 else 'cov'}: {dt*1e3:8.3f} ms/estimate  {evals/dt:.3e} evals/s", flush=True)
