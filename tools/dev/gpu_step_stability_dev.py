"""Dev: is the per-estimate time stable over consecutive batches (GPU clock / power states)?"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator, level_stats
_lib.init(0, flags=1)
dev = torch.device("cuda", 0)
L, n, R = 3, 10_000_000, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
chunks = []
for l in range(L):
    f = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    chunks.append((l, f, None if l == 0 else f + 0.01))
torch.cuda.synchronize()
acc = LevelAccumulator(Legendre(R, (-3.719, 3.719)), L, LevelAccumulator.MOMENTS)
mode = os.environ.get("MODE", "plain")
for b in range(16):
    acc.kernel_time()
    t0 = time.perf_counter()
    for _ in range(300):
        r = acc.estimate(chunks)
        if mode == "stats":
            lm, lv = level_stats(r[0], r[2], r[3])
            m = np.sum(lm, axis=0); v = np.sum(lv / r[0][:, None], axis=0)
    dt = (time.perf_counter() - t0) / 300
    ms, launches, _ = acc.kernel_time()
    print(f"batch {b:2d} mode {mode}: {1e6*dt:7.1f} us per estimate, kernel {1e3*ms/max(launches,1):7.1f} us", flush=True)
    if mode == "pause":
        time.sleep(0.05)
