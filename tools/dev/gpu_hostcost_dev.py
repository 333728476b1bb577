import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator, level_stats
_lib.init(0, int(os.environ.get("TFLAG", "0")))
dom = (-3.719, 3.719)
n = 10_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
data = []
for l in range(3):
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    data.append(((x + 0.07 * torch.sqrt(1e-4 + x.abs())).contiguous(), (x + 0.5 * torch.sqrt(1e-4 + x.abs())).contiguous()))
SHARED = os.environ.get("SHARED", "0") == "1"
fn = Legendre(32, dom)
L = 3
acc = LevelAccumulator(fn, L)
T = {k: 0.0 for k in ("reset", "push", "finalize", "stats")}
N = 200
for it in range(N + 20):
    if it == 20:
        T = {k: 0.0 for k in T}; t_all = time.perf_counter()
    t0 = time.perf_counter(); acc.reset(); t1 = time.perf_counter()
    for l in range(L):
        f, c = data[0] if SHARED else data[l]
        acc.push(l, f, None if l == 0 else c)
    t2 = time.perf_counter()
    r = acc.finalize(); t3 = time.perf_counter()
    lm, lv = level_stats(r[0], r[2], r[3]); m = np.sum(lm, axis=0); v = np.sum(lv / r[0][:, None], axis=0); t4 = time.perf_counter()
    T["reset"] += t1 - t0; T["push"] += t2 - t1; T["finalize"] += t3 - t2; T["stats"] += t4 - t3
tot = time.perf_counter() - t_all
print({k: round(1e6 * v / N, 1) for k, v in T.items()}, "total us/step", round(1e6 * tot / N, 1))
