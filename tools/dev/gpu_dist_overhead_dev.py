"""Dev: cost of the pieces of the distributed finalize (world size 1, RCCL) on top of the kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator, _pinned_like
_lib.init(0)
dev = torch.device("cuda", 0)
L, n, R = 3, 10_000_000, 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
chunks = []
for l in range(L):
    f = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    chunks.append((l, f, None if l == 0 else f + 0.01))
acc = LevelAccumulator(Legendre(R, (-3.719, 3.719)), L, LevelAccumulator.MOMENTS)
def step_dist():
    acc.reset()
    for l, f, c in chunks: acc.push(l, f, c)
    return acc.finalize(group=dist.group.WORLD)
def step_local():
    acc.reset()
    for l, f, c in chunks: acc.push(l, f, c)
    return acc.finalize(reduce=False)
def loop(fn, N=400):
    for _ in range(300): fn()
    t0 = time.perf_counter()
    for _ in range(N): fn()
    return 1e6 * (time.perf_counter() - t0) / N
print("local finalize        : %.1f us" % loop(step_local))
print("distributed finalize  : %.1f us" % loop(step_dist))
packed = torch.empty(2 * L + 2 * L * R, dtype=torch.float64, device=dev)
lib = _lib.lib()
def pieces(k):
    acc.reset()
    for l, f, c in chunks: acc.push(l, f, c)
    _lib.check(lib.mlmc_accum_finalize_packed(acc._h, _lib.ptr(packed), _lib.DEVICE))
    if k >= 1: dist.all_reduce(packed)
    if k >= 2:
        host = _pinned_like(packed); host.copy_(packed, non_blocking=True)
    _lib.check(lib.mlmc_synchronize())
for k, name in enumerate(("finalize_packed + sync", "+ all_reduce", "+ copy to pinned")):
    print("%-22s: %.1f us" % (name, loop(lambda: pieces(k))))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step_dist()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(12)
dist.destroy_process_group()
