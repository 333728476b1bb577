"""Dev: where the non-kernel time of one configs[1] estimate goes (C ABI call in a tight loop vs the Python wrapper)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator, level_stats
_lib.init(0, flags=int(os.environ.get("FLAGS", "0")))
dev = torch.device("cuda", 0)
L, n, R = 3, int(os.environ.get("N", 10_000_000)), 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
chunks = []
for l in range(L):
    f = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    chunks.append((l, f, None if l == 0 else f + 0.01))
torch.cuda.synchronize()
fn = Legendre(R, (-3.719, 3.719))
acc = LevelAccumulator(fn, L, LevelAccumulator.MOMENTS)
lib = _lib.lib()
k = len(chunks)
levels = (C.c_int32 * k)(*[c[0] for c in chunks])
fine = (C.c_void_p * k)(*[c[1].data_ptr() for c in chunks])
coarse = (C.c_void_p * k)(*[None if c[2] is None else c[2].data_ptr() for c in chunks])
ns = (C.c_int64 * k)(*[n] * k)
nn = np.empty(L, dtype=np.int64); nr = np.empty(L, dtype=np.int64); s = np.empty((L, R)); sp = np.empty((L, R))
args = (acc._h, k, levels, fine, coarse, ns, _lib.DEVICE, _lib.ptr(nn), _lib.ptr(nr), _lib.ptr(s), _lib.ptr(sp))
def loop(fn_, N=300):
    for _ in range(30): fn_()
    t0 = time.perf_counter()
    for _ in range(N): fn_()
    return 1e6 * (time.perf_counter() - t0) / N
print("C ABI mlmc_accum_estimate      : %.1f us" % loop(lambda: lib.mlmc_accum_estimate(*args)))
print("wrapper acc.estimate           : %.1f us" % loop(lambda: acc.estimate(chunks)))
def full():
    n_, nr_, s_, sp_ = acc.estimate(chunks)
    lm, lv = level_stats(n_, s_, sp_)
    return np.sum(lm, axis=0), np.sum(lv / n_[:, None], axis=0)
print("wrapper + level_stats + sums   : %.1f us" % loop(full))
def pieces():
    lib.mlmc_accum_reset(acc._h)
    for l, f, c in chunks:
        lib.mlmc_accum_push(acc._h, l, C.c_void_p(f.data_ptr()), None if c is None else C.c_void_p(c.data_ptr()), n, _lib.DEVICE)
    lib.mlmc_accum_finalize(acc._h, _lib.ptr(nn), _lib.ptr(nr), _lib.ptr(s), _lib.ptr(sp), _lib.HOST)
print("reset + 3 push + finalize      : %.1f us" % loop(pieces))
# the GPU-side floor: same launches, host waits only at the end of a batch of 50 estimates
def batch():
    for _ in range(50):
        lib.mlmc_accum_reset(acc._h)
        for l, f, c in chunks:
            lib.mlmc_accum_push(acc._h, l, C.c_void_p(f.data_ptr()), None if c is None else C.c_void_p(c.data_ptr()), n, _lib.DEVICE)
        lib.mlmc_accum_finalize_packed(acc._h, C.c_void_p(packed.data_ptr()), _lib.DEVICE)
    lib.mlmc_synchronize()
packed = torch.empty(2 * L + 2 * L * R, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
print("50 estimates back to back / 50 : %.1f us" % (loop(batch, 20) / 50))
