"""dev script: first GPU contact (not a pytest file)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlmc_amd import _lib, Legendre, Monomial
from mlmc_amd.engine import LevelAccumulator, level_stats
from oracle import oracle_np as onp
import scipy.stats

_lib.init(0, _lib.FLAG_TIMING)
print(_lib.device_info())
dom = tuple(scipy.stats.norm().ppf([1e-4, 1 - 1e-4]))
g = np.linspace(dom[0]-0.1, dom[1]+0.1, 1001)
for R in (5, 32, 64, 100):
    e = Legendre(R, dom).eval_all(g)
    o = onp.eval_all(onp.Basis(onp.LEGENDRE, R, dom), g)
    print("eval R", R, "nan match", np.array_equal(np.isnan(e), np.isnan(o)), "maxerr", np.nanmax(np.abs(e - o)))
L = 3
steps = [s[0] for s in onp.determine_level_parameters(L, [0.5, 0.01])]
for R in (5, 10, 32, 64, 100):
    N = [40000, 30000, 20001]
    fn = Legendre(R, dom)
    acc = LevelAccumulator(fn, L)
    chunks = []
    for l in range(L):
        f, c = onp.synth_level_samples(l, N[l], steps)
        acc.push(l, f, c)
        x = np.stack([f, c if c is not None else np.zeros_like(f)], axis=-1)[None]
        chunks.append([x[:, :, :1] if l == 0 else x])
    n, n_rm, s, sp = acc.finalize()
    b = onp.Basis(onp.LEGENDRE, R, dom)
    r = onp.estimate_mean(chunks, lambda x: onp.moments_rows(b, x))
    lm, lv = level_stats(n, s, sp)
    print("R", R, "n", n, n_rm, "counts ok", np.array_equal(n, r.n_samples), np.array_equal(n_rm, r.n_rm_samples),
          "l_means err", np.max(np.abs(lm - r.l_means)), "l_vars relerr", np.max(np.abs(lv - r.l_vars) / (np.abs(r.l_vars) + 1e-300)),
          "mean0", lm[0, 0], "ktime", acc.kernel_time())
# covariance
for R in (8, 16, 32, 64):
    N = [3000, 2000, 1001]
    fn = Legendre(R, dom)
    acc = LevelAccumulator(fn, L, LevelAccumulator.COV)
    chunks = []
    for l in range(L):
        f, c = onp.synth_level_samples(l, N[l], steps)
        acc.push(l, f, c)
        x = np.stack([f, c if c is not None else np.zeros_like(f)], axis=-1)[None]
        chunks.append([x[:, :, :1] if l == 0 else x])
    n, n_rm, s, sp = acc.finalize()
    b = onp.Basis(onp.LEGENDRE, R, dom)
    r = onp.estimate_mean(chunks, lambda x: onp.covariance_rows(b, x))
    print(n, r.n_samples, n_rm, r.n_rm_samples, s[:, 0], r.sums[:, 0])
    print("COV R", R, "counts ok", np.array_equal(n, r.n_samples), "s err", np.max(np.abs(s - r.sums)), "scale", np.max(np.abs(r.sums)),
          "sp err", np.max(np.abs(sp - r.sums_sq)), "scale", np.max(np.abs(r.sums_sq)))
# perf probe
import torch
for R, Nl in ((32, 10_000_000), (64, 10_000_000)):
    fn = Legendre(R, dom)
    acc = LevelAccumulator(fn, L)
    data = []
    for l in range(L):
        f, c = onp.synth_level_samples(l, Nl, steps)
        data.append((torch.from_numpy(f).cuda(), None if c is None else torch.from_numpy(c).cuda()))
    for it in range(3):
        acc.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for l in range(L):
            acc.push(l, *data[l])
        n, n_rm, s, sp = acc.finalize()
        dt = time.perf_counter() - t0
        print("perf R", R, "wall ms", dt * 1e3, "evals/s", L * Nl * R / dt, "kernel", acc.kernel_time())
