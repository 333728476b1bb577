"""Where one BASELINE configs[2] estimate step spends its time outside the covariance kernels: the C call (reset, five
accumulate launches with their reductions, finalize, the copy back) against the host formulas (level statistics, regression,
re-allocation)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator, level_stats, moments_from_covariance
from mlmc_amd.estimator import Estimate, estimate_n_samples_for_target_variance
_lib.init(0, _lib.FLAG_TIMING)
g = torch.Generator(device="cuda"); g.manual_seed(1)
n, L, R = 10_000_000, 5, 64
data = []
for l in range(L):
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    data.append((l, (x + 0.07 * torch.sqrt(1e-4 + x.abs())).contiguous(), None if l == 0 else (x + 0.5 * torch.sqrt(1e-4 + x.abs())).contiguous()))
fn = Legendre(R, (-3.719, 3.719))
acc = LevelAccumulator(fn, L, LevelAccumulator.COV)
regress = Estimate(None, None, fn)._all_moments_variance_regression
steps_h = np.array([0.5, 0.19, 0.07, 0.027, 0.01]); n_ops = [(1 / h) ** 2 * np.log(max(1 / h, 2.0)) for h in steps_h]
t_c = t_h1 = t_h2 = 0.0
for it in range(30):
    if it == 10:
        t_c = t_h1 = t_h2 = 0.0; acc.kernel_time()
    t0 = time.perf_counter()
    nn, n_rm, s, sp = acc.estimate(data)
    t1 = time.perf_counter()
    l_means, l_vars = level_stats(nn, s, sp)
    mean, var = np.sum(l_means, axis=0), np.sum(l_vars / nn[:, None], axis=0)
    t2 = time.perf_counter()
    s_m, sp_m = moments_from_covariance(s, sp, R)
    _, raw = level_stats(nn, s_m, sp_m)
    reg = regress(raw, steps_h)
    ne = estimate_n_samples_for_target_variance(1e-6, reg, n_ops, n_levels=L)
    t3 = time.perf_counter()
    t_c += t1 - t0; t_h1 += t2 - t1; t_h2 += t3 - t2
k_ms = acc.kernel_time()[0] / 20
print("per step: C call %.3f ms (kernels %.3f ms, rest %.3f ms)  level stats of the covariance %.3f ms  regression + allocation %.3f ms"
      % (t_c / 20 * 1e3, k_ms, t_c / 20 * 1e3 - k_ms, t_h1 / 20 * 1e3, t_h2 / 20 * 1e3))
