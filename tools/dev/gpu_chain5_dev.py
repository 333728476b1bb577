"""Where does the construct_density chain of BASELINE configs[4]'s per-GPU share (spline R = 128, 1.25e7 samples) spend its time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Spline, Legendre
from mlmc_amd.engine import LevelAccumulator
from mlmc_amd.tool import simple_distribution as sd
_lib.init(0, _lib.FLAG_TIMING)
dom = (-3.7190164854556804, 3.7190164854556804)
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(12_500_000, dtype=torch.float64, device="cuda", generator=g)
f = (x + 0.01 * torch.sqrt(1e-4 + x.abs())).contiguous()
torch.cuda.synchronize()
def sync(): _lib.check(_lib.lib().mlmc_synchronize())
import gc
gc.collect(); gc.disable()
for name, fn in (("Spline 128", Spline(128, dom)), ("Legendre 128", Legendre(128, dom)), ("Legendre 64", Legendre(64, dom))):
    for rep in range(6):
        t = [time.perf_counter()]
        acc = LevelAccumulator(fn, 1, LevelAccumulator.COV, mean_only=True); sync(); t.append(time.perf_counter())
        n, _, s, _ = acc.estimate([(0, f, None)], reduce=False); t.append(time.perf_counter())
        kt = acc.kernel_time()
        cov = (s[0] / n[0]).reshape(fn.size, fn.size)
        ortho, info = sd.construct_ortogonal_moments(fn, cov, tol=1e-4); t.append(time.perf_counter())
        acc2 = LevelAccumulator(ortho, 1, LevelAccumulator.MOMENTS, mean_only=True); sync(); t.append(time.perf_counter())
        n2, _, s2, _ = acc2.estimate([(0, f, None)], reduce=False); t.append(time.perf_counter())
        kt2 = acc2.kernel_time()
        means = s2[0] / n2[0]
        distr = sd.SimpleDistribution(ortho, np.stack([means, np.ones_like(means)], axis=1), domain=fn.domain)
        res = distr.estimate_density_minimize(tol=1e-8); t.append(time.perf_counter())
        acc.close(); acc2.close(); t.append(time.perf_counter())
        d = [1e3 * (b - a) for a, b in zip(t[:-1], t[1:])]
        print(f"{name} rep {rep}: create {d[0]:.2f}  cov estimate {d[1]:.2f} (kernels {kt[0]:.2f} ms / {kt[1]} launches)  ortho {d[2]:.2f}  create2 {d[3]:.2f}  "
              f"moments {d[4]:.2f} (kernels {kt2[0]:.2f} / {kt2[1]})  solve {d[5]:.2f}  close {d[6]:.2f}   n_ortho {ortho.size}", flush=True)
