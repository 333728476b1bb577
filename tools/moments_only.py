#!/usr/bin/env python3
"""Single-process driver for the rocprofv3 PMC passes of the moments kernels (tools/collect_profiles.sh): the stand-alone
mean + variance estimate of BASELINE configs[2]'s samples at R = 64 (term-split kernel) and the mean-only estimate of 127
moments (the pass behind the linearised covariance mean), 20 estimates each on device-generated samples."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator

_lib.init(0, _lib.FLAG_TIMING)
n, L = 10_000_000, 5
g = torch.Generator(device="cuda")
g.manual_seed(1)
steps = [0.5, 0.19, 0.07, 0.027, 0.01]
chunks = []
for l in range(L):
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    root = torch.sqrt(1e-4 + x.abs())
    chunks.append((l, (x + steps[l] * root).contiguous(), (x + steps[l - 1] * root).contiguous() if l else None))
torch.cuda.synchronize()
dom = (-3.7190164854556804, 3.7190164854556804)
for R, mean_only in ((64, False), (127, True)):
    acc = LevelAccumulator(Legendre(R, dom), L, LevelAccumulator.MOMENTS, mean_only=mean_only)
    for _ in range(5):
        acc.estimate(chunks, reduce=False)
    acc.kernel_time()
    for _ in range(20):
        acc.estimate(chunks, reduce=False)
    ms, launches, _ = acc.kernel_time()
    print("R = %d mean_only = %s: %.4f ms per estimate (%d launches)" % (R, mean_only, ms / 20, launches // 20))
    acc.close()
