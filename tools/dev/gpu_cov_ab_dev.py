"""A/B of covariance-kernel builds on ONE box: BASELINE configs[2] shape (5 levels x 1e7, R = 64), kernel time per estimate,
libraries alternating in child processes (MLMC_HIP_LIB)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, os
sys.path.insert(0, %r)
import torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator
_lib.init(0, _lib.FLAG_TIMING)
g = torch.Generator(device="cuda"); g.manual_seed(1)
n = 10_000_000
data = []
for l in range(5):
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    data.append((l, (x + 0.07 * torch.sqrt(1e-4 + x.abs())).contiguous(), None if l == 0 else (x + 0.5 * torch.sqrt(1e-4 + x.abs())).contiguous()))
out = []
for mean_only in (False, True):
    acc = LevelAccumulator(Legendre(int(os.environ.get("R", "64")), (-3.719, 3.719)), 5, LevelAccumulator.COV, mean_only=mean_only)
    for it in range(16):
        if it == 4: acc.kernel_time()
        acc.estimate(data, reduce=False)
    ms, launches, nb = acc.kernel_time()
    out.append(ms / 12)
    acc0 = LevelAccumulator(Legendre(int(os.environ.get("R", "64")), (-3.719, 3.719)), 1, LevelAccumulator.COV, mean_only=mean_only)
    for it in range(16):
        if it == 4: acc0.kernel_time()
        acc0.estimate(data[:1], reduce=False)
    out.append(acc0.kernel_time()[0] / 12)
print("%%.4f %%.4f %%.4f %%.4f" %% tuple(out))
''' % root
libs = {"new": os.path.join(root, "mlmc_amd", "libmlmc_hip.so"), "old": os.path.join(root, "tools", "dev", "libmlmc_covold.so")}
if os.environ.get("LIBS"):   # LIBS="name=path,name=path": other variants (paths relative to the repository root)
    libs = {kv.split("=", 1)[0]: os.path.join(root, kv.split("=", 1)[1]) for kv in os.environ["LIBS"].split(",")}
res = {k: [] for k in libs}
for rnd in range(3):
    for name, lib in libs.items():
        extra = {}
        if "@" in lib:   # "path@VAR=VAL": the same library under another environment switch
            lib, kv = lib.split("@")
            extra = dict([kv.split("=")])
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MLMC_HIP_LIB=lib, **extra), capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print(name, "FAILED", out.stderr[-500:]); continue
        res[name].append([float(v) for v in out.stdout.strip().splitlines()[-1].split()])
        print(name, res[name][-1], flush=True)
print("columns: 5-level estimate mean+var | level 0 alone mean+var | 5-level mean only | level 0 alone mean only  (kernel ms)")
for name, v in res.items():
    if v:
        print(name, "min", [round(min(c), 4) for c in zip(*v)])
