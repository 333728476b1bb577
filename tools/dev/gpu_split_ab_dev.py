"""A/B of the term-split kernel's head / tail sizes on ONE box: the same estimate with libraries built with different
-DMLMC_SPLIT_HEAD, alternating (child processes, MLMC_HIP_LIB)."""
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
acc = LevelAccumulator(Legendre(64, (-3.719, 3.719)), 5)
for it in range(260):
    if it == 60: acc.kernel_time()
    acc.estimate(data, reduce=False)
ms, launches, nb = acc.kernel_time()
print("%%.4f" %% (ms / 200))
''' % root
libs = {"30/34 (shipped)": os.path.join(root, "mlmc_amd", "libmlmc_hip.so"), "32/32": os.path.join(root, "tools", "dev", "libmlmc_split32.so"),
        "28/36": os.path.join(root, "tools", "dev", "libmlmc_split28.so")}
if os.environ.get("LIBS"):   # LIBS="name=path,name=path" (paths relative to the repository root)
    libs = {kv.split("=", 1)[0]: os.path.join(root, kv.split("=", 1)[1]) for kv in os.environ["LIBS"].split(",")}
res = {k: [] for k in libs}
for rnd in range(3):
    for name, lib in libs.items():
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MLMC_HIP_LIB=lib), capture_output=True, text=True, timeout=300)
        res[name].append(float(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else float("nan"))
        print(name, res[name][-1], flush=True)
for name, v in res.items():
    print(name, "min %.4f ms  median %.4f ms" % (min(v), sorted(v)[len(v) // 2]))
