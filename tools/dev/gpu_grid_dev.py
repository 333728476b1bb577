import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator
_lib.init(0, _lib.FLAG_TIMING)
dom = (-3.719, 3.719)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
blocks_list = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
n = 10_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
f = (x + 0.07 * torch.sqrt(1e-4 + x.abs())).contiguous(); c = (x + 0.5 * torch.sqrt(1e-4 + x.abs())).contiguous()
fn = Legendre(R, dom)
acc = LevelAccumulator(fn, 2)
for blocks in blocks_list:
    if blocks:
        os.environ["MLMC_HIP_DEV_BLOCKS"] = str(blocks)
    for it in range(3):
        acc.reset(); acc.push(1, f, c); r = acc.finalize()
    acc.reset()
    for it in range(5):
        acc.push(1, f, c)
    r = acc.finalize()
    ms, launches, nb = acc.kernel_time()
    acc.reset()
    for it in range(5):
        acc.push(0, f)
    r0 = acc.finalize()
    ms0, launches0, nb0 = acc.kernel_time()
    print(os.environ.get("MLMC_HIP_LIB", "default"), "R", R, "blocks", blocks, "pair us/launch", 1e3 * ms / launches, "level0 us/launch", 1e3 * ms0 / launches0, "n", r[0][1] // 5)
