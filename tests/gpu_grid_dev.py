import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mlmc_amd import _lib, Legendre
from mlmc_amd.engine import LevelAccumulator
_lib.init(0, _lib.FLAG_TIMING)
dom = (-3.719, 3.719)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 10_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
f = (x + 0.07 * torch.sqrt(1e-4 + x.abs())).contiguous(); c = (x + 0.5 * torch.sqrt(1e-4 + x.abs())).contiguous()
fn = Legendre(R, dom)
acc = LevelAccumulator(fn, 2)
for blocks in (128, 256, 384, 512, 640, 768, 1024, 1536, 2048, 4096):
    os.environ["MLMC_HIP_DEV_BLOCKS"] = str(blocks)
    for it in range(3):
        acc.reset(); acc.push(1, f, c); r = acc.finalize()
    acc.reset()
    for it in range(5):
        acc.push(1, f, c)
    r = acc.finalize()
    ms, launches, nb = acc.kernel_time()
    print("R", R, "blocks", blocks, "us/launch", 1e3 * ms / launches, "n", r[0][1] // 5)
