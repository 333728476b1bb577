"""Dev timing of the byte-code kernel: wall clock around N async launches + mlmc_synchronize."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib
from mlmc_amd.quantity import lowering
from mlmc_amd.quantity.quantity import make_root_quantity, Quantity
from mlmc_amd.quantity.quantity_spec import QuantitySpec
from mlmc_amd.sample_storage import Memory
_lib.init(0)
spec = [QuantitySpec(name="q", unit="", shape=(4, 1), times=[1], locations=['0'])]
st = Memory(); st.save_global_data(result_format=spec, level_parameters=[[0.5], [0.1]])
st.set_level_samples(0, np.ones((4, 4)), None); st.set_level_samples(1, np.ones((4, 4)), np.ones((4, 4)))
root = make_root_quantity(st, spec)['q'][1]['0']
x, y, z, w = root[0], root[1], root[2], root[3]
n = int(os.environ.get("N", 10_000_000))
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
rows = [torch.randn(n, 2, dtype=torch.float64, device=dev, generator=g) + 2.0 for _ in range(4)]
torch.cuda.synchronize()
trees = {"copy": x, "central": (x - 2.0) * (x - 2.0), "ratio": (x * y) / (np.abs(y) + 1.0),
         "deep": np.log1p(np.abs(np.tanh(x) * np.cos(y) + np.sqrt(np.square(y) + 1.0))) / (1.0 + np.exp2(np.negative(x))),
         "four_rows": root * 2.0 + 1.0, "select": x.select(x > 2.0, y < 3.0),
         "sum4": x + y + z + w}
trees["bench6"] = (x - 0.1) * (x - 0.1) / (np.abs(y) + 1.0)
trees["poly"] = ((x * 0.3 + 1.0) * x - 0.5) * x + (y * y - z) * (w + 2.0)
CHAIN = os.environ.get("CHAIN", "1") == "1"
for name, q in trees.items():
    plan = lowering.lower(q, chain=CHAIN)
    rr = [rows[r] for r in plan.in_rows]
    for _ in range(3):
        plan.evaluate(rr, True, n, sync=True)
    N = 20
    t0 = time.perf_counter()
    for _ in range(N):
        plan.evaluate(rr, True, n)
    _lib.lib().mlmc_synchronize()
    dt = (time.perf_counter() - t0) / N
    alg = 16.0 * n * (len(plan.in_rows) + plan.n_out)
    print(f"{name:10s} instr {len(plan.prog):3d} regs {plan.n_regs:2d} in {len(plan.in_rows)} out {plan.n_out}  {dt*1e3:7.3f} ms  {alg/dt/1e9:8.1f} GB/s (algorithmic)")
