"""Interpreter vs compiled expression kernel on one box, alternating in one process (the switch is read at every evaluation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib
from mlmc_amd.quantity import lowering
from mlmc_amd.quantity.quantity import make_root_quantity
from tests.test_lowering import _spec, expression_zoo, make_storage
_lib.init(0, _lib.FLAG_TIMING)
os.environ["MLMC_EXPR_JIT_AFTER"] = "0"
st = make_storage((30, 20, 10))
root = make_root_quantity(st, _spec())
zoo = expression_zoo(root)
g = torch.Generator(device="cuda"); g.manual_seed(3)
for n in (10_000_000, 12_000_000):
    big = [torch.randn(n, 2, dtype=torch.float64, device="cuda", generator=g) for _ in range(4)]
    for name in ("leaf_scalar", "add_const", "mul_div", "central", "ufunc_binary", "select_gt"):
        plan = lowering.lower(zoo[name])
        rows = [big[i % 4] for i in range(len(plan.in_rows))]
        res = {}
        for rnd in range(3):
            for mode in ("0", "1"):
                os.environ["MLMC_EXPR_JIT"] = mode
                plan.kernel_time()
                for _ in range(5):
                    f, c, _ = plan.evaluate(rows, has_coarse=True, n=n, sync=True)
                ms, launches, nbytes = plan.kernel_time()
                res.setdefault(mode, []).append(ms / launches)
        b = 16.0 * n * (len(plan.in_rows) + plan.n_out)
        print(f"n {n} {name:14s} instr {len(plan.prog):3d} rows in {len(plan.in_rows)} out {plan.n_out}: interpreter {min(res['0']):.4f} ms ({b/min(res['0'])/1e9:.2f} TB/s)  compiled {min(res['1']):.4f} ms ({b/min(res['1'])/1e9:.2f} TB/s)", flush=True)
    del big
