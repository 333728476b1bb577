import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib
from mlmc_amd.sim import synth_device as sd
_lib.init(0)
for n in (10_000_000, 100_000_000):
    for rows in ([0], list(range(24))[: (24 if n <= 10_000_000 else 4)]):
        sd.generate_rows(1, 0, 1000, 0.1, 0.5, rows); _lib.lib().mlmc_synchronize()
        t0 = time.perf_counter()
        out = sd.generate_rows(1, 0, n, 0.1, 0.5, rows)
        _lib.lib().mlmc_synchronize()
        dt = time.perf_counter() - t0
        print(f"n {n:.0e} rows {len(rows):2d}: {dt*1e3:8.2f} ms  {n/dt:.3e} samples/s  {16*n*len(rows)/dt/1e9:7.1f} GB/s written")
        del out
