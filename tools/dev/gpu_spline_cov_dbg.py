import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle_np as onp
from tests.util import level_arrays, to_chunks
from mlmc_amd import _lib, Spline
from mlmc_amd.engine import LevelAccumulator, level_stats
_lib.init(0)
dom = (-3.7, 3.7)
levels = level_arrays([6001, 4000, 1501], [0.5, 0.07, 0.01], 1, 11)
for R, cut in ((24, 1200), (60, 900), (70, 700)):
    b = onp.Basis(onp.SPLINE, R, dom)
    lv = [(f[:, :cut], None if c is None else c[:, :cut]) for f, c in levels]
    acc = LevelAccumulator(Spline(R, dom), 3, LevelAccumulator.COV)
    for l, (f, c) in enumerate(lv):
        acc.push(l, np.ravel(f), None if c is None else np.ravel(c))
    n, n_rm, s, sp = acc.finalize()
    ref = onp.estimate_mean(to_chunks(lv), lambda x: onp.covariance_rows(b, x))
    lm, lvv = level_stats(n, s, sp)
    for l in range(3):
        num = np.abs(lvv[l] - ref.l_vars[l]); den = np.abs(ref.l_vars[l])
        rel = np.where(den > 0, num / np.maximum(den, 1e-300), num)
        k = int(np.argmax(rel))
        print(R, "level", l, "max rel err of l_vars %.3e at entry (%d,%d): got %.6e ref %.6e; level max %.3e; sums_sq abs err max %.3e" %
              (rel[k], k // R, k % R, lvv[l][k], ref.l_vars[l][k], den.max(), np.max(np.abs(sp[l] - ref.sums_sq[l]))))
