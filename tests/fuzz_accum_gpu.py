"""Randomized validation, run by hand on the GPU box (`python tests/fuzz_accum_gpu.py`; pytest does not collect it):
accumulators vs the NumPy oracle over random shapes (N cases, SEED)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlmc_amd import _lib, Legendre, Monomial, Fourier
from mlmc_amd.engine import LevelAccumulator
from oracle import oracle_np as onp
from tests.util import level_arrays, to_chunks
_lib.init(0)
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
kinds = [(Legendre, onp.LEGENDRE), (Monomial, onp.MONOMIAL), (Legendre, onp.LEGENDRE)]   # the oracle's Fourier is 1-D only, like the reference's
bad = 0
t0 = time.time()
for it in range(int(os.environ.get("ITERS", 150))):
    cls, ok = kinds[rng.integers(3)]
    mode = LevelAccumulator.MOMENTS if rng.random() < 0.7 else LevelAccumulator.COV
    R = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 17, 24, 31, 32, 33, 40, 47, 48, 49, 56, 63, 64, 65, 80, 100, 128, 129, 140]))
    if mode == LevelAccumulator.COV:
        R = min(R, int(rng.choice([5, 16, 17, 32, 33, 64, 70, 129, 140])))       # > 128: the values path (64 x 64 blocks)
    if cls is Fourier and R > 128:
        R = 128
    L = int(rng.integers(1, 5))
    N = [int(rng.choice([1, 2, 63, 64, 65, 255, 257, 511, 1000, 4097, 20011])) for _ in range(L)]
    if mode == LevelAccumulator.COV and R > 70:
        N = [min(n, 1000) for n in N]                      # the reference-form oracle materialises [n, R, R]
    M = 1 if rng.random() < 0.8 else int(rng.integers(2, 4))
    nan_every = int(rng.choice([0, 0, 7, 101]))
    steps = [0.5 / (l + 1) for l in range(L)]
    levels = level_arrays(N, steps, M, nan_every, seed=int(rng.integers(1 << 30)))
    dom = (-3.0, 3.3) if cls is not Monomial else (-4.0, 4.5)
    safe = bool(rng.random() < 0.8) or mode == LevelAccumulator.COV
    log = bool(rng.random() < 0.15)
    if log:                                                # log-domain moments: samples e^x, domain edges inside the sample range
        levels = [(np.exp(f), None if c is None else np.exp(c)) for f, c in levels]
        dom = (float(np.exp(-2.5)), float(np.exp(2.7)))
    mean_only = bool(rng.random() < 0.3)
    fn = cls(R, dom, log=log, safe_eval=safe)
    b = onp.Basis(ok, R, dom, log=log, safe_eval=safe)
    resident = rng.random() < 0.5
    try:
        acc = LevelAccumulator(fn, L, mode, n_comp=M, mean_only=mean_only)
        if resident:
            import torch
            dev = torch.device("cuda", 0)
            keep = []
        for l, (f, c) in enumerate(levels):
            ff = f if M > 1 else f[0]
            cc = None if c is None else (c if M > 1 else c[0])
            if resident:
                ff = torch.from_numpy(np.ascontiguousarray(ff)).to(dev)
                cc = None if cc is None else torch.from_numpy(np.ascontiguousarray(cc)).to(dev)
                torch.cuda.synchronize()
                keep.append((ff, cc))
            acc.push(l, ff, cc)
        n, n_rm, s, sp = acc.finalize()
        acc.close()
        rows = onp.moments_rows if mode == LevelAccumulator.MOMENTS else onp.covariance_rows
        with np.errstate(all="ignore"):
            ref = onp.estimate_mean(to_chunks(levels), lambda x: rows(b, x))
        okc = np.array_equal(n, ref.n_samples) and np.array_equal(n_rm, ref.n_rm_samples)
        scale = np.sqrt(np.abs(ref.sums_sq) * np.maximum(ref.n_samples[:, None], 1)) + 1e-300
        fin = np.isfinite(ref.sums) & np.isfinite(ref.sums_sq)
        if mode == LevelAccumulator.COV:
            # a covariance row f_i f_j - c_i c_j is accumulated as (d_i s_j + s_i d_j) / 2: an entry far below the level's
            # typical row can lose relative accuracy (levels of one or two samples); gate it on the level's rms row scale
            lvl = np.sqrt(np.mean(np.where(fin, np.abs(ref.sums_sq), 0.0), axis=1, keepdims=True) * np.maximum(ref.n_samples[:, None], 1))
            scale = np.maximum(scale, lvl)
        e1 = np.max(np.abs(s - ref.sums)[fin] / np.maximum(np.abs(ref.sums), scale)[fin]) if fin.any() else 0.0
        # sums of squares: relative to the entry, with a floor of 1e-3 of the level's largest one (the covariance variance
        # comes from three Gram matrices whose terms can cancel in a single entry when a level holds a handful of samples)
        floor2 = 1e-3 * np.max(np.where(fin, np.abs(ref.sums_sq), 0.0), axis=1, keepdims=True) + 1e-300
        fin2 = fin & ~np.isnan(sp)                          # mean-only accumulators return NaN where the squares were skipped
        e2 = np.max(np.abs(sp - ref.sums_sq)[fin2] / np.maximum(np.abs(ref.sums_sq), floor2)[fin2]) if fin2.any() else 0.0
        if not okc or e1 > 1e-10 or e2 > 1e-9:
            bad += 1
            print("MISMATCH", cls.__name__, "mode", mode, "R", R, "N", N, "M", M, "nan", nan_every, "safe", safe, "log", log, "mean_only", mean_only,
                  "resident", resident, "counts", okc, e1, e2, flush=True)
    except Exception as e:
        if "All samples were masked" in str(e):
            continue
        bad += 1
        print("ERROR", cls.__name__, "mode", mode, "R", R, "N", N, "M", M, repr(e)[:200], flush=True)
print("done", it + 1, "cases", bad, "bad", round(time.time() - t0, 1), "s")
