#!/usr/bin/env python3
"""Headline benchmark: moment-evals/s of one complete MLMC estimate on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5|6]

Defaults (no --config):
  N = 1  BASELINE.json configs[2], the largest single-GPU configuration: 5 levels x 1e7 synthetic samples, Legendre
         n_moments = 64, moment covariance mean + variance, level-variance regression and n_samples re-allocation
         (reference: estimator.py:44-74,366-385).  The same JSON line carries three secondary blocks, each with its own
         roofline: "moments_r64" (the stand-alone mean+var estimate on the same samples), "configs1" (BASELINE
         configs[1]: 3 x 1e7, R = 32, mean+var) and "north_star" (1e8 samples x 64 moments, moments + covariance).
  N > 1  BASELINE.json configs[3]: 5 levels x 1e8 samples sharded over the N ranks (strong scaling of the fixed 5e8
         samples; launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`),
         one packed all-reduce (RCCL) of the [L, 2 + 2 R^2] partial sums per estimate.

A "step" = one complete estimate over data already resident in HBM: accumulator reset, push of every level (fused
transform + recurrence + mask + level-difference accumulation kernels), finalize (grid reduction, all-reduce for
N > 1, copy of the sums to the host) and the host formulas (level means / variances, MLMC totals; for the covariance
configurations also the regression over the levels and the re-allocation, with the level variances of the moments
read from row 0 of the covariance sums -- no second pass over the samples).
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

# the hosts of this pool support dmabuf IPC only: RCCL between the ranks of one node needs this before HIP starts (it is
# exported by the image; kept here for launchers that build their own environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (guides/MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X fp64 vector spec: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix spec
DOMAIN = (-3.7190164854556804, 3.7190164854556804)   # scipy.stats.norm().ppf([1e-4, 1 - 1e-4]) (test/test_run.py:71)

CONFIGS = {
    2: dict(L=3, n_per_level=10_000_000, R=32, mode="moments",
            workload="BASELINE configs[1]: 3 levels x 1e7 synthetic samples per GPU, Legendre n_moments=32, mean+var estimate"),
    3: dict(L=5, n_per_level=10_000_000, R=64, mode="cov",
            workload="BASELINE configs[2]: 5 levels x 1e7 synthetic samples per GPU, Legendre n_moments=64, moment covariance mean+var + level-variance regression + n_samples re-allocation"),
    4: dict(L=5, n_per_level=12_500_000, R=64, mode="cov",
            workload="BASELINE configs[3] per-GPU share: 5 levels x 1.25e7 synthetic samples per GPU (1e8 per level over 8 GPUs), "
                     "Legendre n_moments=64, moment covariance mean+var + level-variance regression + re-allocation as configs[2]; one all-reduce of the [L, 2 + 2 R^2] partial sums per estimate"),
    5: dict(L=1, n_per_level=12_500_000, R=128, mode="moments", basis="Spline",
            workload="BASELINE configs[4] per-GPU share: 1 level x 1.25e7 synthetic samples per GPU (1e8 over 8 GPUs), Spline "
                     "n_moments=128 (cubic B-spline moments, not part of the reference), mean+var estimate + max-entropy PDF"),
    6: dict(L=3, n_per_level=10_000_000, R=32, mode="tree",
            workload="SURVEY 8(f) row 1: derived quantity (x - 0.1)^2 / (|y| + 1) of two stored rows, evaluated by the byte-code "
                     "kernel (3 levels x 1e7 samples per GPU), then the Legendre n_moments=32 mean+var estimate"),
    # not selectable with --config: the N > 1 default and the north-star block of the N = 1 line
    "sharded": dict(L=5, n_total_per_level=100_000_000, R=64, mode="cov",
                    workload="BASELINE configs[3]: 5 levels x 1e8 synthetic samples sharded over the ranks, Legendre n_moments=64, "
                             "moment covariance mean+var + level-variance regression + re-allocation; ONE all-reduce (RCCL) of the "
                             "packed [L, 2 + 2 R^2] partial sums per estimate"),
    "north_star": dict(L=5, n_per_level=20_000_000, R=64, mode="cov",
                       workload="BASELINE north_star size: 1e8 synthetic samples (5 levels x 2e7) x 64 Legendre moments, "
                                "moments mean+var estimate and moment covariance mean+var, 1 GPU"),
}


SYNTH_BLOCK = 1_000_000


def synth_level_host_blocks(level, lo, hi, steps, seed=1234):
    """samples [lo, hi) of level `level` of the synthetic workload, generated on the HOST with NumPy's
    default_rng(seed + level) in blocks of 1e6 samples -- x ~ N(0,1); fine = x + h_l sqrt(1e-4+|x|); coarse = x + h_{l-1}
    sqrt(1e-4+|x|) (formula of mlmc/sim/synth_simulation.py:37-46) -- so that the GPU and the CPU baseline consume identical
    bits.  The stream of a level is ONE generator: a shard [lo, hi) draws (and drops) the blocks in front of it, so the
    shards of N ranks are exactly the slices of the one-GPU job.  Yields (offset - lo, fine, coarse | None)."""
    rng = np.random.default_rng(seed + level)
    pos = 0
    while pos < hi:
        m = min(SYNTH_BLOCK, hi - pos)
        x = rng.standard_normal(m)
        if pos + m > lo:
            a, b = max(lo - pos, 0), m
            x = x[a:b]
            root = np.sqrt(1e-4 + np.abs(x))
            yield pos + a - lo, x + steps[level] * root, (None if level == 0 else x + steps[level - 1] * root)
        pos += m


def synth_device(levels, lo, hi, steps, device, seed=1234):
    """The host-generated workload of `levels` (list of level indices), samples [lo, hi) of each, uploaded to HBM:
    [(fine, coarse | None)] of torch tensors.  One generator thread per level (NumPy's generators and ufuncs release the GIL)."""
    import threading
    import torch
    n = hi - lo
    out = {}
    for l in levels:
        out[l] = (torch.empty(n, dtype=torch.float64, device=device),
                  None if l == 0 else torch.empty(n, dtype=torch.float64, device=device))
    errors = []

    def work(l):
        try:
            torch.cuda.set_device(device)
            fine, coarse = out[l]
            for off, f, c in synth_level_host_blocks(l, lo, hi, steps, seed):
                fine[off:off + f.size].copy_(torch.from_numpy(f))
                if c is not None:
                    coarse[off:off + c.size].copy_(torch.from_numpy(c))
        except Exception as e:      # surfaced by the caller
            errors.append(e)

    threads = [threading.Thread(target=work, args=(l,)) for l in levels]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    torch.cuda.synchronize()
    return [out[l] for l in levels]


class Ctx:
    """What every measured block needs: ranks, device, collective handles."""

    def __init__(self, world, rank, dev, dist, torch):
        self.world, self.rank, self.dev, self.dist, self.torch = world, rank, dev, dist, torch

    def sync(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def _reduce(self, values, op):
        if self.world == 1:
            return list(values)
        on_gpu = self.dist.get_backend() == "nccl"
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.dev if on_gpu else "cpu")
        self.dist.all_reduce(t, op=op)
        return [float(v) for v in t]

    def max_over_ranks(self, values):
        return self._reduce(values, self.dist.ReduceOp.MAX)

    def sum_over_ranks(self, values):
        return self._reduce(values, self.dist.ReduceOp.SUM)


def preroll(step, ms_target, ctx):
    """Run `step` untimed for about ms_target (the same number of times on every rank: steps may hold a collective)."""
    if ms_target <= 0:
        return 0
    probe = 3
    ctx.torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(probe):
        step()
    ctx.torch.cuda.synchronize()
    per = ctx.max_over_ranks([(time.perf_counter() - t0) / probe])[0]
    count = min(int(ms_target / 1e3 / max(per, 1e-6)) + 1, 20000)
    for _ in range(count):
        step()
    return probe + count


def timed_loop(step, acc, steps, warmup, preroll_ms, ctx):
    """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; elapsed = max over ranks.
    -> (elapsed s, preroll steps, last result, [kernel ms, launches, algorithmic bytes] of `acc` inside the timed region)"""
    # no cyclic-GC pauses inside the timed region (a gen-2 collection of the torch-sized heap costs ~40 ms); collected
    # before the warm-up so that the timed steps follow the warm-up without an idle gap (an idle GPU drops its clock)
    gc.collect()
    gc.disable()
    try:
        pre = preroll(step, preroll_ms, ctx)
        for _ in range(warmup):
            step()
        acc.kernel_time()                               # drop the warm-up launches from the totals
        acc.kernel_flops()
        acc.aux_kernel_time()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
        ctx.sync()
        elapsed = time.perf_counter() - t0
    finally:
        gc.enable()
    # HIP events around every launch of the dominant kernel in the timed region, recorded on the library's stream and
    # read back once, after the clock has stopped
    kt = list(acc.kernel_time())
    mfma_flops = acc.kernel_flops()
    aux = acc.aux_kernel_time()
    elapsed, k_ms = ctx.max_over_ranks([elapsed, kt[0]])
    return elapsed, pre, res, [k_ms, int(kt[1]), int(kt[2]), int(mfma_flops), float(aux[0]), int(aux[1])]


def alg_flops(mode, R, pairs, singles):
    """Flops by the reference's operation count: 14 R per pair, 8 R per level-0 sample; the covariance adds
    6 R^2 (three R x R x n contractions, every entry of every matrix) per pair, 4 R^2 at level 0."""
    if mode == "moments":
        return (14 * R) * pairs + (8 * R) * singles
    return (14 * R + 6 * R * R) * pairs + (8 * R + 4 * R * R) * singles


def roofline_block(mode, basis, R, pairs_per_step, singles_per_step, kt, steps, config_key):
    """Roofline of the dominant kernel of one block.  The BINDING roof is the top-level one: fp64 VALU for the moments
    kernel at R >= 12 (28 flop/B at R = 32 against a ridge of 9.8), fp64 MFMA for the covariance kernel; the HBM figures
    (algorithmic bytes / average launch duration) ride beside it under "hbm".
    Covariance: `achieved` / `frac` count the flops the matrix pipe EXECUTES -- 512 per 16 x 16 tile and sample, the tiles
    of the symmetric Gram matrices once (SURVEY 8(d) allows symmetric halves), as reported by the library
    (mlmc_accum_kernel_flops) -- a physical fraction of the fp64 matrix peak.  At 17..128 polynomial moments the matrix
    cores compute the two Gram matrices of the VARIANCE only, and up to 64 moments only at the pair levels (R = 64: 16 + 10 =
    26 tiles per pair; with the mean's Gram matrix it was 42, and 20 per level-0 sample): the mean of the covariance comes
    from one mean-only launch of the moments kernel over 2 R - 1 terms, and level 0 -- one value per sample, so its second
    moments linearise as well -- from two launches over 4 R - 3 terms (product linearisation, mlmc_hip.h), reported beside
    it under "aux_kernel".  The
    reference-form count (6 R^2 + 14 R per pair: every entry of three R x R matrices) rides beside it under
    "reference_form"; it exceeds what any kernel that uses the symmetry has to execute, so its fraction can pass 1."""
    k_ms, launches, k_bytes, mfma_flops = kt[:4]
    aux_ms, aux_launches = (kt[4], kt[5]) if len(kt) > 4 else (0.0, 0)
    per_step = max(launches // max(steps, 1), 1)
    avg_launch_ms = k_ms / max(launches, 1)
    gbs = (k_bytes / 1e9) / (k_ms / 1e3) if k_ms > 0 else 0.0
    hbm = {"achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
    common = {"avg_launch_ms": round(avg_launch_ms, 5), "alg_bytes_per_launch": int(k_bytes / max(launches, 1)),
              "launches_per_step": per_step}
    if basis == "Spline":
        kname = "k_spline_accum"
        return dict(bound="hbm", **hbm, **pmc_traffic(config_key, kname), kernel=kname, **common,
                    note="sparse accumulation (<= 8 of the R sums touched per sample pair): no dense flop count applies")
    flops = alg_flops(mode, R, pairs_per_step, singles_per_step)
    step_kernel_s = (k_ms / 1e3) / max(steps, 1)
    tflops = flops / step_kernel_s / 1e12 if step_kernel_s > 0 else 0.0
    if mode == "moments":
        kname, peak = "k_moments_accum", FP64_VALU_PEAK_TFLOPS
        if R < 12:                                      # HBM-bound below the ridge
            return dict(bound="hbm", **hbm, **pmc_traffic(config_key, kname), kernel=kname, **common)
        return dict(bound="valu_f64", achieved=round(tflops, 3), peak=peak, unit="TFLOP/s", frac=round(tflops / peak, 4),
                    **pmc_traffic(config_key, kname), kernel=kname, alg_flops_per_step=int(flops), hbm=hbm, **common)
    kname, peak = "k_cov_accum", FP64_MFMA_PEAK_TFLOPS
    ex = mfma_flops / max(steps, 1)
    ex_tf = ex / step_kernel_s / 1e12 if step_kernel_s > 0 else 0.0
    aux = {}
    if aux_launches:
        # the covariance MEAN comes from the level sums of the 2 R - 1 moments of the product linearisation: one mean-only
        # launch of the moments kernel per estimate beside the matrix-core launches (which then execute G1, G2 only)
        # ... and at level 0 (<= 64 moments) the second moments too, from 4 R - 3 moments in two windows: no matrix pass there
        K = 2 * R - 1
        K0 = 4 * R - 3 if R <= 64 else K
        # flops per term: recurrence (multiply + FMA = 3) per value, difference 1, sum 1 -> 8 per pair, 4 per level-0 sample
        aux_flops = (8 * K) * pairs_per_step + (4 * K0) * singles_per_step
        aux_s = aux_ms / 1e3 / max(steps, 1)
        aux = {"aux_kernel": {"kernel": "k_moments_accum_split (mean-only: %d terms over the pair levels = product linearisation of the "
                                        "covariance mean; %d terms at level 0 = mean and second moments there)" % (K, K0),
                              "ms_per_step": round(1e3 * aux_s, 4), "launches_per_step": aux_launches // max(steps, 1),
                              "bound": "valu_f64", "achieved": round(aux_flops / aux_s / 1e12, 3) if aux_s > 0 else 0.0,
                              "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(aux_flops / aux_s / 1e12 / FP64_VALU_PEAK_TFLOPS, 4) if aux_s > 0 else 0.0}}
    pipe = pmc_fp64_pipe(config_key, kname)
    if pipe is not None:
        aux["fp64_pipe"] = pipe
    return dict(**aux, bound="mfma", achieved=round(ex_tf, 3), peak=peak, unit="TFLOP/s", frac=round(ex_tf / peak, 4),
                **pmc_traffic(config_key, kname), kernel=kname, executed_mfma_flops_per_step=int(ex),
                reference_form={"alg_flops_reference_form": int(flops), "achieved": round(tflops, 3), "frac": round(tflops / peak, 4),
                                "note": "6 R^2 + 14 R per pair (4 R^2 + 8 R at level 0), quantity_estimate.py:131-147: every entry of "
                                        "every Gram matrix; the kernel computes symmetric tiles once and executes fewer"},
                hbm=hbm, **common)


def host_formulas(n, s, sp, level_stats):
    l_means, l_vars = level_stats(n, s, sp)
    with np.errstate(all="ignore"):
        return l_means, l_vars, np.sum(l_means, axis=0), np.sum(l_vars / n[:, None], axis=0)


def estimate_block(cfg, data, fn, steps_h, ctx, steps, warmup, preroll_ms, config_key, mean_only=False):
    """One measured workload on resident data -> result dict (value, ms_per_step, roofline, result_check, ...).
    cfg["mode"] == "cov": covariance estimate + regression + re-allocation, the moments' level variances taken from the
    covariance sums (engine.moments_from_covariance)."""
    from mlmc_amd.engine import LevelAccumulator, level_stats, moments_from_covariance
    from mlmc_amd.estimator import Estimate, estimate_n_samples_for_target_variance
    L, R = cfg["L"], cfg["R"]
    mode = LevelAccumulator.MOMENTS if cfg["mode"] == "moments" else LevelAccumulator.COV
    acc = LevelAccumulator(fn, L, mode, mean_only=mean_only)
    chunks = [(l, data[l][0], data[l][1]) for l in range(L)]
    n_ops = [(1.0 / h) ** 2 * np.log(max(1.0 / h, 2.0)) for h in steps_h]      # synth_simulation.py:133-134
    regress = Estimate(None, None, fn)._all_moments_variance_regression
    extra = {}

    def one_estimate():
        # reset + push of every level + finalize: one call of the C ABI; with more than one rank the packed partial sums
        # stay on the device and go through ONE all-reduce before the host reads them
        n, n_rm, s, sp = acc.estimate(chunks)
        _, _, mean, var = host_formulas(n, s, sp, level_stats)
        if cfg["mode"] == "cov":
            s_m, sp_m = moments_from_covariance(s, sp, R)
            _, raw_vars = level_stats(n, s_m, sp_m)
            reg_vars = regress(raw_vars, np.array(steps_h))
            extra["n_estimated"] = [int(v) for v in estimate_n_samples_for_target_variance(1e-6, reg_vars, n_ops, n_levels=L)]
        return n, n_rm, mean, var

    elapsed, pre, res, kt = timed_loop(one_estimate, acc, steps, warmup, preroll_ms, ctx)
    n, n_rm, mean, var = res
    n_local = [int(data[l][0].shape[0]) for l in range(L)]
    samples_per_step = int(ctx.sum_over_ranks([float(sum(n_local))])[0])      # whole job: every rank's shard
    pairs, singles = sum(n_local[1:]), n_local[0]         # this rank's launches
    out = {
        "value": samples_per_step * R * steps / elapsed, "unit": "moment-evals/s", "steps": steps, "warmup": warmup,
        "preroll_steps": pre, "ms_per_step": 1e3 * elapsed / steps,
        "roofline": roofline_block(cfg["mode"], cfg.get("basis", "Legendre"), R, pairs, singles, kt, steps, config_key),
        "result_check": {"mean0": float(np.ravel(mean)[0]), "var0": float(np.ravel(var)[0]), "n_removed": [int(v) for v in n_rm]},
    }
    out["result_check"].update(extra)
    acc.close()
    return out


def exchange_block(fn, L, R, ctx, reps=50):
    """The exchange step alone: `reps` all-reduces of a packed [2 L + 2 L R^2] fp64 buffer (what one sharded covariance
    estimate sends), barrier + synchronize on both sides, max over ranks."""
    from mlmc_amd.engine import allreduce_partials
    torch = ctx.torch
    on_gpu = ctx.dist.get_backend() == "nccl"
    k = 2 * L + 2 * L * R * R
    buf = torch.zeros(k, dtype=torch.float64, device=ctx.dev if on_gpu else "cpu")
    for _ in range(5):
        allreduce_partials(buf)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        allreduce_partials(buf)
    ctx.sync()
    ms = ctx.max_over_ranks([1e3 * (time.perf_counter() - t0) / reps])[0]
    return {"collective": "all-reduce (sum) of the packed fp64 partials n | n_rm | s | sp", "backend": ctx.dist.get_backend(),
            "bytes_per_rank": 8 * k, "allreduce_ms": round(ms, 4),
            "note": "all-reduce + copy of the reduced sums to the host, timed alone; inside a step it follows the kernels on the same stream"}


def h2d_inclusive_block():
    """the first estimate over samples that still sit in a host storage, delivered the way SampleStorageHDF
    delivers them -- 300 chunks of 1e5 samples (3 levels x 1e7), every chunk a freshly allocated [n, 2, M] array
    (Memory(copy_chunks=True) stands in for the HDF5 reads; h5py is not part of the image).  `stream` (the default feed,
    quantity_estimate._LevelStreamer): helper threads read the chunks of a level into pinned staging blocks of 32 MB, one
    asynchronous DMA per block, the level lands as ONE device tensor and the tree runs as one launch per level; `sync`: read,
    upload, launch chunk by chunk (MLMC_HIP_STREAM_UPLOAD=0); `resident`: the next estimate, from HBM.  PCIe-inclusive
    figures are never the headline `value`."""
    from mlmc_amd import Legendre
    from mlmc_amd.estimator import Estimate, determine_level_parameters
    from mlmc_amd.quantity import quantity_estimate as qe
    from mlmc_amd.quantity.quantity import make_root_quantity
    from mlmc_amd.quantity.quantity_spec import QuantitySpec
    from mlmc_amd.sample_storage import Memory
    L, n_l, chunk, R = 3, 10_000_000, 100_000, 32
    steps_h = [s[0] for s in determine_level_parameters(L, [0.5, 0.01])]
    spec = [QuantitySpec(name="q", unit="", shape=(1, 1), times=[1], locations=['0'])]
    st = Memory(chunk_size=chunk, copy_chunks=True)
    st.save_global_data(result_format=spec, level_parameters=[[h] for h in steps_h])
    for l in range(L):
        x = np.random.default_rng(99 + l).standard_normal(n_l)
        root = np.sqrt(1e-4 + np.abs(x))
        st.set_level_samples(l, x + steps_h[l] * root, None if l == 0 else x + steps_h[l - 1] * root)
    q = make_root_quantity(st, spec)['q'][1]['0'][0, 0]
    est = Estimate(q, st, Legendre(R, DOMAIN))
    out = {"workload": "3 levels x 1e7 samples in a host storage, 300 chunks of 1e5 samples, Legendre n_moments=32 mean+var estimate",
           "chunks": L * (n_l // chunk), "host_bytes": int(L * n_l * 16)}
    saved = os.environ.get("MLMC_HIP_STREAM_UPLOAD")
    results = {}
    try:
        for mode, flag, into in (("stream", "1", "1"), ("stream_arrays", "1", "0"), ("sync", "0", "0")):
            # stream (the default feed): the storage writes every chunk straight into the pinned staging block
            # (Memory.sample_records_into -- what an HDF5 storage does with Dataset.read_direct): one host copy per chunk;
            # stream_arrays: every chunk arrives as a freshly allocated array (the reference's storage interface,
            # sample_pairs_level) and is copied into the staging block: two host copies and freshly faulted pages per chunk
            os.environ["MLMC_HIP_STREAM_UPLOAD"] = flag
            os.environ["MLMC_HIP_STREAM_READ_INTO"] = into
            times = []
            for _ in range(3):
                qe.device_cache_clear()
                t0 = time.perf_counter()
                results[mode] = est.estimate_moments()
                times.append(time.perf_counter() - t0)
            out[mode + "_ms"] = round(1e3 * min(times), 3)
            out[mode + "_gbs"] = round(out["host_bytes"] / min(times) / 1e9, 2)
            out[mode + "_evals_per_s"] = L * n_l * R / min(times)
        t0 = time.perf_counter()
        for _ in range(5):
            warm = est.estimate_moments()
        out["resident_ms"] = round(1e3 * (time.perf_counter() - t0) / 5, 3)
        out["stream_threads"] = qe._LevelStreamer.n_threads()
        out["stream_block_mb"] = qe._LevelStreamer.block_doubles() * 8 / 2 ** 20
        out["same_result"] = bool(all(np.array_equal(a, b) for r in ("sync", "stream_arrays") for a, b in zip(results["stream"], results[r]))
                                  and all(np.array_equal(a, b) for a, b in zip(results["stream"], warm)))
    finally:
        os.environ.pop("MLMC_HIP_STREAM_READ_INTO", None)
        if saved is None:
            os.environ.pop("MLMC_HIP_STREAM_UPLOAD", None)
        else:
            os.environ["MLMC_HIP_STREAM_UPLOAD"] = saved
        qe.device_cache_clear()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=lambda v: int(v) if v.isdigit() else v, default=None, choices=[2, 3, 4, 5, 6, "sharded"],
                    help="default: 3 (BASELINE configs[2]) + secondary blocks at N = 1, configs[3] sharded at N > 1; "
                         "`--gpus 1 --config sharded` runs the WHOLE configs[3] job (5 x 1e8 samples, 8 GB) on one GPU: the "
                         "N = 1 point of the same strong-scaling workload the N > 1 default measures")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N = 1 default line without the secondary blocks")
    ap.add_argument("--preroll-ms", type=float, default=250.0,
                    help="untimed load before the W warm-up steps: the shader clock of an idle MI355X needs ~70 ms of "
                         "continuous work to reach its steady state (0 = none)")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: native libraries (the RCCL banner, amdgpu notices) write to fd 1 directly,
    # so fd 1 points at stderr until the result is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("MLMC_HIP_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # RCCL logs to stdout by default; stdout carries the JSON line
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # MLMC_BENCH_BACKEND=gloo / MLMC_BENCH_DEVICE=0: rehearse the multi-rank flow on a one-GPU box (tests only)
        backend = os.environ.get("MLMC_BENCH_BACKEND", "nccl")
        if "MLMC_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["MLMC_BENCH_DEVICE"])
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node {}".format(args.gpus)

    from mlmc_amd import _lib, Legendre, Spline
    from mlmc_amd.engine import shard_bounds
    from mlmc_amd.estimator import determine_level_parameters

    _lib.init(local_rank, _lib.FLAG_TIMING)
    dev = torch.device("cuda", local_rank)
    ctx = Ctx(world, rank, dev, dist, torch)

    def finish(out):
        if dist.is_initialized():
            dist.barrier()
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist.is_initialized():
            dist.destroy_process_group()

    key = args.config
    if key is None:
        key = 3 if world == 1 else "sharded"
    cfg = dict(CONFIGS[key])
    if cfg["mode"] == "tree":
        return finish(tree_bench(args, cfg, world, rank, dev, dist))
    L, R = cfg["L"], cfg["R"]
    if key == "sharded":
        # MLMC_BENCH_TOTAL_PER_LEVEL: smaller totals for rehearsals on one GPU (tests); the default is BASELINE's 1e8
        total = int(os.environ.get("MLMC_BENCH_TOTAL_PER_LEVEL", cfg["n_total_per_level"]))
        lo, hi = shard_bounds(total, rank, world)
        cfg["n_per_level"] = hi - lo
        cfg["n_total_per_level"] = total
    else:
        lo, hi = 0, cfg["n_per_level"]
    n_l = cfg["n_per_level"]
    steps_h = [s[0] for s in determine_level_parameters(L, [0.5, 0.01])] if L > 1 else [0.01]
    fn = Spline(R, DOMAIN) if cfg.get("basis") == "Spline" else Legendre(R, DOMAIN)
    # host-generated default_rng(1234 + l) samples; sharded: this rank's slice of the one stream per level,
    # otherwise (weak scaling, fixed work per GPU) every rank its own stream
    data_seed = 1234 if key == "sharded" else 1234 + 1000 * rank
    data = synth_device(list(range(L)), lo, hi, steps_h, dev, seed=data_seed)

    head = estimate_block(cfg, data, fn, steps_h, ctx, args.steps, args.warmup, args.preroll_ms, key)
    out = {
        "metric": "moment-evals/sec (samples x n_moments)", "value": head["value"], "unit": "moment-evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preroll_steps": head["preroll_steps"],
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong" if key == "sharded" else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": cfg["workload"], "levels": L, "samples_per_level_per_gpu": n_l, "n_moments": R,
                   "basis": cfg.get("basis", "Legendre"), "estimate": cfg["mode"],
                   "exchange": "all-reduce of [L,(2+2K)] partial sums" if world > 1 else "none"},
        "roofline": head["roofline"], "result_check": head["result_check"],
    }
    if key == "sharded":
        out["config"]["samples_per_level_total"] = cfg["n_total_per_level"]
    if dist.is_initialized() and cfg["mode"] == "cov":
        out["exchange"] = exchange_block(fn, L, R, ctx)

    # ---- secondary blocks of the default N = 1 line -----------------------------------------------------------------
    if args.config is None and world == 1 and not args.no_secondary:
        # the stand-alone mean + variance estimate on the same samples (R = 64): the fp64-VALU figure beside the MFMA one
        mcfg = dict(cfg, mode="moments", workload="configs[2] samples, stand-alone Legendre n_moments=64 mean+var estimate")
        blk = estimate_block(mcfg, data, fn, steps_h, ctx, 50, 10, 0.0, 3)
        out["moments_r64"] = dict(blk, config={"workload": mcfg["workload"], "levels": L, "samples_per_level_per_gpu": n_l, "n_moments": R})
        if "aux_kernel" in head["roofline"]:
            # the same estimate with ALL THREE Gram matrices on the matrix cores (the headline's form before the mean was
            # linearised): more matrix work per sample at a higher matrix-pipe fraction, and a longer estimate
            prev = os.environ.get("MLMC_HIP_LINEARIZE")
            os.environ["MLMC_HIP_LINEARIZE"] = "0"
            try:
                blk = estimate_block(cfg, data, fn, steps_h, ctx, 10, 3, 0.0, None)
            finally:
                if prev is None:
                    del os.environ["MLMC_HIP_LINEARIZE"]
                else:
                    os.environ["MLMC_HIP_LINEARIZE"] = prev
            r3 = blk["roofline"]
            out["three_gram_form"] = {
                "ms_per_step": blk["ms_per_step"], "value": blk["value"], "unit": blk["unit"], "steps": blk["steps"], "warmup": blk["warmup"],
                "roofline": {k: r3[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "executed_mfma_flops_per_step",
                                                "avg_launch_ms", "launches_per_step")},
                "note": "MLMC_HIP_LINEARIZE=0: covariance mean as a third Gram matrix and level 0 on the matrix cores (42 tiles per pair, "
                        "20 per level-0 sample, instead of 26 / 0 + the moments launches); same samples, same outputs"}
            out["roofline"]["note"] = ("the matrix cores accumulate the two Gram matrices of the covariance's VARIANCE at the pair levels; its "
                                       "MEAN and all of level 0 come from the aux_kernel launches (product linearisation).  three_gram_form "
                                       "in this line = the same estimate with everything on the matrix cores: higher matrix-pipe fraction, "
                                       "longer estimate")
    # ---- max-entropy PDF solve time (second half of BASELINE's metric), outside the timed region, rank 0 ------------
    if rank == 0:
        from mlmc_amd.engine import LevelAccumulator, level_stats
        out["pdf_solve"] = pdf_solve_timing(fn, L, data, LevelAccumulator, level_stats, steps_h, through_api=(world == 1))
    # ---- CPU baseline + parity gate on a bounded sample (rank 0, N = 1 only) ---------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from mlmc_amd.engine import LevelAccumulator, level_stats
        from oracle import oracle_np as onp            # the checker: only this leg touches oracle/
        out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(cfg, fn, DOMAIN, steps_h, onp, LevelAccumulator, level_stats)
    if args.config is None and world == 1 and not args.no_secondary:
        del data
        torch.cuda.empty_cache()
        for name, k2, st, wu in (("configs1", 2, 100, 20), ("north_star", "north_star", 3, 1)):
            c2 = CONFIGS[k2]
            sh = [s[0] for s in determine_level_parameters(c2["L"], [0.5, 0.01])]
            f2 = Legendre(c2["R"], DOMAIN)
            d2 = synth_device(list(range(c2["L"])), 0, c2["n_per_level"], sh, dev)
            blk = estimate_block(c2, d2, f2, sh, ctx, st, wu, 100.0, 3 if k2 == "north_star" else k2)
            blk["config"] = {"workload": c2["workload"], "levels": c2["L"], "samples_per_level_per_gpu": c2["n_per_level"],
                             "n_moments": c2["R"], "estimate": c2["mode"]}
            if k2 == "north_star":
                # the north star names moments AND covariance: the stand-alone moments estimate of the same 1e8 samples
                m2 = estimate_block(dict(c2, mode="moments"), d2, f2, sh, ctx, 20, 5, 0.0, 3)
                blk["moments"] = {k: m2[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "roofline")}
                blk["both_ms"] = blk["ms_per_step"] + m2["ms_per_step"]
                # the MEANS of moments and covariance alone (what Estimate.construct_density reads): one mean-only pass of
                # 2 R - 1 moments, the covariance means by product linearisation (mlmc_amd/linearize.py) -- no variances
                from mlmc_amd.quantity import quantity_estimate as qe_
                ext2 = qe_._linearized_basis(f2)
                if ext2 is not None:
                    m3 = estimate_block(dict(c2, mode="moments", R=ext2.size), d2, ext2, sh, ctx, 20, 5, 0.0, 3, mean_only=True)
                    blk["means_only"] = {"ms_per_step": m3["ms_per_step"], "n_moments_pass": ext2.size,
                                         "hbm_frac": round((c2["L"] * 2 - 1) * c2["n_per_level"] * 8.0 / (m3["ms_per_step"] / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
                                         "note": "level means of the 64 moments AND of their 64 x 64 covariance from ONE pass of 127 moments "
                                                 "(6 fp64 instructions per term and pair: vector-pipe bound, not HBM)"}
                blk["hbm_frac_both"] = round((c2["L"] * 2 - 1) * c2["n_per_level"] * 8.0 * 2 / (blk["both_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS, 5)
                blk["note"] = ("north_star asks for >= 60 % of the HBM roofline on moments + covariance at R = 64; in fp64 the "
                               "covariance is matrix-core bound (>= 1000 flop/B) and the moments pass VALU bound (56 flop/B): the "
                               "binding roofs are reported, the HBM fraction of both passes is hbm_frac_both (SURVEY 8(d), fact 9)")
            out[name] = blk
            del d2
            torch.cuda.empty_cache()
        out["h2d_inclusive"] = h2d_inclusive_block()
    finish(out)


def tree_bench(args, cfg, world, rank, dev, dist):
    """--config 6: a Quantity tree over two stored rows, lowered to a register program and evaluated by k_expr
    (mlmc_amd/csrc/expr.hip), feeding the moments estimate.  Step = evaluation of every level's chunk + one estimate."""
    import torch
    from mlmc_amd import Legendre
    from mlmc_amd.engine import LevelAccumulator, level_stats
    from mlmc_amd.quantity import lowering
    from mlmc_amd.quantity.quantity import make_root_quantity
    from mlmc_amd.quantity.quantity_spec import QuantitySpec
    from mlmc_amd.sample_storage import Memory
    from mlmc_amd.estimator import determine_level_parameters
    L, n_l, R = cfg["L"], cfg["n_per_level"], cfg["R"]
    ctx = Ctx(world, rank, dev, dist, torch)
    steps = [s[0] for s in determine_level_parameters(L, [0.5, 0.01])]
    # the tree, built through the reference-style API over a (tiny) storage with the same two stored rows
    spec = [QuantitySpec(name="q", unit="", shape=(2, 1), times=[1], locations=['0'])]
    st = Memory()
    st.save_global_data(result_format=spec, level_parameters=[[h] for h in steps])
    for l in range(L):
        st.set_level_samples(l, np.ones((2, 2)), np.ones((2, 2)) if l else None)
    root = make_root_quantity(st, spec)['q'][1]['0']
    x, y = root[0], root[1]
    q = (x - 0.1) * (x - 0.1) / (np.abs(y) + 1.0)
    plan = lowering.lower(q)
    dom = (0.0, 12.0)
    fn = Legendre(R, dom)
    acc = LevelAccumulator(fn, L, LevelAccumulator.MOMENTS)
    # stored rows in the storage layout: interleaved (fine, coarse) pairs [n, 2]; level 0: [n, 1]
    stored = []
    for l in range(L):
        rows = []
        for r in range(2):
            f, c = synth_device([l], 0, n_l, steps, dev, seed=1234 + 1000 * rank + 77 * r)[0]
            rows.append(f.reshape(-1, 1).contiguous() if c is None else torch.stack([f, c], dim=1).contiguous())
        stored.append(rows)
    torch.cuda.synchronize()

    def one_step():
        acc.reset()
        keep = []
        for l in range(L):
            f, c, _ = plan.evaluate(stored[l], has_coarse=(l > 0), n=n_l)
            keep.append((f, c))
            acc.push(l, f[0], None if c is None else c[0])
        n, n_rm, s, sp = acc.finalize()
        _, _, mean, var = host_formulas(n, s, sp, level_stats)
        return n, n_rm, mean, var

    plan.kernel_time()
    steps_k, warm = args.steps, args.warmup
    gc.collect()
    gc.disable()
    preroll_steps = preroll(one_step, args.preroll_ms, ctx)
    for _ in range(warm):
        one_step()
    plan.kernel_time()
    acc.kernel_time()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps_k):
        n, n_rm, mean, var = one_step()
    ctx.sync()
    elapsed = ctx.max_over_ranks([time.perf_counter() - t0])[0]
    gc.enable()
    x_ms, x_launches, x_bytes = plan.kernel_time()
    a_ms, a_launches, _ = acc.kernel_time()
    gbs = (x_bytes / 1e9) / (x_ms / 1e3) if x_ms > 0 else 0.0
    traffic = pmc_traffic(6, "k_expr")
    out = {
        "metric": "moment-evals/sec (samples x n_moments)", "value": world * L * n_l * R * steps_k / elapsed,
        "unit": "moment-evals/s", "n_gpus": world, "steps": steps_k, "warmup": warm, "preroll_steps": preroll_steps,
        "ms_per_step": 1e3 * elapsed / steps_k, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": cfg["workload"], "levels": L, "samples_per_level_per_gpu": n_l, "n_moments": R,
                   "tree": "(x - 0.1) * (x - 0.1) / (np.abs(y) + 1.0)", "program_instructions": len(plan.prog),
                   "program_registers": plan.n_regs, "stored_rows_read": len(plan.in_rows), "result_rows": plan.n_out},
        "roofline": {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(gbs / HBM_PEAK_GBS, 4), **traffic, "kernel": "k_expr",
                     "avg_launch_ms": round(x_ms / max(x_launches, 1), 5),
                     "alg_bytes_per_launch": int(x_bytes / max(x_launches, 1)), "launches_per_step": x_launches // max(steps_k, 1),
                     "moments_kernel_ms_per_step": round(a_ms / max(steps_k, 1), 5)},
        "result_check": {"mean0": float(mean[0]), "var0": float(var[0]), "n_removed": [int(v) for v in n_rm]},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU leg: the same tree evaluated the way the reference does it (NumPy closures over [M, n, 2] chunks,
        # mlmc/quantity/quantity.py) on a bounded sample, and the parity of the device rows against it
        m = 2_000_000
        host_rows = np.stack([stored[1][0][:m].cpu().numpy(), stored[1][1][:m].cpu().numpy()])     # [2, m, 2]
        st2 = Memory()
        st2.save_global_data(result_format=spec, level_parameters=[[h] for h in steps])
        st2.set_level_samples(0, host_rows[:, :4, 0].T.copy(), None)
        st2.set_level_samples(1, host_rows[:, :, 0].T.copy(), host_rows[:, :, 1].T.copy())
        root2 = make_root_quantity(st2, spec)['q'][1]['0']
        q2 = (root2[0] - 0.1) * (root2[0] - 0.1) / (np.abs(root2[1]) + 1.0)
        chunk = [c for c in st2.chunks() if c.level_id == 1][0]
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            from mlmc_amd.quantity import quantity as qmod
            qmod.cache_clear()
            want = q2.samples(chunk)
        dt = (time.perf_counter() - t0) / reps
        f, c, _ = plan.evaluate([stored[1][0][:m].contiguous(), stored[1][1][:m].contiguous()], True, m, sync=True)
        got = np.stack([f.cpu().numpy(), c.cpu().numpy()], axis=2)
        out["cpu_baseline"] = {"value": m / dt, "unit": "samples/s (tree evaluation only)", "cores": 1, "kind": "port",
                               "sample": "NumPy evaluation of the same tree on one level-1 chunk of 2e6 sample pairs (reference "
                                         "algorithm: one temporary per node)", "gpu_samples_per_s": n_l * x_launches / (x_ms / 1e3) if x_ms else None}
        out["parity"] = {"rows_bit_exact": bool(np.array_equal(got, want)), "max_abs_err": float(np.max(np.abs(got - want))),
                         "ok": bool(np.array_equal(got, want))}
    return out


def pdf_solve_timing(fn, L, data, LevelAccumulator, level_stats, steps_h=None, through_api=True):
    """Estimate.construct_density on this rank's HBM-resident samples (estimator.py:304-331): covariance mean -> orthogonal
    moments (host LAPACK, R x R) -> means of the orthogonal moments -> max-entropy Newton solve on the device.  Only the
    means of the two estimates are used (as in the reference).
    through_api (N = 1): the statements of `Estimate.construct_density` itself, through the product's Python API
    (`estimate_mean(covariance(q, fn), variance=False)` ...) over a `DeviceMemory` storage that holds the samples where they
    are -- quantity tree, resident-sample cache, pooled accumulators and `QuantityMean` included.  Otherwise (rank 0 of a
    multi-rank job, where the API's estimates would wait for the other ranks' all-reduce): the same estimates rank-locally
    on `LevelAccumulator`s, the way quantity_estimate._estimate_mean runs them for this basis.
    Reports the solve alone and the estimates ("chain"), each for the first (cold) and the second call."""
    import torch
    from mlmc_amd import linearize
    from mlmc_amd.quantity import quantity_estimate as qe
    from mlmc_amd.tool import simple_distribution as sd
    ext = qe._linearized_basis(fn)     # the product's own rule: Legendre / monomial / Fourier
    accs = {}
    if through_api:
        from mlmc_amd.quantity.quantity import make_root_quantity
        from mlmc_amd.quantity.quantity_spec import QuantitySpec
        from mlmc_amd.sample_storage import DeviceMemory
        spec = [QuantitySpec(name="q", unit="", shape=(1, 1), times=[1], locations=['0'])]
        st = DeviceMemory()
        st.save_global_data(result_format=spec, level_parameters=[[h] for h in (steps_h or [1.0] * L)])
        for l in range(L):
            f, c = data[l]
            st.set_level_samples(l, (f.reshape(1, -1, 1) if c is None else torch.stack([f, c], dim=1)[None]))
        torch.cuda.synchronize()
        q = make_root_quantity(st, spec)['q'][1]['0'][0, 0]

        def chain():
            t0 = time.perf_counter()
            cov = qe.estimate_mean(qe.covariance(q, fn), variance=False).mean
            ortho, info = sd.construct_ortogonal_moments(fn, cov, tol=1e-4)
            means = qe.estimate_mean(qe.moments(q, ortho), variance=False).mean
            t1 = time.perf_counter()
            distr = sd.SimpleDistribution(ortho, np.stack([means, np.ones_like(means)], axis=1), domain=ortho.domain)
            res = distr.estimate_density_minimize(1e-8, 0.0)
            t2 = time.perf_counter()
            return 1e3 * (t1 - t0), 1e3 * (t2 - t1), ortho, res
    else:
        chunks = [(l, data[l][0], data[l][1]) for l in range(L)]

        def chain():
            t0 = time.perf_counter()
            if ext is not None:
                acc = accs.get("ext") or accs.setdefault("ext", LevelAccumulator(ext, L, LevelAccumulator.MOMENTS, mean_only=True))
                n, _, s, _ = acc.estimate(chunks, reduce=False)      # rank-local: only rank 0 runs this chain, no collective
                cov = np.sum(linearize.covariance_sums_from_moment_sums(fn, s) / n[:, None], axis=0).reshape(fn.size, fn.size)
                ortho, info = sd.construct_ortogonal_moments(fn, cov, tol=1e-4)
                means = ortho._base_matrix @ np.sum(s[:, :fn.size] / n[:, None], axis=0)
            else:
                acc = accs.get("cov") or accs.setdefault("cov", LevelAccumulator(fn, L, LevelAccumulator.COV, mean_only=True))
                n, _, s, _ = acc.estimate(chunks, reduce=False)
                cov = np.sum(s / n[:, None], axis=0).reshape(fn.size, fn.size)
                ortho, info = sd.construct_ortogonal_moments(fn, cov, tol=1e-4)
                acc2 = LevelAccumulator(ortho, L, LevelAccumulator.MOMENTS, mean_only=True)
                n2, _, s2, _ = acc2.estimate(chunks, reduce=False)
                means = np.sum(s2 / n2[:, None], axis=0)
                acc2.close()
            t1 = time.perf_counter()
            distr = sd.SimpleDistribution(ortho, np.stack([means, np.ones_like(means)], axis=1), domain=fn.domain)
            res = distr.estimate_density_minimize(tol=1e-8)
            t2 = time.perf_counter()
            return 1e3 * (t1 - t0), 1e3 * (t2 - t1), ortho, res

    gc.collect()
    gc.disable()      # a generation-2 pass of the cyclic collector (40-80 ms with torch loaded) would land in one of the timings
    try:
        c1, s1, _, _ = chain()
        c2, s2, ortho, res = chain()
        c3, s3, ortho, res = chain()
    finally:
        gc.enable()
    for a in accs.values():
        a.close()
    if through_api:
        qe.device_cache_clear()
    c2, s2 = min(c2, c3), min(s2, s3)
    if ext is not None:
        how = ("one mean-only moments pass of {} terms; covariance mean by product linearisation, orthogonal-moments means from "
               "the same sums".format(ext.size))
    elif type(fn).__name__ == "Spline":
        how = "banded covariance-mean pass of the spline moments (k_spline_band_accum, no matrix cores) + moments pass over the orthogonal moments"
    else:
        how = "matrix-core covariance pass (mean only) + moments pass over the orthogonal moments"
    return {"solve_ms": round(s2, 3), "first_solve_ms": round(s1, 3), "estimate_chain_ms": round(c2, 3),
            "first_estimate_chain_ms": round(c1, 3), "n_moments_in": fn.size, "n_moments_orthogonal": int(ortho.size),
            "estimate_chain": how, "through": ("the Python API (Estimate.construct_density's statements over a DeviceMemory storage)"
                                                if through_api else "LevelAccumulator, rank-local"),
            "nit": int(res.nit), "grad_norm": float(res.fun_norm), "success": bool(res.success)}


def csrc_sha():
    """sha256 over the kernel sources (mlmc_amd/csrc/*.hip, *.hpp, include/mlmc_hip.h): names the build a profile belongs to."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "mlmc_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "mlmc_amd", "csrc", "*.hpp")))
    for f in files + [os.path.join(ROOT, "include", "mlmc_hip.h")]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _latest_pmc(config_key):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_config{}.json".format(config_key))))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.basename(files[-1])


def pmc_traffic(config_key, kname):
    """HBM bytes per accumulation launch.  Hardware counters cannot be read from inside this process: the figure comes from
    the most recent COMMITTED rocprofv3 PMC passes of this command (profiles/rNN_pmc_config<k>.json: separate --pmc
    FETCH_SIZE / WRITE_SIZE runs, tools/collect_profiles.sh) and is reported as `traffic_from_profile` with the file in
    `traffic_source`.  It is promoted to `traffic` only when the profile was taken from THIS build (the profile records the
    sha of the kernel sources, `csrc_sha`); a profile of older kernels leaves `traffic` null instead of going stale silently.
    FETCH_SIZE is in KB and, on gfx950, counts half of a coalesced streaming read (guide, HBM section) -> read bytes =
    2 * FETCH_SIZE * 1024; per-dispatch averages over the dispatches of the matching kernels."""
    value, name, sha = pmc_avg_bytes_per_dispatch(config_key, kname)
    same = value is not None and sha is not None and sha == csrc_sha()
    return {"traffic": value if same else None, "traffic_from_profile": value, "traffic_source": name,
            "traffic_profile_matches_build": bool(same)}


def pmc_fp64_pipe(config_key, kname, n_simd=1024):
    """Occupation of the shared fp64 pipe by the launches of `kname`, from the SQ counters of the committed PMC pass (separate
    run, tools/collect_profiles.sh): matrix and vector fp64 instructions time-share ONE pipe per SIMD on gfx950
    (tools/ubench_mfma_f64.hip), a v_mfma_f64_16x16x4_f64 holds it 64 cycles (SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA), any other
    vector instruction of a 64-wide wave 4.  Fractions of GRBM_GUI_ACTIVE / 8 (XCDs), weighted over the matching kernels by
    their dispatch counts.  None when the profile holds no SQ pass or was taken from other kernel sources."""
    prof, name = _latest_pmc(config_key)
    if prof is None or (prof.get("_meta") or {}).get("csrc_sha") != csrc_sha():
        return None
    mfma = valu = active = 0.0
    for kn, e in prof.items():
        if kn.startswith("_") or kname not in kn or "SQ_VALU_MFMA_BUSY_CYCLES_avg_per_dispatch" not in e or "[" in kn:
            continue
        d = e.get("dispatches_sq", 0)
        mfma += e["SQ_VALU_MFMA_BUSY_CYCLES_avg_per_dispatch"] / n_simd * d
        valu += 4.0 * (e["SQ_INSTS_VALU_avg_per_dispatch"] - e["SQ_INSTS_MFMA_avg_per_dispatch"]) / n_simd * d
        active += e["GRBM_GUI_ACTIVE_avg_per_dispatch"] / 8.0 * d
    if active <= 0:
        return None
    return {"mfma_frac": round(mfma / active, 4), "other_vector_frac": round(valu / active, 4), "busy_frac": round((mfma + valu) / active, 4),
            "source": name, "note": "SQ counters of the committed profile of this build: cycles the one fp64 pipe of a SIMD is held by matrix "
                                    "instructions (64 each) and by the other vector instructions (4 each), over GRBM_GUI_ACTIVE / 8"}


def pmc_avg_bytes_per_dispatch(config_key, kname):
    prof, name = _latest_pmc(config_key)
    if prof is None:
        return None, None, None
    total = 0.0
    n_disp = 0
    for kn, e in prof.items():
        if kn.startswith("_") or kname not in kn or "FETCH_SIZE_avg_per_dispatch" not in e:
            continue
        d = e.get("dispatches_fetch", 0)
        total += (2.0 * e["FETCH_SIZE_avg_per_dispatch"] + e.get("WRITE_SIZE_avg_per_dispatch", 0.0)) * 1024.0 * d
        n_disp += d
    if n_disp == 0:
        return None, None, None
    return int(total / n_disp), name, (prof.get("_meta") or {}).get("csrc_sha")


def cpu_baseline_and_parity(cfg, fn, dom, steps, onp, LevelAccumulator, level_stats):
    """NumPy restatement of the reference path (oracle, same operation order as the reference: legvander, transpose,
    NaN mask, fine - coarse, two np.sum; covariance: the reference's per-sample einsum outer products), single thread
    like the reference, streamed in chunks; a bounded sample (about 10 s of CPU work)."""
    L, R = cfg["L"], cfg["R"]
    if cfg["mode"] == "moments":
        n_s, chunk = 10_000_000 * 32 // max(R, 32), 250_000   # about 10 s of NumPy on one core
        rows = onp.moments_rows
    else:
        n_s, chunk = 60_000 * (64 * 64) // (R * R), 1_000     # the reference form materialises [2, n, R, R] per chunk
        rows = onp.covariance_rows
    b = onp.Basis(onp.SPLINE if cfg.get("basis") == "Spline" else onp.LEGENDRE, R, dom)
    if cfg.get("basis") == "Spline":
        n_s, chunk = 400_000, 50_000
    # the first n_s samples of every level of the headline workload (rank 0, N = 1): the very bits the GPU step consumed
    host = []
    for l in range(L):
        blocks = list(synth_level_host_blocks(l, 0, n_s, steps))
        host.append((np.concatenate([b[1] for b in blocks]), None if l == 0 else np.concatenate([b[2] for b in blocks])))
    level_chunks = []
    for l, (f, c) in enumerate(host):
        cl = []
        for i in range(0, n_s, chunk):
            x = np.stack([f[i:i + chunk], (c if c is not None else f)[i:i + chunk]], axis=-1)[None]
            cl.append(x[:, :, :1] if l == 0 else x)
        level_chunks.append(cl)
    t0 = time.perf_counter()
    ref = onp.estimate_mean(level_chunks, lambda x: rows(b, x))
    cpu_s = time.perf_counter() - t0
    cpu = {"value": L * n_s * R / cpu_s, "unit": "moment-evals/s", "cores": 1, "kind": "port",
           "sample": "the first {1} samples of each of the {0} levels of the headline workload, Legendre R={2}, {3} estimate, NumPy "
                     "restatement of the reference path (oracle/oracle_np.py), chunks of {4}; {5:.2f} s on 1 of {6} host cores".format(
                         L, n_s, R, cfg["mode"], chunk, cpu_s, os.cpu_count()).replace("Legendre", cfg.get("basis", "Legendre"))}
    mode = LevelAccumulator.MOMENTS if cfg["mode"] == "moments" else LevelAccumulator.COV
    acc = LevelAccumulator(fn, L, mode)
    for l, (f, c) in enumerate(host):
        acc.push(l, f, c)
    n, n_rm, s, sp = acc.finalize(reduce=False)
    l_means, l_vars = level_stats(n, s, sp)
    rms = np.sqrt(np.abs(ref.sums_sq) / np.maximum(ref.n_samples[:, None], 1))
    err_mean = float(np.max(np.abs(l_means - ref.l_means) / np.maximum(np.abs(ref.l_means), rms + 1e-300)))
    err_var = float(np.max(np.abs(l_vars - ref.l_vars) / np.maximum(np.abs(ref.l_vars), 1e-300)))
    parity = {"counts_bit_exact": bool(np.array_equal(n, ref.n_samples) and np.array_equal(n_rm, ref.n_rm_samples)),
              "max_rel_err_l_means": err_mean, "max_rel_err_l_vars": err_var, "tolerance": 1e-10,
              "ok": bool(np.array_equal(n, ref.n_samples) and np.array_equal(n_rm, ref.n_rm_samples)
                         and err_mean <= 1e-10 and err_var <= 1e-10)}
    return cpu, parity


if __name__ == "__main__":
    main()
