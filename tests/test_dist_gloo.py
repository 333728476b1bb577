"""world_size = 2 `gloo` tests (CPU): the N > 1 path = shard each level's samples over ranks + one packed all-reduce
of the per-level partials + the host formulas.  Without a GPU the per-rank partial sums come from the oracle (the
checker standing in for the kernels); everything else -- shard_bounds, allreduce_partials, level_stats,
QuantityMean -- is the product code that runs on the GPU box under RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle_np as onp
from tests.util import level_arrays, to_chunks


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, steps, R, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mlmc_amd.engine import allreduce_partials, level_stats, shard_bounds, unpack_partials
        dom = (-3.7190164854556804, 3.7190164854556804)
        b = onp.Basis(onp.LEGENDRE, R, dom)
        levels = level_arrays(N, steps, 1, 13)
        L = len(N)
        packed = torch.zeros(2 * L + 2 * L * R, dtype=torch.float64)      # n | n_rm | s | sp
        for l, (f, c) in enumerate(levels):
            lo, hi = shard_bounds(f.shape[1], rank, world)
            x = f[:, lo:hi, None] if c is None else np.stack([f[:, lo:hi], c[:, lo:hi]], axis=-1)
            rows = onp.moments_rows(b, x)
            chunk, n_mask = onp.mask_nan_samples(rows)
            d = chunk[:, :, 0] if l == 0 else chunk[:, :, 0] - chunk[:, :, 1]
            packed[l] = chunk.shape[1]
            packed[L + l] = n_mask
            packed[2 * L + l * R:2 * L + (l + 1) * R] = torch.from_numpy(d.sum(axis=1))
            packed[2 * L + (L + l) * R:2 * L + (L + l + 1) * R] = torch.from_numpy((d ** 2).sum(axis=1))
        n, n_rm, s, sp = unpack_partials(allreduce_partials(packed), L, R)
        l_means, l_vars = level_stats(n, s, sp)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), n=n, n_rm=n_rm, l_means=l_means, l_vars=l_vars)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_estimate_matches_single(tmp_path):
    N, steps, R = [5001, 3000, 1777], [0.5, 0.07, 0.01], 9
    port = _free_port()
    mp.spawn(_worker, args=(2, port, N, steps, R, str(tmp_path)), nprocs=2, join=True)
    dom = (-3.7190164854556804, 3.7190164854556804)
    b = onp.Basis(onp.LEGENDRE, R, dom)
    ref = onp.estimate_mean(to_chunks(level_arrays(N, steps, 1, 13)), lambda x: onp.moments_rows(b, x))
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for k in ("n", "n_rm", "l_means", "l_vars"):
        assert np.array_equal(r0[k], r1[k])                      # every rank ends with the same result
    assert np.array_equal(r0["n"], ref.n_samples) and np.array_equal(r0["n_rm"], ref.n_rm_samples)   # counts reduce exactly
    assert np.allclose(r0["l_means"], ref.l_means, rtol=1e-12, atol=1e-14)
    assert np.allclose(r0["l_vars"], ref.l_vars, rtol=1e-10, atol=0)


def _worker_cov(rank, world, port, N, steps, R, M, out_dir):
    """covariance rows (K = M R^2 per level) of a vector quantity through the same packed all-reduce"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mlmc_amd.engine import allreduce_partials, level_stats, moments_from_covariance, shard_bounds, unpack_partials
        dom = (-3.7190164854556804, 3.7190164854556804)
        b = onp.Basis(onp.LEGENDRE, R, dom)
        levels = level_arrays(N, steps, M, 13)
        L, K = len(N), M * R * R
        packed = torch.zeros(2 * L + 2 * L * K, dtype=torch.float64)      # n | n_rm | s | sp: the layout of mlmc_accum_finalize_packed
        for l, (f, c) in enumerate(levels):
            lo, hi = shard_bounds(f.shape[1], rank, world)
            x = f[:, lo:hi, None] if c is None else np.stack([f[:, lo:hi], c[:, lo:hi]], axis=-1)
            chunk, n_mask = onp.mask_nan_samples(onp.covariance_rows(b, x))
            d = chunk[:, :, 0] if l == 0 else chunk[:, :, 0] - chunk[:, :, 1]
            packed[l] = chunk.shape[1]
            packed[L + l] = n_mask
            packed[2 * L + l * K:2 * L + (l + 1) * K] = torch.from_numpy(d.sum(axis=1))
            packed[2 * L + (L + l) * K:2 * L + (L + l + 1) * K] = torch.from_numpy((d ** 2).sum(axis=1))
        n, n_rm, s, sp = unpack_partials(allreduce_partials(packed), L, K)
        l_means, l_vars = level_stats(n, s, sp)
        s_m, sp_m = moments_from_covariance(s, sp, R, n_comp=M)          # row 0 of every component's R x R block
        np.savez(os.path.join(out_dir, f"cov_rank{rank}.npz"), n=n, n_rm=n_rm, l_means=l_means, l_vars=l_vars, s_m=s_m, sp_m=sp_m)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_covariance_of_a_vector_quantity(tmp_path):
    """configs[3]'s exchange in small: K = M R^2 covariance rows per level, M = 2 components, two ranks."""
    N, steps, R, M = [1501, 900, 333], [0.5, 0.07, 0.01], 6, 2
    mp.spawn(_worker_cov, args=(2, _free_port(), N, steps, R, M, str(tmp_path)), nprocs=2, join=True)
    dom = (-3.7190164854556804, 3.7190164854556804)
    b = onp.Basis(onp.LEGENDRE, R, dom)
    chunks = to_chunks(level_arrays(N, steps, M, 13))
    ref = onp.estimate_mean(chunks, lambda x: onp.covariance_rows(b, x))
    refm = onp.estimate_mean(chunks, lambda x: onp.moments_rows(b, x))
    r0, r1 = np.load(tmp_path / "cov_rank0.npz"), np.load(tmp_path / "cov_rank1.npz")
    for k in ("n", "n_rm", "l_means", "l_vars", "s_m", "sp_m"):
        assert np.array_equal(r0[k], r1[k])
    assert np.array_equal(r0["n"], ref.n_samples) and np.array_equal(r0["n_rm"], ref.n_rm_samples)
    assert np.allclose(r0["l_means"], ref.l_means, rtol=1e-12, atol=1e-13)
    assert np.allclose(r0["l_vars"], ref.l_vars, rtol=1e-10, atol=0)
    # the moments' level sums sit in row 0 of each component's covariance block
    assert np.allclose(r0["s_m"], refm.sums, rtol=1e-12, atol=1e-12) and np.allclose(r0["sp_m"], refm.sums_sq, rtol=1e-12, atol=0)


def test_shard_bounds_partition():
    from mlmc_amd.engine import shard_bounds
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
