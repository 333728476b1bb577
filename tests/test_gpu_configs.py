"""GPU tests of BASELINE.json configs[3] and configs[4] at the per-GPU share of an 8-GPU node (run with -m gpu):

  configs[3]  5 levels x 1.25e7 samples per GPU, Legendre n_moments = 64, covariance through mlmc_accum_estimate_packed and
              the ONE packed all-reduce -- with one `nccl` (RCCL) rank and with two `gloo` ranks that split the share --
              equal to the unsharded estimate (counts exact, sums <= 1e-12) and to the C oracle on a prefix;
  configs[4]  1.25e7 samples, cubic B-spline moments R = 128 (not part of the reference: pinned to scipy's BSpline through
              the oracle), mean + variance estimate, then the construct_density chain and the max-entropy solve.

Every rank of these tests lives on the single GPU of the test box (the 8-GPU run itself is the driver's)."""
import os
import socket

import numpy as np
import pytest

from oracle import oracle_c, oracle_np as onp
from tests.util import close

pytestmark = pytest.mark.gpu
DOM = (-3.7190164854556804, 3.7190164854556804)
SHARE = 12_500_000                     # 1e8 samples per level over 8 GPUs


@pytest.fixture(scope="module")
def hip():
    from mlmc_amd import _lib
    _lib.init(0)
    return _lib


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _share_on_device(L, n, seed=4242):
    """The per-GPU share of a sharded synthetic run, generated in HBM; identical in every process that asks for it."""
    import torch
    steps = [s[0] for s in onp.determine_level_parameters(L, [0.5, 0.01])] if L > 1 else [0.01]
    gen = torch.Generator(device="cuda")
    out = []
    for l in range(L):
        gen.manual_seed(seed + l)
        x = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen)
        root = torch.sqrt(1e-4 + x.abs())
        out.append(((x + steps[l] * root).contiguous(), None if l == 0 else (x + steps[l - 1] * root).contiguous()))
    return out


def _cov_worker(rank, world, backend, port, L, n, R, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      MLMC_HIP_FORCE_DIST="1")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    try:
        from mlmc_amd import _lib, Legendre
        from mlmc_amd.engine import LevelAccumulator, shard_bounds
        _lib.init(0)
        data = _share_on_device(L, n)
        chunks = []
        for l, (f, c) in enumerate(data):
            lo, hi = shard_bounds(n, rank, world)
            chunks.append((l, f[lo:hi].contiguous(), None if c is None else c[lo:hi].contiguous()))
        torch.cuda.synchronize()
        acc = LevelAccumulator(Legendre(R, DOM), L, LevelAccumulator.COV)
        n_s, n_rm, s, sp = acc.estimate(chunks)          # mlmc_accum_estimate_packed + ONE all-reduce of [L (2 + 2 R^2)] doubles
        again = acc.estimate(chunks)
        for a, b in zip(again, (n_s, n_rm, s, sp)):
            assert np.array_equal(a, b)                   # the packed path is reproducible run to run
        np.savez(os.path.join(out_dir, f"{backend}{world}_rank{rank}.npz"), n=n_s, n_rm=n_rm, s=s, sp=sp)
    finally:
        dist.destroy_process_group()


def test_config3_share_covariance_through_the_packed_allreduce(hip, tmp_path):
    import torch.multiprocessing as mp
    from mlmc_amd import Legendre
    from mlmc_amd.engine import LevelAccumulator, moments_from_covariance
    L, n, R = 5, SHARE, 64
    mp.spawn(_cov_worker, args=(1, "nccl", _free_port(), L, n, R, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_cov_worker, args=(2, "gloo", _free_port(), L, n, R, str(tmp_path)), nprocs=2, join=True)
    data = _share_on_device(L, n)
    fn = Legendre(R, DOM)
    acc = LevelAccumulator(fn, L, LevelAccumulator.COV)
    n0, n_rm0, s0, sp0 = acc.estimate([(l, f, c) for l, (f, c) in enumerate(data)], reduce=False)     # unsharded, no collective
    assert np.all(n0 + n_rm0 == n)
    scale = np.sqrt(np.abs(sp0) * n0[:, None])
    one = np.load(tmp_path / "nccl1_rank0.npz")
    assert np.array_equal(one["n"], n0) and np.array_equal(one["n_rm"], n_rm0)
    assert np.array_equal(one["s"], s0) and np.array_equal(one["sp"], sp0)       # one rank: the all-reduce adds nothing
    r0, r1 = np.load(tmp_path / "gloo2_rank0.npz"), np.load(tmp_path / "gloo2_rank1.npz")
    for k in ("n", "n_rm", "s", "sp"):
        assert np.array_equal(r0[k], r1[k])                                       # every rank ends with the same sums
    assert np.array_equal(r0["n"], n0) and np.array_equal(r0["n_rm"], n_rm0)     # counts reduce exactly
    assert close(r0["s"], s0, scale, 1e-12) and close(r0["sp"], sp0, None, 1e-12)
    # the moments' level sums read from row 0 of the covariance sums = a moments estimate of the same samples
    accm = LevelAccumulator(fn, L, LevelAccumulator.MOMENTS)
    nm, _, sm, spm = accm.estimate([(l, f, c) for l, (f, c) in enumerate(data)], reduce=False)
    s_row, sp_row = moments_from_covariance(s0, sp0, R)
    assert np.array_equal(nm, n0)
    assert close(s_row, sm, np.sqrt(spm * n0[:, None]), 1e-10) and close(sp_row, spm, None, 1e-10)
    # C oracle (reference form: per-sample outer products) on a prefix
    k = 20000
    accp = LevelAccumulator(fn, L, LevelAccumulator.COV)
    n3, n_rm3, s3, sp3 = accp.estimate([(l, f[:k].contiguous(), None if c is None else c[:k].contiguous())
                                        for l, (f, c) in enumerate(data)], reduce=False)
    b = onp.Basis(onp.LEGENDRE, R, DOM)
    for l in (0, 3):
        f, c = data[l]
        nk, nr, so, spo = oracle_c.cov_level(b, f[:k].cpu().numpy(), None if c is None else c[:k].cpu().numpy())
        assert nk == n3[l] and nr == n_rm3[l]
        assert close(s3[l], so, np.sqrt(spo / nk) * nk, 1e-10) and close(sp3[l], spo, None, 1e-10)


def _spline_worker(rank, world, port, n, R, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        from mlmc_amd import _lib, Spline
        from mlmc_amd.engine import LevelAccumulator, shard_bounds
        _lib.init(0)
        f, _ = _share_on_device(1, n)[0]
        lo, hi = shard_bounds(n, rank, world)
        shard = f[lo:hi].contiguous()
        torch.cuda.synchronize()
        acc = LevelAccumulator(Spline(R, DOM), 1, LevelAccumulator.MOMENTS)
        n_s, n_rm, s, sp = acc.estimate([(0, shard, None)])
        np.savez(os.path.join(out_dir, f"spline_rank{rank}.npz"), n=n_s, n_rm=n_rm, s=s, sp=sp)
    finally:
        dist.destroy_process_group()


def test_config4_share_spline_moments_and_maxent(hip, tmp_path):
    import torch.multiprocessing as mp
    from mlmc_amd import Spline
    from mlmc_amd.engine import LevelAccumulator, level_stats
    from mlmc_amd.tool import simple_distribution as sd
    n, R = SHARE, 128
    mp.spawn(_spline_worker, args=(2, _free_port(), n, R, str(tmp_path)), nprocs=2, join=True)
    f, _ = _share_on_device(1, n)[0]
    fn = Spline(R, DOM)
    acc = LevelAccumulator(fn, 1, LevelAccumulator.MOMENTS)
    n0, n_rm0, s0, sp0 = acc.estimate([(0, f, None)], reduce=False)
    assert n0[0] + n_rm0[0] == n and s0[0, 0] == float(n0[0]) and sp0[0, 0] == float(n0[0])      # phi_0 = 1: exact counts
    # partition of unity: B_0 = 1 - sum of the others lies in [0, 1] for every kept sample
    assert 0.0 <= n0[0] - s0[0, 1:].sum() <= n0[0] * 1e-2
    assert np.all(sp0 <= s0 * (1 + 1e-12)) and np.all(s0 >= 0)                                   # 0 <= B_r <= 1
    r0, r1 = np.load(tmp_path / "spline_rank0.npz"), np.load(tmp_path / "spline_rank1.npz")
    for k in ("n", "n_rm", "s", "sp"):
        assert np.array_equal(r0[k], r1[k])
    assert np.array_equal(r0["n"], n0) and np.array_equal(r0["n_rm"], n_rm0)
    assert close(r0["s"], s0, None, 1e-12) and close(r0["sp"], sp0, None, 1e-12)
    # oracle (scipy BSpline) on a prefix
    k = 200_000
    accp = LevelAccumulator(fn, 1, LevelAccumulator.MOMENTS)
    n3, n_rm3, s3, sp3 = accp.estimate([(0, f[:k].contiguous(), None)], reduce=False)
    b = onp.Basis(onp.SPLINE, R, DOM)
    ref = onp.estimate_mean([[f[:k].cpu().numpy()[None, :, None]]], lambda x: onp.moments_rows(b, x))
    assert np.array_equal(n3, ref.n_samples) and np.array_equal(n_rm3, ref.n_rm_samples)
    l_means, l_vars = level_stats(n3, s3, sp3)
    assert close(l_means, ref.l_means, 1e-3, 1e-10) and close(l_vars, ref.l_vars, None, 1e-10)
    # the construct_density chain on the whole share (estimator.py:304-331): covariance -> orthogonal moments -> moments
    # in the orthogonal basis -> max-entropy solve; the reconstructed density reproduces the estimated moments
    accc = LevelAccumulator(fn, 1, LevelAccumulator.COV, mean_only=True)
    nc, _, sc, _ = accc.estimate([(0, f, None)], reduce=False)
    cov = (sc[0] / nc[0]).reshape(R, R)
    assert np.array_equal(cov, cov.T) and cov[0, 0] == 1.0
    assert close(cov[0], s0[0] / n0[0], 1e-3, 1e-10)                                            # row 0 = the moment means
    ortho, info = sd.construct_ortogonal_moments(fn, cov, tol=1e-4)
    acc2 = LevelAccumulator(ortho, 1, LevelAccumulator.MOMENTS, mean_only=True)
    n2, _, s2, _ = acc2.estimate([(0, f, None)], reduce=False)
    means = s2[0] / n2[0]
    distr = sd.SimpleDistribution(ortho, np.stack([means, np.ones_like(means)], axis=1), domain=fn.domain)
    res = distr.estimate_density_minimize(tol=1e-8)
    assert res.success and res.fun_norm < 1e-6
    # moments of the reconstructed density by Gauss-Legendre on every knot span (the integrands are piecewise smooth there)
    gx, gw = np.polynomial.legendre.leggauss(12)
    edges = np.linspace(DOM[0], DOM[1], R - 3 + 1)
    mid, half = (edges[1:] + edges[:-1]) / 2, (edges[1:] - edges[:-1]) / 2
    xq = (mid[:, None] + half[:, None] * gx[None, :]).ravel()
    wq = (half[:, None] * gw[None, :]).ravel()
    got = (ortho.eval_all(xq) * (distr.density(xq) * wq)[:, None]).sum(axis=0)
    assert np.max(np.abs(got - means)) < 1e-5, np.max(np.abs(got - means))
    x = np.linspace(DOM[0], DOM[1], 4001)
    dens = distr.density(x)
    # the samples are N(0, 1) (+ a 1 % step perturbation) clipped to the domain: the density follows the normal pdf
    core = np.abs(x) < 2.5
    assert np.max(np.abs(dens[core] - np.exp(-x[core] ** 2 / 2) / np.sqrt(2 * np.pi))) < 2e-2


def test_bench_multi_rank_default_is_config3_sharded(hip):
    """`bench.py --gpus N` (N > 1) as the driver launches it -- torch.distributed.run, one rank per GPU -- rehearsed with two
    gloo ranks on the one GPU of the box and a reduced total (MLMC_BENCH_TOTAL_PER_LEVEL): BASELINE configs[3] sharded over
    the ranks, covariance R = 64 through the packed all-reduce, strong scaling, exchange block in the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    total = 3_000_001
    env = dict(os.environ, MLMC_BENCH_BACKEND="gloo", MLMC_BENCH_DEVICE="0", MLMC_BENCH_TOTAL_PER_LEVEL=str(total))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 2 and d["warmup"] == 1
    assert "configs[3]" in d["config"]["workload"] and d["config"]["estimate"] == "cov" and d["config"]["n_moments"] == 64
    assert d["config"]["samples_per_level_total"] == total and d["config"]["samples_per_level_per_gpu"] in (total // 2, total - total // 2)
    assert d["exchange"]["bytes_per_rank"] == 8 * (2 * 5 + 2 * 5 * 64 * 64) and d["exchange"]["allreduce_ms"] > 0
    # top level: executed matrix-core flops (symmetric tiles once) -- a physical fraction; the reference-form count rides beside it
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1.0
    assert d["roofline"]["frac"] < d["roofline"]["reference_form"]["frac"]
    n_loc = d["config"]["samples_per_level_per_gpu"]
    # variance Grams only at the pair levels (16 + 10 tiles per pair), no matrix pass at level 0; the mean's Gram matrix and the
    # level-0 second moments are replaced by the aux passes (127 terms over the pair levels, 253 in two windows at level 0)
    assert d["roofline"]["executed_mfma_flops_per_step"] == 512 * (26 * 4) * n_loc
    aux = d["roofline"]["aux_kernel"]
    assert aux["launches_per_step"] == 3 and aux["ms_per_step"] > 0 and "127" in aux["kernel"] and "253" in aux["kernel"]
    assert d["roofline"]["reference_form"]["alg_flops_reference_form"] == (6 * 64 * 64 + 14 * 64) * 4 * n_loc + (4 * 64 * 64 + 8 * 64) * n_loc
    assert "traffic_from_profile" in d["roofline"] and (d["roofline"]["traffic"] is None or d["roofline"]["traffic_profile_matches_build"])
    rc = d["result_check"]
    assert rc["mean0"] == 1.0 and rc["var0"] == 0.0 and len(rc["n_estimated"]) == 5
    assert all(0 < r < 0.01 * total for r in rc["n_removed"])                          # both shards were counted
    assert abs(d["value"] - 5 * total * 64 / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]   # whole-job samples x R / time


def test_bench_default_line_carries_the_contract(hip):
    """`python bench.py` (N = 1 defaults apart from K / W): ONE JSON line with the driver's keys, BASELINE configs[2] as the
    workload, `roofline` (bound / achieved / peak / unit / frac / traffic) of the dominant kernel with the auxiliary moments
    launch and the three-Gram form beside it, `cpu_baseline` on a bounded sample, the in-run parity gate green."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2"], capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert "configs[2]" in d["config"]["workload"] and d["config"]["n_moments"] == 64 and d["config"]["samples_per_level_per_gpu"] == 10_000_000
    assert abs(d["value"] - 5 * 10_000_000 * 64 / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.5 < r["frac"] < 1.0 and "traffic" in r
    assert r["executed_mfma_flops_per_step"] == 512 * (26 * 4) * 10_000_000
    assert r["aux_kernel"]["launches_per_step"] == 3 and 0 < r["aux_kernel"]["ms_per_step"] < 0.3 * d["ms_per_step"]
    if r["traffic"] is not None:                                        # a profile of this very build is committed
        assert r["traffic_profile_matches_build"] and 0.5 < r["fp64_pipe"]["busy_frac"] <= 1.0
    t3 = d["three_gram_form"]
    assert t3["roofline"]["executed_mfma_flops_per_step"] == 512 * (42 * 4 + 20) * 10_000_000
    assert t3["ms_per_step"] > d["ms_per_step"] and t3["roofline"]["frac"] > r["frac"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == d["unit"] and cb["value"] > 0 and "sample" in cb
    assert d["parity"]["ok"] and d["parity"]["counts_bit_exact"]
    assert d["result_check"]["mean0"] == 1.0 and d["result_check"]["var0"] == 0.0
    for k in ("moments_r64", "configs1", "north_star", "pdf_solve", "h2d_inclusive"):
        assert k in d, k
