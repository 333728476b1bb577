"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the golden
vectors generated from the imported reference (tests/golden) and against the oracle on seeded inputs.

Tolerances: counts / masks / indices bit-exact; fp64 values |a - b| <= 1e-10 * max(|b|, rms of the level)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_c, oracle_np as onp
from tests.util import close, level_arrays, to_chunks

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-10


@pytest.fixture(scope="module")
def hip():
    from mlmc_amd import _lib
    _lib.init(0)
    return _lib


@pytest.fixture(scope="module")
def g1():
    return np.load(os.path.join(GOLDEN, "G1_basis.npz"))


@pytest.fixture(scope="module")
def g2():
    return np.load(os.path.join(GOLDEN, "G2_estimate_mean.npz"))


def _same_nan(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b))


def _vals_close(a, b, tol=TOL):
    m = ~np.isnan(b)
    # basis values are O(1) (Legendre, Fourier) or grow like t^k (monomial, safe_eval False): relative to max(1, |b|)
    return np.all(np.abs(a[m] - b[m]) <= tol * np.maximum(1.0, np.abs(b[m])))


def test_basis_eval_golden(hip, g1):
    from mlmc_amd import Legendre, Monomial, Fourier
    dom = tuple(g1["dom"])
    for R in (1, 2, 5, 10, 32, 64):
        for safe in (True, False):
            g = g1["grid"] if safe else g1["grid_nosafe"]
            for cls, name in ((Legendre, "legendre"), (Monomial, "monomial")):
                got = cls(R, dom, safe_eval=safe).eval_all(g)
                ref = g1[f"{name}_R{R}_safe{int(safe)}"]
                assert got.shape == ref.shape
                assert _same_nan(got, ref), (name, R, safe)
                if name == "monomial" and not safe:
                    m = np.isfinite(ref) & (np.abs(ref) < 1e200)
                    assert np.all(np.abs(got[m] - ref[m]) <= 1e-10 * np.maximum(1.0, np.abs(ref[m])))
                else:
                    assert _vals_close(got, ref), (name, R, safe)
    for R in (1, 2, 5, 6, 33):
        got = Fourier(R, dom).eval_all(g1["grid"])
        ref = g1[f"fourier_R{R}_safe1"]
        # reference quirk, kept: Fourier column 0 is the constant 1 even in the rows of masked (NaN) inputs (moments.py:156)
        assert _same_nan(got, ref) and _vals_close(got, ref)
        assert np.all(got[:, 0] == 1.0)
    assert np.array_equal(Legendre(4, (-1.0, 1.0))(np.array([0.0, 0.25, 0.5, 0.75, 1.0])), g1["kat_legendre"]) or \
        _vals_close(Legendre(4, (-1.0, 1.0))(np.array([0.0, 0.25, 0.5, 0.75, 1.0])), g1["kat_legendre"], 1e-15)


def test_basis_eval_log_ref_nd_transformed(hip, g1):
    from mlmc_amd import Legendre, Monomial, TransformedMoments
    ldom, dom = tuple(g1["ldom"]), tuple(g1["dom"])
    for R in (5, 32):
        for cls, name in ((Legendre, "legendre"), (Monomial, "monomial")):
            got = cls(R, ldom, log=True).eval_all(g1["lgrid"])
            ref = g1[f"{name}_log_R{R}"]
            assert _same_nan(got, ref) and _vals_close(got, ref)
    got = Legendre(7, dom, ref_domain=(-0.5, 0.75)).eval_all(g1["grid"])
    assert _same_nan(got, g1["legendre_ref_R7"]) and _vals_close(got, g1["legendre_ref_R7"])
    got = Monomial(7, dom, ref_domain=(-1.0, 2.0)).eval_all(g1["grid"])
    assert _same_nan(got, g1["monomial_ref_R7"]) and _vals_close(got, g1["monomial_ref_R7"])
    got = Legendre(9, dom).eval_all(g1["x3"])
    assert got.shape == (3, 17, 2, 9) and _same_nan(got, g1["legendre_x3_R9"]) and _vals_close(got, g1["legendre_x3_R9"])
    tm = TransformedMoments(Legendre(9, dom), g1["tm_matrix"])
    got = tm.eval_all(g1["x3"])
    ref = g1["transformed_x3"]
    assert _same_nan(got, ref) and np.all(np.abs(got - ref)[~np.isnan(ref)] <= 1e-10 * 10)
    got = tm.eval_all(g1["grid"], 4)
    ref = g1["transformed_grid_size4"]
    assert got.shape == ref.shape and _same_nan(got, ref)
    fn = Legendre(6, dom)
    for key, got in (("legendre_single3", fn.eval_single_moment(3, g1["grid"])), ("legendre_eval3", fn.eval(3, g1["grid"])),
                     ("legendre_diff", fn.eval_diff(g1["grid"])), ("legendre_diff2", fn.eval_diff2(g1["grid"])),
                     ("legendre_der1", fn.eval_all_der(g1["grid"], degree=1)),
                     ("monomial_eval3", Monomial(6, dom).eval(3, g1["grid"]))):
        ref = g1[key]
        assert got.shape == ref.shape and _same_nan(got, ref), key
        m = ~np.isnan(ref)
        assert np.all(np.abs(got[m] - ref[m]) <= 1e-10 * np.maximum(1.0, np.abs(ref[m]))), key


def _run_accum(fn, levels, mode=None, n_comp=1):
    from mlmc_amd.engine import LevelAccumulator
    acc = LevelAccumulator(fn, len(levels), LevelAccumulator.MOMENTS if mode is None else mode, n_comp=n_comp)
    for l, (f, c) in enumerate(levels):
        acc.push(l, f if n_comp > 1 else np.ravel(f), c if (c is None or n_comp > 1) else np.ravel(c))
    return acc.finalize()


def _check_against(n, n_rm, s, sp, ref):
    from mlmc_amd.engine import level_stats
    assert np.array_equal(n, ref.n_samples), (n, ref.n_samples)
    assert np.array_equal(n_rm, ref.n_rm_samples), (n_rm, ref.n_rm_samples)
    l_means, l_vars = level_stats(n, s, sp)
    rms = np.sqrt(np.abs(ref.sums_sq) / np.maximum(ref.n_samples[:, None], 1))
    assert close(l_means, ref.l_means, rms, TOL)
    assert close(l_vars, ref.l_vars, None, TOL)
    mean = np.sum(l_means, axis=0)
    with np.errstate(all="ignore"):
        var = np.sum(l_vars / n[:, None], axis=0)
    return mean, var


@pytest.mark.parametrize("tag", ["L3", "L5", "L3nan", "L1"])
def test_estimate_moments_golden(hip, g2, tag):
    """estimate_mean(moments(q, Legendre)) against outputs of the imported reference (G2)."""
    from mlmc_amd import Legendre, Monomial
    dom = tuple(g2["domain"])
    N, steps, nan_every = g2[f"{tag}_N"], g2[f"{tag}_steps"], int(g2[f"{tag}_nan_every"])
    levels = level_arrays(N, steps, 1, nan_every)
    for R in (5, 10, 32, 64):
        n, n_rm, s, sp = _run_accum(Legendre(R, dom), levels)
        key = f"{tag}_leg{R}_b1"
        assert np.array_equal(n, g2[key + "_n"]) and np.array_equal(n_rm, g2[key + "_n_rm"])
        b = onp.Basis(onp.LEGENDRE, R, dom)
        ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.moments_rows(b, x))
        mean, var = _check_against(n, n_rm, s, sp, ref)
        rms = np.sqrt(np.sum(np.abs(ref.sums_sq) / np.maximum(ref.n_samples[:, None], 1), axis=0))
        assert close(mean, g2[key + "_mean"], rms, TOL)
        assert close(var, g2[key + "_var"], None, TOL)
        assert mean[0] == 1.0 and var[0] == 0.0          # reference: test/test_run.py:106-107
    n, n_rm, s, sp = _run_accum(Monomial(6, dom), levels)
    b = onp.Basis(onp.MONOMIAL, 6, dom)
    ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.moments_rows(b, x))
    mean, var = _check_against(n, n_rm, s, sp, ref)
    assert close(mean, g2[f"{tag}_mono6_mean"], 1.0, TOL) and close(var, g2[f"{tag}_mono6_var"], None, TOL)
    # plain quantity mean (no moments node): IDENTITY kind
    n, n_rm, s, sp = _run_accum(None, levels)
    ref = onp.estimate_mean(to_chunks(levels))
    mean, var = _check_against(n, n_rm, s, sp, ref)
    assert np.array_equal(n_rm, g2[f"{tag}_plain_n_rm"])
    assert close(mean, g2[f"{tag}_plain_mean"].ravel(), 1.0, TOL) and close(var, g2[f"{tag}_plain_var"].ravel(), None, TOL)


def test_estimate_moments_vector_quantity(hip, g2):
    """M = 4 components: a sample is dropped when any component of fine or coarse is masked."""
    from mlmc_amd import Legendre
    tag = "L3M4"
    dom = tuple(g2["domain"])
    levels = level_arrays(g2[f"{tag}_N"], g2[f"{tag}_steps"], 4, int(g2[f"{tag}_nan_every"]))
    n, n_rm, s, sp = _run_accum(Legendre(5, dom), levels, n_comp=4)
    key = f"{tag}_leg5_b1"
    assert np.array_equal(n, g2[key + "_n"]) and np.array_equal(n_rm, g2[key + "_n_rm"])
    b = onp.Basis(onp.LEGENDRE, 5, dom)
    ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.moments_rows(b, x))
    mean, var = _check_against(n, n_rm, s, sp, ref)
    assert close(mean, g2[key + "_mean"].ravel(), 1.0, TOL) and close(var, g2[key + "_var"].ravel(), None, TOL)


@pytest.mark.parametrize("lin", ["default", "forced"])
@pytest.mark.parametrize("tag", ["L3", "L3nan", "L1"])
def test_estimate_covariance_golden(hip, g2, tag, lin, monkeypatch):
    """Covariance estimates against the REFERENCE's own outputs (G3, generated by importing the reference).  `forced`: every
    chunk takes the library's linearised route (mean from 2 R - 1 moments, level 0 from 4 R - 3; R = 24 and 64 here) -- at the
    fixtures' sizes the default keeps all three Gram matrices on the matrix cores."""
    from mlmc_amd import Legendre
    from mlmc_amd.engine import LevelAccumulator
    if lin == "forced":
        monkeypatch.setenv("MLMC_HIP_LINEARIZE_MIN_N", "0")
    g3 = np.load(os.path.join(GOLDEN, "G3_cov.npz"))
    dom = tuple(g2["domain"])
    N, steps, nan_every = g2[f"{tag}_N"], g2[f"{tag}_steps"], int(g2[f"{tag}_nan_every"])
    levels = level_arrays(N, steps, 1, nan_every)
    for R in (8, 16, 24, 64):
        key = f"{tag}_cov{R}"
        Ncov = g3[key + "_Ncov"]
        lv = [(f[:, :k], None if c is None else c[:, :k]) for (f, c), k in zip(levels, Ncov)]
        n, n_rm, s, sp = _run_accum(Legendre(R, dom), lv, mode=LevelAccumulator.COV)
        assert np.array_equal(n, g3[key + "_n"]) and np.array_equal(n_rm, g3[key + "_n_rm"])
        b = onp.Basis(onp.LEGENDRE, R, dom)
        ref = onp.estimate_mean(to_chunks(lv), lambda x: onp.covariance_rows(b, x))
        mean, var = _check_against(n, n_rm, s, sp, ref)
        rms = np.sqrt(np.sum(np.abs(ref.sums_sq) / np.maximum(ref.n_samples[:, None], 1), axis=0))
        assert close(mean.reshape(R, R), g3[key + "_mean"], rms.reshape(R, R), TOL)
        assert close(var.reshape(R, R), g3[key + "_var"], None, TOL)
        cov = mean.reshape(R, R)
        assert np.array_equal(cov, cov.T)                       # exactly symmetric by construction
        n2, _, s2, sp2 = _run_accum(Legendre(R, dom), lv)      # cov[:, 0] == moment means (test_quantity_concept.py:613)
        from mlmc_amd.engine import level_stats
        mom_mean = np.sum(level_stats(n2, s2, sp2)[0], axis=0)
        assert close(cov[:, 0], mom_mean, 1.0, 1e-12)


def test_reference_golden_chain(hip):
    """The reference's own golden vector (test/test_sampling_pools.py:18,85-87)."""
    from mlmc_amd import Legendre
    from mlmc_amd.engine import level_stats
    with open(os.path.join(GOLDEN, "G7_chain.json")) as f:
        g7 = json.load(f)
    levels = []
    for l, lev in enumerate(g7["levels"]):
        fine = np.array(lev["fine"])[:, 0]
        coarse = np.array(lev["coarse"])[:, 0]
        levels.append((fine, None if l == 0 else coarse))
    n, n_rm, s, sp = _run_accum(Legendre(5, tuple(g7["domain"])), levels)
    l_means, l_vars = level_stats(n, s, sp)
    means = np.sum(l_means, axis=0)
    vars_ = np.sum(l_vars / n[:, None], axis=0)
    assert means[0] == 1 and vars_[0] == 0
    assert np.allclose(g7["ref_means_test_sampling_pools_py_18"], means, atol=1e-5)
    assert close(means, g7["means"], 1.0, TOL) and close(vars_, g7["vars"], None, 1e-9)


@pytest.mark.parametrize("R", [1, 3, 17, 48, 49, 56, 64, 65, 100, 130])
def test_moment_counts_and_multi_pass(hip, R):
    """ragged sizes, every register-tile instantiation, and R > 64 (several passes over the terms)"""
    from mlmc_amd import Legendre
    dom = (-3.7, 3.7)
    N = [7001, 4099, 513, 1]
    steps = [0.5, 0.1, 0.03, 0.01]
    levels = level_arrays(N, steps, 1, 5)
    n, n_rm, s, sp = _run_accum(Legendre(R, dom), levels)
    b = onp.Basis(onp.LEGENDRE, R, dom)
    ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.moments_rows(b, x))
    _check_against(n, n_rm, s, sp, ref)
    assert s[0, 0] == float(n[0])                     # P0 sums are exact counts


def test_largest_accepted_legendre_size(hip):
    """512 Legendre moments, the largest size mlmc_basis_create accepts: the device recurrence runs on q_i = 2^i Q_i (the
    monic Q_i themselves shrink like 2^-i, their squares would be subnormal from i ~ 480), values and level sums of the top
    moments against legvander (oracle)."""
    from mlmc_amd import Legendre
    dom = (-3.7, 3.7)
    R = 512
    fn = Legendre(R, dom)
    grid = np.linspace(dom[0], dom[1], 1501)
    got = fn.eval_all(grid)
    ref = onp.eval_all(onp.Basis(onp.LEGENDRE, R, dom), grid)
    assert got.shape == ref.shape and np.all(np.abs(got - ref) <= 1e-11)
    assert np.max(np.abs(got[:, -16:])) > 0.01                                    # the top columns are not flushed to zero
    levels = level_arrays([3001, 1500], [0.3, 0.02], 1, 7)
    n, n_rm, s, sp = _run_accum(fn, levels)
    b = onp.Basis(onp.LEGENDRE, R, dom)
    ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.moments_rows(b, x))
    assert np.array_equal(n, ref.n_samples) and np.array_equal(n_rm, ref.n_rm_samples)
    scale = np.sqrt(np.abs(ref.sums_sq) * ref.n_samples[:, None])
    assert close(s, ref.sums, scale, 1e-10) and close(sp, ref.sums_sq, None, 1e-10)
    assert np.all(sp[:, -32:] > 1e-3)                                              # sums of squares of the top moments survive


def test_term_split_kernel_for_49_to_64_moments(hip):
    """48 < R <= 64 of a polynomial family runs k_moments_accum_split (waves 0-1: terms 0..31, waves 2-3: terms 32..63 of the
    same samples, recurrence state handed over through LDS): ragged and tiny sizes (fewer samples than a workgroup has
    lanes, one trip, many trips), level 0 only, host and device inputs, monomials, a vector quantity (mask kernel: the
    general loop), log transform -- against the oracle, and against the two-pass form of the same library (child process
    with MLMC_HIP_NO_SPLIT=1) to 1e-13."""
    import subprocess
    import sys
    import torch
    from mlmc_amd import Legendre, Monomial
    from mlmc_amd.engine import LevelAccumulator
    dom = (-3.7, 3.7)
    steps = [0.5, 0.1, 0.03, 0.01]
    for N in ([1], [127, 1], [128, 129, 255], [257, 1000, 5], [70001, 40099, 513, 2]):
        levels = level_arrays(N, steps[:len(N)], 1, 5 if max(N) > 50 else 0)
        for R in (49, 60, 64):
            b = onp.Basis(onp.LEGENDRE, R, dom)
            ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.moments_rows(b, x))
            n, n_rm, s, sp = _run_accum(Legendre(R, dom), levels)
            _check_against(n, n_rm, s, sp, ref)
            assert s[0, 0] == float(n[0]) and not s[1:, 0].any()
            # device-resident chunks: all levels in ONE launch
            acc = LevelAccumulator(Legendre(R, dom), len(N))
            chunks = [(l, torch.from_numpy(f[0].copy()).cuda(), None if c is None else torch.from_numpy(c[0].copy()).cuda())
                      for l, (f, c) in enumerate(levels)]
            got = acc.estimate(chunks, reduce=False)
            assert np.array_equal(got[0], n) and np.array_equal(got[1], n_rm)
            scale = np.sqrt(np.abs(sp) * np.maximum(n[:, None], 1))
            assert close(got[2], s, scale, 1e-13) and close(got[3], sp, None, 1e-13)     # another grid split: other summation order
            again = acc.estimate(chunks, reduce=False)
            assert all(np.array_equal(a, bb) for a, bb in zip(again, got))               # run to run: bitwise
    levels = level_arrays([9001, 5000, 1300], steps[:3], 1, 7)
    bm = onp.Basis(onp.MONOMIAL, 52, (-3.7, 3.7))
    n, n_rm, s, sp = _run_accum(Monomial(52, (-3.7, 3.7)), levels)
    _check_against(n, n_rm, s, sp, onp.estimate_mean(to_chunks(levels), lambda x: onp.moments_rows(bm, x)))
    # vector quantity (M = 3): keep flags come from the mask kernel
    lv = level_arrays([6001, 3000, 1001], steps[:3], 3, 11)
    b = onp.Basis(onp.LEGENDRE, 64, dom)
    n, n_rm, s, sp = _run_accum(Legendre(64, dom), lv, n_comp=3)
    _check_against(n, n_rm, s, sp, onp.estimate_mean(to_chunks(lv), lambda x: onp.moments_rows(b, x)))
    # log transform (general loop)
    rng = np.random.default_rng(5)
    x = rng.lognormal(mean=0.3, sigma=0.8, size=30011)
    f1, c1 = x * (1 + 0.01 * rng.normal(size=x.size)), x * (1 + 0.03 * rng.normal(size=x.size))
    f1[::501] = -1.0
    ll = [(x[None], None), (f1[None], c1[None])]
    bl = onp.Basis(onp.LEGENDRE, 50, (0.05, 30.0), log=True)
    n, n_rm, s, sp = _run_accum(Legendre(50, (0.05, 30.0), log=True), ll)
    _check_against(n, n_rm, s, sp, onp.estimate_mean(to_chunks(ll), lambda v: onp.moments_rows(bl, v)))
    # the two-pass form of the same build
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from mlmc_amd import _lib, Legendre\nfrom mlmc_amd.engine import LevelAccumulator\nfrom tests.util import level_arrays\n"
            "_lib.init(0)\nlv = level_arrays([70001, 40099, 513, 2], [0.5, 0.1, 0.03, 0.01], 1, 5)\n"
            "acc = LevelAccumulator(Legendre(64, (-3.7, 3.7)), 4)\n"
            "[acc.push(l, f[0], None if c is None else c[0]) for l, (f, c) in enumerate(lv)]\n"
            "n, n_rm, s, sp = acc.finalize()\nnp.savez(sys.argv[1], n=n, n_rm=n_rm, s=s, sp=sp)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "two_pass.npz")
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, MLMC_HIP_NO_SPLIT="1"), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        two = np.load(out)
    lv = level_arrays([70001, 40099, 513, 2], steps, 1, 5)
    n, n_rm, s, sp = _run_accum(Legendre(64, dom), lv)
    assert np.array_equal(two["n"], n) and np.array_equal(two["n_rm"], n_rm)
    scale = np.sqrt(np.abs(sp) * np.maximum(n[:, None], 1))
    assert close(s, two["s"], scale, 1e-13) and close(sp, two["sp"], None, 1e-13)


def test_fourier_and_transformed_accumulate(hip):
    from mlmc_amd import Fourier, Legendre, TransformedMoments
    dom = (-3.7, 3.7)
    levels = level_arrays([5000, 3000, 1001], [0.5, 0.07, 0.01], 1, 9)
    # Fourier: the reference cannot run this inside qe.moments (np.outer); the oracle evaluates the same
    # columns sample by sample
    R = 9
    n, n_rm, s, sp = _run_accum(Fourier(R, dom), levels)
    b = onp.Basis(onp.FOURIER, R, dom)

    def rows(x):
        flat = onp.eval_all(b, x.reshape(-1)).reshape(x.shape + (R,))
        flat[np.isnan(flat).any(axis=-1)] = np.nan
        return flat.transpose((0, 3, 1, 2)).reshape(R, x.shape[1], x.shape[2])
    ref = onp.estimate_mean(to_chunks(levels), rows)
    _check_against(n, n_rm, s, sp, ref)
    # TransformedMoments: variance through the accumulated second moments T G T^T
    rng = np.random.default_rng(3)
    for R0, R1 in ((9, 6), (33, 33), (64, 40), (80, 70), (40, 100), (150, 130)):      # more than 64 rows: 64 x 64 output blocks
        mat = rng.normal(size=(R1, R0)) / np.sqrt(R0)
        mat[0] = 0
        mat[0, 0] = 1
        n, n_rm, s, sp = _run_accum(TransformedMoments(Legendre(R0, dom), mat), levels)
        bt = onp.Basis(onp.LEGENDRE, R0, dom, matrix=mat)
        ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.moments_rows(bt, x))
        _check_against(n, n_rm, s, sp, ref)


def test_edge_cases(hip):
    from mlmc_amd import Legendre, _lib
    from mlmc_amd.engine import LevelAccumulator
    dom = (-1.0, 1.0)
    fn = Legendre(7, dom)
    acc = LevelAccumulator(fn, 2)
    acc.push(0, np.zeros(0))                                   # empty chunk
    acc.push(1, np.zeros(0), np.zeros(0))
    n, n_rm, s, sp = acc.finalize()
    assert n.tolist() == [0, 0] and n_rm.tolist() == [0, 0] and not s.any() and not sp.any()
    # everything masked
    acc.reset()
    acc.push(0, np.full(1000, 5.0))
    acc.push(1, np.full(77, np.nan), np.zeros(77))
    n, n_rm, s, sp = acc.finalize()
    assert n.tolist() == [0, 0] and n_rm.tolist() == [1000, 77] and not s.any()
    # boundary values: end points kept, one ulp outside dropped, -0.0, tiny
    x = np.array([-1.0, 1.0, np.nextafter(1.0, 2.0), np.nextafter(-1.0, -2.0), -0.0, 1e-300, np.inf, -np.inf, np.nan])
    acc.reset()
    acc.push(0, x)
    n, n_rm, s, sp = acc.finalize()
    keep = ~np.isnan(onp.transform(onp.Basis(onp.LEGENDRE, 7, dom), x))     # NumPy rounding decides (1 + ulp maps to exactly 1.0)
    assert n[0] == int(keep.sum()) == 5 and n_rm[0] == 4
    # chunked pushes == one push (additivity), finalize idempotent
    rng = np.random.default_rng(0)
    f = rng.uniform(-1.1, 1.1, 30001)
    c = f + 0.01 * rng.normal(size=f.size)
    acc.reset()
    acc.push(1, f, c)
    r1 = acc.finalize()
    r1b = acc.finalize()
    acc.reset()
    for lo, hi in ((0, 1), (1, 4097), (4097, 30001)):
        acc.push(1, f[lo:hi], c[lo:hi])
    r2 = acc.finalize()
    for a, b_, c_ in zip(r1, r1b, r2):
        assert np.array_equal(a, b_)
    assert np.array_equal(r1[0], r2[0]) and np.array_equal(r1[1], r2[1])
    assert close(r1[2], r2[2], 1.0, 1e-12) and close(r1[3], r2[3], 1.0, 1e-12)
    # fine == coarse -> all level differences are exactly zero
    acc.reset()
    acc.push(1, f, f.copy())
    n, n_rm, s, sp = acc.finalize()
    assert not s[1].any() and not sp[1].any()
    # argument errors come back as exceptions
    with pytest.raises(_lib.MlmcHipError):
        acc.push(5, f)
    with pytest.raises(ValueError):
        acc.push(1, f, c[:10])


def test_device_resident_inputs_and_large_property(hip):
    """BASELINE-size run (3 x 1e7, R = 32) from HBM-resident tensors: size-independent properties +
    C-oracle parity on a 2e6 prefix."""
    import torch
    from mlmc_amd import Legendre
    from mlmc_amd.engine import LevelAccumulator, level_stats
    dom = (-3.7190164854556804, 3.7190164854556804)
    L, n_l, R = 3, 10_000_000, 32
    steps = [s[0] for s in onp.determine_level_parameters(L, [0.5, 0.01])]
    fn = Legendre(R, dom)
    acc = LevelAccumulator(fn, L)
    host = [onp.synth_level_samples(l, n_l, steps, seed=99) for l in range(L)]
    dev = [(torch.from_numpy(f).cuda(), None if c is None else torch.from_numpy(c).cuda()) for f, c in host]
    for l in range(L):
        acc.push(l, *dev[l])
    n, n_rm, s, sp = acc.finalize()
    assert np.all(n + n_rm == n_l)
    assert s[0, 0] == float(n[0]) and sp[0, 0] == float(n[0])            # P0: exact counts
    assert not s[1:, 0].any() and not sp[1:, 0].any()                    # P0 differences: exactly 0
    # counts == number of samples whose transformed fine and coarse fall inside [-1, 1] (bit-exact mask)
    b = onp.Basis(onp.LEGENDRE, R, dom)
    for l, (f, c) in enumerate(host):
        keep = ~np.isnan(onp.transform(b, f))
        if c is not None:
            keep &= ~np.isnan(onp.transform(b, c))
        assert int(keep.sum()) == n[l]
    # additivity: two halves pushed separately give the same sums
    acc2 = LevelAccumulator(fn, L)
    for l in range(L):
        f, c = dev[l]
        h = n_l // 2 + 3
        acc2.push(l, f[:h], None if c is None else c[:h])
        acc2.push(l, f[h:], None if c is None else c[h:])
    n2, n_rm2, s2, sp2 = acc2.finalize()
    assert np.array_equal(n, n2) and np.array_equal(n_rm, n_rm2)
    rms = np.sqrt(sp / n[:, None]) * n[:, None]
    assert close(s2, s, rms, 1e-12) and close(sp2, sp, None, 1e-12)
    # C oracle on a prefix
    k = 2_000_000
    accp = LevelAccumulator(fn, L)
    for l in range(L):
        f, c = dev[l]
        accp.push(l, f[:k], None if c is None else c[:k])
    n3, n_rm3, s3, sp3 = accp.finalize()
    for l, (f, c) in enumerate(host):
        nk, nr, so, spo = oracle_c.moments_level(b, f[:k], None if c is None else c[:k])
        assert nk == n3[l] and nr == n_rm3[l]
        rms_l = np.sqrt(spo / nk) * nk
        assert close(s3[l], so, rms_l, 1e-10) and close(sp3[l], spo, None, 1e-10)


@pytest.mark.parametrize("R", [65, 80, 128, 129, 200, 320])
def test_covariance_more_than_64_moments(hip, R):
    """R > 64: the covariance is assembled from 64 x 64 blocks -- up to 128 moments from term windows evaluated in registers
    (two LDS windows off the diagonal), beyond that (the reference has no size limit, quantity_estimate.py:122-156; the library
    accepts 512 Legendre moments) from moment values materialised chunk by chunk, the window offsets run-time arguments."""
    from mlmc_amd import Legendre, TransformedMoments
    from mlmc_amd.engine import LevelAccumulator
    dom = (-3.7, 3.7)
    levels = level_arrays([1500, 901], [0.3, 0.02], 1, 7)
    b = onp.Basis(onp.LEGENDRE, R, dom)
    n, n_rm, s, sp = _run_accum(Legendre(R, dom), levels, mode=LevelAccumulator.COV)
    for l, (f, c) in enumerate(levels):
        nk, nr, so, spo = oracle_c.cov_level(b, f[0], None if c is None else c[0])
        assert nk == n[l] and nr == n_rm[l]
        rms = np.sqrt(spo / max(nk, 1)) * nk
        assert close(s[l], so, rms, TOL) and close(sp[l], spo, None, TOL)
    if R == 80:   # variance of transformed moments with more than 64 underlying moments (wide difference Gram)
        rng = np.random.default_rng(5)
        mat = rng.normal(size=(70, R)) / np.sqrt(R)
        mat[0] = 0
        mat[0, 0] = 1
        n, n_rm, s, sp = _run_accum(TransformedMoments(Legendre(R, dom), mat), levels)
        bt = onp.Basis(onp.LEGENDRE, R, dom, matrix=mat)
        ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.moments_rows(bt, x))
        _check_against(n, n_rm, s, sp, ref)


def test_covariance_of_130_moments_of_a_vector_quantity(hip):
    """More than 128 moments (values path, 64 x 64 blocks) with two components sharing one mask: every component equals the
    scalar estimate of its own samples when no sample is masked, and the shared mask drops a sample for both."""
    from mlmc_amd import Legendre
    from mlmc_amd.engine import LevelAccumulator
    dom = (-3.7, 3.7)
    R = 130
    levels = level_arrays([700, 401], [0.3, 0.02], 2, 0)
    clean = [(np.clip(f, -3.6, 3.6), None if c is None else np.clip(c, -3.6, 3.6)) for f, c in levels]
    n, n_rm, s, sp = _run_accum(Legendre(R, dom), clean, mode=LevelAccumulator.COV, n_comp=2)
    assert np.all(n_rm == 0)
    for m in range(2):
        one = [(f[m:m + 1], None if c is None else c[m:m + 1]) for f, c in clean]
        n1, _, s1, sp1 = _run_accum(Legendre(R, dom), one, mode=LevelAccumulator.COV)
        assert np.array_equal(n, n1)
        blk = slice(m * R * R, (m + 1) * R * R)
        assert np.array_equal(s[:, blk], s1) and np.array_equal(sp[:, blk], sp1)
    masked = [(f.copy(), None if c is None else c.copy()) for f, c in clean]
    masked[0][0][1, 5] = np.nan                     # component 1 of sample 5 at level 0
    masked[1][1][0, 7] = 9.0                        # coarse value of component 0 outside the domain at level 1
    n2, n_rm2, s2, _ = _run_accum(Legendre(R, dom), masked, mode=LevelAccumulator.COV, n_comp=2)
    assert list(n_rm2) == [1, 1] and np.array_equal(n2, n - 1)
    assert not np.array_equal(s2[:, :R * R], s[:, :R * R])           # the other component lost the sample too


def test_rccl_allreduce_path_single_rank(hip):
    """The N > 1 exchange step (finalize into device buffers + torch.distributed all-reduce, backend nccl = RCCL) run
    with one rank in a child process: bench.py with MLMC_HIP_FORCE_DIST=1 must give the same estimate."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MLMC_HIP_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    # the default N = 1 workload: BASELINE configs[2] (covariance R = 64 through mlmc_accum_estimate_packed + the all-reduce)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["result_check"]["mean0"] == 1.0 and d["result_check"]["var0"] == 0.0
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"]["estimate"] == "cov" and d["exchange"]["bytes_per_rank"] == 8 * (2 * 5 + 2 * 5 * 64 * 64)
    ref = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"],
                         capture_output=True, text=True, timeout=600)
    d0 = json.loads([l for l in ref.stdout.splitlines() if l.startswith("{")][-1])
    assert d0["result_check"] == d["result_check"] and len(d["result_check"]["n_estimated"]) == 5
    # the moments mode through the same path (BASELINE configs[1])
    for e in (env, None):
        o = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                           env=e, capture_output=True, text=True, timeout=600)
        assert o.returncode == 0, o.stderr[-2000:]
        dm = json.loads([l for l in o.stdout.splitlines() if l.startswith("{")][-1])
        assert dm["result_check"]["mean0"] == 1.0 and dm["result_check"]["var0"] == 0.0
        if e is None:
            assert dm["result_check"] == last
        last = dm["result_check"]


def test_covariance_full_size_properties(hip):
    """BASELINE configs[2] scale (5 levels x 1e7 samples, R = 64) from HBM-resident tensors: size-independent properties
    of the covariance estimate + C-oracle parity on a prefix."""
    import torch
    from mlmc_amd import Legendre
    from mlmc_amd.engine import LevelAccumulator, level_stats
    dom = (-3.7190164854556804, 3.7190164854556804)
    L, n_l, R = 5, 10_000_000, 64
    steps = [s[0] for s in onp.determine_level_parameters(L, [0.5, 0.01])]
    fn = Legendre(R, dom)
    dev = []
    gen = torch.Generator(device="cuda")
    for l in range(L):
        gen.manual_seed(77 + l)
        x = torch.randn(n_l, dtype=torch.float64, device="cuda", generator=gen)
        root = torch.sqrt(1e-4 + x.abs())
        dev.append(((x + steps[l] * root).contiguous(), None if l == 0 else (x + steps[l - 1] * root).contiguous()))
    acc = LevelAccumulator(fn, L, LevelAccumulator.COV)
    accm = LevelAccumulator(fn, L)
    for l in range(L):
        acc.push(l, *dev[l])
        accm.push(l, *dev[l])
    n, n_rm, s, sp = acc.finalize()
    nm, n_rm_m, sm, spm = accm.finalize()
    assert np.array_equal(n, nm) and np.array_equal(n_rm, n_rm_m) and np.all(n + n_rm == n_l)
    S = s.reshape(L, R, R)
    SP = sp.reshape(L, R, R)
    assert np.array_equal(S, S.transpose(0, 2, 1)) and np.array_equal(SP, SP.transpose(0, 2, 1))      # exactly symmetric
    assert S[0, 0, 0] == float(n[0]) and not S[1:, 0, 0].any()                                       # P0 P0: counts / exact zeros
    # first row/column of the covariance sums = the moment sums; squares likewise (P0 = 1)
    rms = np.sqrt(spm / n[:, None]) * n[:, None]
    assert close(S[:, 0, :], sm, rms, 1e-10) and close(SP[:, 0, :], spm, None, 1e-10)
    assert np.all(SP >= 0)
    # run to run: bitwise (static batch schedule, fixed-order reduction; the issue priorities of the kernel change timing only)
    for l in range(L):
        acc.push(l, *dev[l]) if l else (acc.reset(), acc.push(l, *dev[l]))
    again = acc.finalize()
    assert np.array_equal(again[0], n) and np.array_equal(again[2], s) and np.array_equal(again[3], sp)
    # Cauchy-Schwarz on the level-0 second moments: (sum f_i f_j)^2 <= (sum f_i^2)(sum f_j^2)
    d0 = np.diag(S[0])
    assert np.all(S[0] ** 2 <= np.outer(d0, d0) * (1 + 1e-12))
    # C oracle on a short prefix (the reference form costs O(R^2) per sample)
    k = 20000
    accp = LevelAccumulator(fn, L, LevelAccumulator.COV)
    for l in range(L):
        f, c = dev[l]
        accp.push(l, f[:k], None if c is None else c[:k])
    n3, n_rm3, s3, sp3 = accp.finalize()
    b = onp.Basis(onp.LEGENDRE, R, dom)
    for l in (0, 2, 4):
        f, c = dev[l]
        nk, nr, so, spo = oracle_c.cov_level(b, f[:k].cpu().numpy(), None if c is None else c[:k].cpu().numpy())
        assert nk == n3[l] and nr == n_rm3[l]
        rms_l = np.sqrt(spo / nk) * nk
        assert close(s3[l], so, rms_l, 1e-10) and close(sp3[l], spo, None, 1e-10)


def test_percentiles_bit_identical_to_numpy(hip):
    """mlmc_percentiles (device radix select) == np.percentile on the NaN-free values, bit for bit."""
    import torch
    from mlmc_amd.engine import percentiles
    rng = np.random.default_rng(11)
    cases = [rng.normal(size=100003), rng.normal(size=7) * 1e-300, np.array([3.0]), np.array([2.0, -1.0]),
             np.concatenate([rng.normal(size=5000), [np.nan] * 17, [np.inf, -np.inf, 0.0, -0.0]]),
             np.round(rng.normal(size=20000), 1),                      # many ties
             np.round(rng.normal(size=400000), 0),                     # ties beyond the host-finish buffer: all six digit passes
             np.concatenate([np.full(300000, 1.5), rng.normal(size=1000)]),
             -np.abs(rng.lognormal(size=30011)) * 1e5]
    qs = [0.0, 1.0, 0.01, 25.0, 50.0, 99.0, 99.999, 100.0]
    for x in cases:
        ref = np.percentile(x[~np.isnan(x)], qs)
        got = percentiles(x, qs)
        assert np.array_equal(got, ref, equal_nan=True), (x.size, got, ref)    # +-inf neighbours interpolate to NaN in NumPy too
    x = rng.normal(size=3_000_001)
    xt = torch.from_numpy(x).cuda()
    assert np.array_equal(percentiles(xt, [1.0, 99.0]), np.percentile(x, [1.0, 99.0]))


def _sharded_worker(rank, world, port, N, steps, R, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mlmc_amd import _lib, Legendre
        from mlmc_amd.engine import LevelAccumulator, shard_bounds
        _lib.init(0)                                   # both ranks share the one GPU of the test box
        dom = (-3.7190164854556804, 3.7190164854556804)
        levels = level_arrays(N, steps, 1, 13)
        for mode, tag in ((LevelAccumulator.MOMENTS, "mom"), (LevelAccumulator.COV, "cov")):
            acc = LevelAccumulator(Legendre(R, dom), len(N), mode)
            for l, (f, c) in enumerate(levels):
                lo, hi = shard_bounds(f.shape[1], rank, world)
                acc.push(l, f[0, lo:hi], None if c is None else c[0, lo:hi])
            n, n_rm, s, sp = acc.finalize()            # all-reduce over the two ranks inside
            np.savez(os.path.join(out_dir, f"{tag}_rank{rank}.npz"), n=n, n_rm=n_rm, s=s, sp=sp)
            # the one-call form over device-resident shards (mlmc_accum_estimate_packed + the same all-reduce)
            import torch
            dev = torch.device("cuda", 0)
            chunks = []
            for l, (f, c) in enumerate(levels):
                lo, hi = shard_bounds(f.shape[1], rank, world)
                chunks.append((l, torch.from_numpy(f[0, lo:hi].copy()).to(dev),
                               None if c is None else torch.from_numpy(c[0, lo:hi].copy()).to(dev)))
            torch.cuda.synchronize()
            again = acc.estimate(chunks)
            for a, b in zip(again, (n, n_rm, s, sp)):
                assert np.array_equal(a, b), tag
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_estimate_on_gpu(hip, tmp_path):
    """The N > 1 path with real kernels: two processes (gloo transport, both on the single GPU of the box) push disjoint
    shards of every level; after the packed all-reduce every rank holds the same sums as one process over all samples."""
    import socket
    import torch.multiprocessing as mp
    from mlmc_amd import Legendre
    from mlmc_amd.engine import LevelAccumulator
    N, steps, R = [50001, 30000, 17777], [0.5, 0.07, 0.01], 12
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_sharded_worker, args=(2, port, N, steps, R, str(tmp_path)), nprocs=2, join=True)
    dom = (-3.7190164854556804, 3.7190164854556804)
    levels = level_arrays(N, steps, 1, 13)
    for mode, tag in ((LevelAccumulator.MOMENTS, "mom"), (LevelAccumulator.COV, "cov")):
        n, n_rm, s, sp = _run_accum(Legendre(R, dom), levels, mode=mode)
        r0, r1 = np.load(tmp_path / f"{tag}_rank0.npz"), np.load(tmp_path / f"{tag}_rank1.npz")
        for k in ("n", "n_rm", "s", "sp"):
            assert np.array_equal(r0[k], r1[k])
        assert np.array_equal(r0["n"], n) and np.array_equal(r0["n_rm"], n_rm)          # counts reduce exactly
        scale = np.sqrt(np.abs(sp) * np.maximum(n[:, None], 1))
        assert close(r0["s"], s, scale, 1e-12) and close(r0["sp"], sp, None, 1e-12)


def _sharded_api_worker(rank, world, port, n_total, steps, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        from mlmc_amd import _lib, Legendre
        from mlmc_amd.estimator import Estimate
        from mlmc_amd.quantity.quantity import make_root_quantity
        from mlmc_amd.sim.synth_device import SynthDeviceStorage
        _lib.init(0)
        st = SynthDeviceStorage(steps, n_total, shard=(rank, world), chunk_size=20000)
        root = make_root_quantity(st, st.load_result_format())
        q = (root['length'][1]['10'][0] - 0.25) * root['width'][2]['40'][1]
        est = Estimate(q, st, Legendre(8, (-12.0, 20.0)))
        means, variances = est.estimate_moments()
        cov, _ = est.estimate_covariance()
        # construct_density over the shards: the linearised covariance-mean pass and the orthogonal-moments pass are both
        # all-reduced (the kept-sums shortcut of the single-process chain is off under a process group)
        distr, _, res, _ = est.construct_density(tol=1e-8)
        dens = distr.density(np.linspace(-12.0, 20.0, 201))
        np.savez(os.path.join(out_dir, f"api_rank{rank}.npz"), means=means, variances=variances, cov=cov, dens=dens,
                 ok=np.array(bool(res.success)))
    finally:
        dist.destroy_process_group()


def test_two_rank_estimate_through_the_python_api(hip, tmp_path):
    """Whole Python path with two ranks (gloo transport, one GPU): every rank generates ITS shard of the synthetic levels
    in HBM (SynthDeviceStorage(shard=...)), evaluates the quantity tree on the device, and Estimate all-reduces the level
    sums; both ranks end with the estimate one process computes over all samples."""
    import socket
    import torch.multiprocessing as mp
    from mlmc_amd import Legendre
    from mlmc_amd.estimator import Estimate
    from mlmc_amd.quantity.quantity import make_root_quantity
    from mlmc_amd.sim.synth_device import SynthDeviceStorage
    n_total, steps = [70001, 40000, 21111], [[0.5], [0.1], [0.02]]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_sharded_api_worker, args=(2, port, n_total, steps, str(tmp_path)), nprocs=2, join=True)
    st = SynthDeviceStorage(steps, n_total)
    root = make_root_quantity(st, st.load_result_format())
    q = (root['length'][1]['10'][0] - 0.25) * root['width'][2]['40'][1]
    est = Estimate(q, st, Legendre(8, (-12.0, 20.0)))
    means, variances = est.estimate_moments()
    cov, _ = est.estimate_covariance()
    distr, _, res, _ = est.construct_density(tol=1e-8)
    r0, r1 = np.load(tmp_path / "api_rank0.npz"), np.load(tmp_path / "api_rank1.npz")
    for k in ("means", "variances", "cov", "dens"):
        assert np.array_equal(r0[k], r1[k])
    assert close(r0["means"], means, 1.0, 1e-12) and close(r0["variances"], variances, np.max(variances), 1e-12)
    assert close(r0["cov"], cov, 1.0, 1e-12)
    assert res.success and bool(r0["ok"])
    assert np.max(np.abs(r0["dens"] - distr.density(np.linspace(-12.0, 20.0, 201)))) < 1e-7


def test_spline_moments(hip):
    """Cubic B-spline moments -- NOT in the reference ("parity unpinned"): device evaluation and estimates
    against scipy.interpolate.BSpline through the oracle."""
    from mlmc_amd import Spline
    from mlmc_amd.engine import LevelAccumulator
    dom = (-3.7, 3.7)
    grid = np.concatenate([np.linspace(dom[0], dom[1], 257), [dom[0] - 1e-9, dom[1] + 1e-9, np.nan, 0.0, 1e-300]])
    for R in (4, 5, 8, 33, 64, 128):
        fn = Spline(R, dom)
        got = fn.eval_all(grid)
        ref = onp.eval_all(onp.Basis(onp.SPLINE, R, dom), grid)
        assert got.shape == ref.shape and _same_nan(got, ref) and _vals_close(got, ref, 1e-12), R
        ok = ~np.isnan(ref[:, 0])
        assert np.all(got[ok, 0] == 1.0)
        assert np.allclose(got[ok, 1:].sum(axis=1) + (1 - got[ok, 1:].sum(axis=1)), 1.0)      # B_0 = 1 - sum of the others
        assert np.all(got[ok] >= 0) and np.all(got[ok] <= 1 + 1e-15)
    # knots of the class are those of the scipy basis
    from scipy.interpolate import BSpline
    fn = Spline(10, (0.0, 1.0))
    x = np.linspace(0, 1, 50)
    dm = BSpline.design_matrix(x, fn.knots(), 3).toarray()
    assert np.allclose(fn.eval_all(x)[:, 1:], dm[:, 1:], atol=1e-14)
    levels = level_arrays([6001, 4000, 1501], [0.5, 0.07, 0.01], 1, 11)
    for R in (6, 40, 128):
        b = onp.Basis(onp.SPLINE, R, dom)
        n, n_rm, s, sp = _run_accum(Spline(R, dom), levels)
        ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.moments_rows(b, x))
        mean, var = _check_against(n, n_rm, s, sp, ref)
        assert mean[0] == 1.0 and var[0] == 0.0
    # covariance: every kernel family (16- and 32-term tiles, the 64 x 64 kernel, two term windows), values placed by index
    from mlmc_amd.engine import level_stats
    for R, cut in ((10, 1200), (24, 1200), (60, 900), (70, 700)):
        b = onp.Basis(onp.SPLINE, R, dom)
        lv = [(f[:, :cut], None if c is None else c[:, :cut]) for f, c in levels]
        n, n_rm, s, sp = _run_accum(Spline(R, dom), lv, mode=LevelAccumulator.COV)
        ref = onp.estimate_mean(to_chunks(lv), lambda x: onp.covariance_rows(b, x))
        assert np.array_equal(n, ref.n_samples) and np.array_equal(n_rm, ref.n_rm_samples)
        l_means, l_vars = level_stats(n, s, sp)
        rms = np.sqrt(np.abs(ref.sums_sq) / np.maximum(ref.n_samples[:, None], 1))
        assert close(l_means, ref.l_means, rms, TOL)
        # products of splines three knot spans apart are 1e-7 of the level's scale: the parity gate is relative to
        # max(|b|, scale of the level); such entries carry ~2e-10 relative error in the d / s formulation
        assert close(l_vars, ref.l_vars, 1e-6 * np.max(ref.l_vars, axis=1, keepdims=True), TOL)


@pytest.mark.parametrize("R", [4, 8, 37, 128, 200])
def test_spline_covariance_mean_is_banded(hip, R):
    """Mean-only covariance of spline moments: phi_i phi_j = 0 for |i - j| > 3, so the level sums are accumulated as a band
    (k_spline_band_accum, 14 LDS atomics per value, no matrix cores) -- against the dense matrix-core pass of the same library
    (mean + variance; R > 128: from materialised values) and the scipy-BSpline oracle; pairs in one span / in different
    spans, level 0, masked samples, a vector quantity."""
    from mlmc_amd import Spline
    from mlmc_amd.engine import LevelAccumulator
    dom = (-3.7, 3.7)
    levels = level_arrays([9001, 5000, 2501], [0.9, 0.07, 0.01], 1, 17)       # level 1: a coarse step large enough to cross spans
    fn = Spline(R, dom)
    band = _run_accum(fn, levels, mode=LevelAccumulator.COV | 0)                # dense, for the counts
    acc = LevelAccumulator(fn, len(levels), LevelAccumulator.COV, mean_only=True)
    for l, (f, c) in enumerate(levels):
        acc.push(l, f[0], None if c is None else c[0])
    n1, r1, s1, sp1 = acc.finalize()
    n0, r0, s0, sp0 = band
    assert np.array_equal(n0, n1) and np.array_equal(r0, r1) and np.all(np.isnan(sp1))
    scale = np.sqrt(np.abs(sp0) * n0[:, None]) + 1e-300
    assert np.max(np.abs(s1 - s0) / np.maximum(np.abs(s0), scale)) < 1e-10     # the parity gate (the dense pass forms 1/2 (d s + s d))
    S = s1.reshape(len(levels), R, R)
    i, j = np.indices((R, R))
    off_band = (np.abs(i - j) > 3) & (i > 0) & (j > 0)
    assert np.all(S[:, off_band] == 0.0) and np.array_equal(S, S.transpose(0, 2, 1))
    if R <= 37:
        b = onp.Basis(onp.SPLINE, R, dom)
        ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.covariance_rows(b, x))
        assert np.array_equal(n1, ref.n_samples) and np.array_equal(r1, ref.n_rm_samples)
        rms = np.sqrt(np.abs(ref.sums_sq) * ref.n_samples[:, None])
        assert np.all(np.abs(s1 - ref.sums) <= 1e-10 * np.maximum(np.abs(ref.sums), rms) + 1e-300)
    # two components: one mask for both
    lv2 = level_arrays([3001, 2000], [0.5, 0.07], 2, 11)
    full = _run_accum(fn, lv2, mode=LevelAccumulator.COV, n_comp=2)
    acc = LevelAccumulator(fn, 2, LevelAccumulator.COV, n_comp=2, mean_only=True)
    for l, (f, c) in enumerate(lv2):
        acc.push(l, f, c)
    n2, r2, s2, _ = acc.finalize()
    assert np.array_equal(n2, full[0]) and np.array_equal(r2, full[1])
    sc2 = np.sqrt(np.abs(full[3]) * full[0][:, None]) + 1e-300
    assert np.max(np.abs(s2 - full[2]) / np.maximum(np.abs(full[2]), sc2)) < 1e-12


def test_covariance_of_transformed_moments(hip):
    """estimate_covariance with a TransformedMoments basis: transformed values are materialised chunk-wise on the device
    and fed to the MFMA covariance kernel; against the oracle (reference form, per-sample matrix product)."""
    from mlmc_amd import Legendre, TransformedMoments
    from mlmc_amd.engine import LevelAccumulator
    dom = (-3.7, 3.7)
    levels = level_arrays([3000, 2001, 700], [0.5, 0.07, 0.01], 1, 9)
    rng = np.random.default_rng(8)
    for R0, R1 in ((9, 6), (33, 33), (64, 40), (80, 70), (40, 100), (150, 130)):      # more than 64 rows: 64 x 64 output blocks
        mat = rng.normal(size=(R1, R0)) / np.sqrt(R0)
        mat[0] = 0
        mat[0, 0] = 1
        n, n_rm, s, sp = _run_accum(TransformedMoments(Legendre(R0, dom), mat), levels, mode=LevelAccumulator.COV)
        bt = onp.Basis(onp.LEGENDRE, R0, dom, matrix=mat)
        ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.covariance_rows(bt, x))
        _check_against(n, n_rm, s, sp, ref)


def test_log_transform_and_other_families_accumulate(hip):
    """log=True moments (device log), Monomial / Fourier covariance, multi-component covariance."""
    from mlmc_amd import Fourier, Legendre, Monomial
    from mlmc_amd.engine import LevelAccumulator
    rng = np.random.default_rng(21)
    # log-normal samples, log-domain Legendre / Monomial moments
    x = rng.lognormal(mean=0.3, sigma=0.8, size=20011)
    f1 = x * (1 + 0.01 * rng.normal(size=x.size))
    c1 = x * (1 + 0.03 * rng.normal(size=x.size))
    f1[::501] = -1.0                                     # log of a non-positive value -> NaN -> masked
    levels = [(x[None], None), (f1[None], c1[None])]
    ldom = (0.05, 30.0)
    for cls, kind, R in ((Legendre, onp.LEGENDRE, 12), (Monomial, onp.MONOMIAL, 6)):
        n, n_rm, s, sp = _run_accum(cls(R, ldom, log=True), levels)
        b = onp.Basis(kind, R, ldom, log=True)
        ref = onp.estimate_mean(to_chunks(levels), lambda v: onp.moments_rows(b, v))
        _check_against(n, n_rm, s, sp, ref)
    # Monomial and Fourier covariance
    dom = (-3.7, 3.7)
    lv = level_arrays([2500, 1500], [0.3, 0.02], 1, 7)
    for cls, kind, R in ((Monomial, onp.MONOMIAL, 7), (Fourier, onp.FOURIER, 9)):
        n, n_rm, s, sp = _run_accum(cls(R, dom), lv, mode=LevelAccumulator.COV)
        b = onp.Basis(kind, R, dom)

        def rows(v, b=b, R=R):
            phi = onp.eval_all(b, v.reshape(-1)).reshape(v.shape + (R,))
            phi[np.isnan(phi).any(axis=-1)] = np.nan
            cov = np.einsum('...i,...j', phi, phi)                   # [M, n, 2|1, R, R]
            return cov.transpose((0, 3, 4, 1, 2)).reshape(R * R, v.shape[1], v.shape[2])
        ref = onp.estimate_mean(to_chunks(lv), rows)
        _check_against(n, n_rm, s, sp, ref)
    # covariance of a 3-component quantity (a sample is dropped when any component is masked)
    lv3 = level_arrays([1800, 1100], [0.3, 0.02], 3, 6)
    n, n_rm, s, sp = _run_accum(Legendre(10, dom), lv3, mode=LevelAccumulator.COV, n_comp=3)
    b = onp.Basis(onp.LEGENDRE, 10, dom)
    ref = onp.estimate_mean(to_chunks(lv3), lambda v: onp.covariance_rows(b, v))
    _check_against(n, n_rm, s, sp, ref)


def test_synth_generation_matches_the_reference_chain(hip):
    """mlmc_synth_generate against (a) the reference's own outputs in G7_chain.json (sample ids -> md5 seeds ->
    SynthSimulation.calculate, 3 levels x 10 samples x 24 rows, written by oracle/gen_golden.py from the imported
    reference) and (b) NumPy's RandomState -- the third-party generator the reference calls -- for 10^5 more samples.
    Seeds (integer work) must be bit-exact; sample values are bit-exact wherever the device log() and glibc's log()
    round alike, otherwise 1 ulp apart."""
    import hashlib
    import json
    from mlmc_amd.sim import synth_device as sd
    g7 = json.load(open(os.path.join(GOLDEN, "G7_chain.json")))
    steps = [0.01, 0.001, 0.0001]
    for l, lv in enumerate(g7["levels"]):
        seeds = sd.sample_seeds(l, 0, 10)
        assert seeds.tolist() == lv["seeds"]
        rows = sd.generate_rows(l, 0, 10, steps[l], steps[l - 1] if l else 0.0, list(range(24)), loc=1.0, scale=2.0)
        hip.check(hip.lib().mlmc_synchronize())
        got = np.stack([t.cpu().numpy() for t in rows])                      # [24, 10, 2|1]
        fine, coarse = np.array(lv["fine"]).T, np.array(lv["coarse"]).T      # [24, 10]
        assert np.allclose(got[:, :, 0], fine, rtol=4e-16, atol=0) and np.mean(got[:, :, 0] == fine) > 0.9
        if l:
            assert np.allclose(got[:, :, 1], coarse, rtol=4e-16, atol=0)
        else:
            assert got.shape[2] == 1 and not np.any(coarse)
    # sample ids beyond 7 digits and other levels: seeds against hashlib
    for level, first in ((0, 9_999_995), (7, 123), (12, 99_999_990)):
        seeds = sd.sample_seeds(level, first, 12)
        want = [int(np.frombuffer(hashlib.md5("L{:02d}_S{:07d}".format(level, first + i).encode("ascii")).digest(), dtype="uint32")[0])
                for i in range(12)]
        assert seeds.tolist() == want
    # 10^5 samples against numpy.random.RandomState (legacy seeding + legacy_gauss)
    n, level, h_f, h_c = 100_000, 3, 0.02, 0.1
    rows = sd.generate_rows(level, 0, n, h_f, h_c, [0, 1, 3])
    hip.check(hip.lib().mlmc_synchronize())
    got = np.stack([t.cpu().numpy() for t in rows])
    seeds = sd.sample_seeds(level, 0, n)
    y = np.array([np.random.RandomState(int(s)).standard_normal(2) for s in seeds])          # [n, 2]
    fn = lambda x, h: x + h * np.sqrt(1e-4 + np.abs(x))
    want = np.stack([np.stack([fn(y[:, 0], h_f), fn(y[:, 0], h_c)], axis=1),
                     np.stack([fn(y[:, 1], h_f), fn(y[:, 1], h_c)], axis=1),
                     np.stack([fn(y[:, 1], h_f) + 1, fn(y[:, 1], h_c) + 1], axis=1)])
    exact = np.mean(got == want)
    # a 1-ulp difference of the normal (|y| < 8) is at most 8.9e-16 in absolute terms, also after x + h sqrt(1e-4 + |x|)
    assert np.max(np.abs(got - want)) < 2e-15, np.max(np.abs(got - want))
    assert exact > 0.99, exact


def test_maxent_cooperative_launch_matches_the_step_by_step_solver(hip):
    """k_me_coop (whole Newton iteration in one cooperative launch) against the kernel-per-step loop it replaces
    (MLMC_MAXENT_STEPWISE=1), on moments of known densities: same convergence flag, multipliers, gradient, Hessian;
    including a far-off start (regularisation + backtracking), an unreachable tolerance (iteration cap), R1 = 1, 2,
    64 / 65 (both layouts of the slices) and 128, a finer quadrature, and the moments of the solution reproduced."""
    from scipy import stats
    from mlmc_amd import Legendre
    from mlmc_amd.tool import simple_distribution as sd
    dom = (-4.0, 6.0)
    pdf = lambda x: 0.6 * stats.norm(0.5, 1.0).pdf(x) + 0.4 * stats.norm(2.5, 0.7).pdf(x)
    cases = [(1, None, 1e-8, 100, 0), (2, None, 1e-8, 100, 0), (9, None, 1e-10, 100, 0), (26, None, 1e-8, 100, 0),
             (26, "far", 1e-8, 100, 0), (26, None, 1e-300, 7, 0), (64, None, 1e-8, 100, 0), (65, None, 1e-8, 100, 0),
             (128, None, 1e-7, 100, 0), (21, None, 1e-9, 100, 200)]
    for R, start, tol, max_it, n_int in cases:
        fn = Legendre(R, dom)
        mom = sd.compute_semiexact_moments(fn, pdf)
        err = np.ones(R)
        lam0 = np.zeros(R)
        lam0[0] = -np.log(1.0 / (dom[1] - dom[0]))
        if start == "far":
            lam0 = lam0 + 3.0 * np.sin(np.arange(R))
        out = {}
        for mode in ("coop", "step"):
            if mode == "step":
                os.environ["MLMC_MAXENT_STEPWISE"] = "1"
            try:
                out[mode] = sd._solve_on_device(fn, mom, err, dom, lam0, tol, max_it, n_intervals=n_int)
            finally:
                os.environ.pop("MLMC_MAXENT_STEPWISE", None)
        (l1, g1, h1, i1), (l2, g2, h2, i2) = out["coop"], out["step"]
        tag = (R, start, tol)
        assert i1.success == i2.success and abs(i1.nit - i2.nit) <= 1, (tag, i1.nit, i2.nit, i1.success, i2.success)
        if tol > 1e-100:
            assert i1.success == 1 and i1.grad_norm < tol, (tag, i1.grad_norm)
            scale = max(1.0, np.max(np.abs(l2)))
            assert np.max(np.abs(l1 - l2)) < 1e-6 * scale, (tag, np.max(np.abs(l1 - l2)))
            assert np.allclose(h1, h2, rtol=1e-6, atol=1e-9 * np.max(np.abs(h2))), tag
            assert np.allclose(h1, h1.T) and np.all(np.linalg.eigvalsh(h1) > 0), tag
            # the moments of the reconstructed density are the prescribed ones
            got = sd.compute_semiexact_moments(fn, lambda x: sd._device_density(fn, l1, err, x))
            assert np.max(np.abs(got - mom)) < 50 * tol + 1e-7, (tag, np.max(np.abs(got - mom)))
        else:
            assert i1.success == 0 and i1.nit <= max_it


@pytest.mark.parametrize("R", [17, 24, 32])
def test_covariance_two_tile_sizes(hip, R):
    """17..32 moments: two 16-term tiles per dimension.  A wave owns a k-slice and all four tiles, and the lower tile of the
    symmetric Gram matrices (everything at level 0, the third Gram matrix of a pair level) is not computed but mirrored:
    level sums and their variances against the oracle, with NaN samples, for Legendre and Monomial; the mean-only mode
    against the full one; transformed moments (values path) against the oracle."""
    from mlmc_amd import Legendre, Monomial, TransformedMoments
    from mlmc_amd.engine import LevelAccumulator
    dom = (-3.7190164854556804, 3.7190164854556804)
    levels = level_arrays([2301, 1500, 777], [0.5, 0.07, 0.01], 1, 19)
    for cls, kind in ((Legendre, onp.LEGENDRE), (Monomial, onp.MONOMIAL)):
        b = onp.Basis(kind, R, dom)
        ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.covariance_rows(b, x))
        n, n_rm, s, sp = _run_accum(cls(R, dom), levels, mode=LevelAccumulator.COV)
        mean, var = _check_against(n, n_rm, s, sp, ref)
        cov = mean.reshape(R, R)
        assert np.array_equal(cov, cov.T) and np.array_equal(var.reshape(R, R), var.reshape(R, R).T)
        acc = LevelAccumulator(cls(R, dom), len(levels), LevelAccumulator.COV, mean_only=True)
        for l, (f, c) in enumerate(levels):
            acc.push(l, f[0], None if c is None else c[0])
        n1, r1, s1, sp1 = acc.finalize()
        acc.close()
        scale = np.sqrt(np.abs(sp) * n[:, None]) + 1e-300
        assert np.array_equal(n1, n) and np.max(np.abs(s1 - s) / scale) < 1e-12
    rng = np.random.default_rng(R)
    mat = np.linalg.qr(rng.normal(size=(40, 40)))[0][:R]                      # R transformed moments of 40 Legendre ones
    tm = TransformedMoments(Legendre(40, dom), mat)
    n, n_rm, s, sp = _run_accum(tm, levels, mode=LevelAccumulator.COV)
    base = onp.Basis(onp.LEGENDRE, 40, dom)

    def rows(v):
        phi = onp.eval_all(base, v.reshape(-1)).reshape(v.shape + (40,)) @ mat.T
        phi[np.isnan(phi).any(axis=-1)] = np.nan
        cov = np.einsum('...i,...j', phi, phi)
        return cov.transpose((0, 3, 4, 1, 2)).reshape(R * R, v.shape[1], v.shape[2])
    _check_against(n, n_rm, s, sp, onp.estimate_mean(to_chunks(levels), rows))


@pytest.mark.parametrize("R", [8, 24, 64, 100])
def test_mean_only_accumulators(hip, R):
    """MLMC_MODE_MEAN_ONLY (what Estimate.construct_density asks for): the level sums equal those of the full estimate to
    rounding, the skipped second moments come back as NaN; covariance (one Gram matrix instead of three) and
    TransformedMoments (no diff-Gram pass)."""
    from mlmc_amd import Legendre, TransformedMoments
    from mlmc_amd.engine import LevelAccumulator
    dom = (-3.7190164854556804, 3.7190164854556804)
    levels = level_arrays([9001, 6000, 3500], [0.5, 0.07, 0.01], 1, 17)
    fn = Legendre(R, dom)

    def run(basis, mode, mean_only):
        acc = LevelAccumulator(basis, len(levels), mode, mean_only=mean_only)
        for l, (f, c) in enumerate(levels):
            acc.push(l, f[0], None if c is None else c[0])
        out = acc.finalize()
        acc.close()
        return out

    n0, r0, s0, sp0 = run(fn, LevelAccumulator.COV, False)
    n1, r1, s1, sp1 = run(fn, LevelAccumulator.COV, True)
    assert np.array_equal(n0, n1) and np.array_equal(r0, r1)
    scale = np.sqrt(np.abs(sp0) * n0[:, None]) + 1e-300
    assert np.max(np.abs(s1 - s0) / scale) < 1e-12 and np.all(np.isnan(sp1))
    if R <= 64:
        rng = np.random.default_rng(R)
        T = np.linalg.qr(rng.normal(size=(R, R)))[0][: max(2, R // 2)]
        tm = TransformedMoments(fn, T)
        n0, r0, s0, sp0 = run(tm, LevelAccumulator.MOMENTS, False)
        n1, r1, s1, sp1 = run(tm, LevelAccumulator.MOMENTS, True)
        assert np.array_equal(n0, n1) and np.array_equal(s0, s1) and np.all(np.isnan(sp1)) and np.all(np.isfinite(sp0))
    a, b = run(fn, LevelAccumulator.MOMENTS, False), run(fn, LevelAccumulator.MOMENTS, True)
    if 64 < R <= 128:
        # one pass of the mean-only term-split kernel (k_moments_accum_split<..., SQ = false>): counts identical, sums to rounding
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.all(np.isnan(b[3]))
        sc = np.sqrt(np.abs(a[3]) * a[0][:, None]) + 1e-300
        assert np.max(np.abs(a[2] - b[2]) / np.maximum(np.abs(a[2]), sc)) < 1e-12
    else:
        # other sizes of plain moments ignore the flag: their sum of squares costs nothing extra
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_estimate_in_one_call(hip):
    """mlmc_accum_estimate (reset + pushes + finalize in one C call) == the three-step protocol, bit for bit."""
    import torch
    from mlmc_amd import Legendre
    from mlmc_amd.engine import LevelAccumulator
    dev = torch.device("cuda", 0)
    levels = level_arrays([30001, 20000, 9000], [0.5, 0.07, 0.01], 1, 23)
    chunks = [(l, torch.from_numpy(f[0]).to(dev), None if c is None else torch.from_numpy(c[0]).to(dev))
              for l, (f, c) in enumerate(levels)]
    torch.cuda.synchronize()
    for mode, R in ((LevelAccumulator.MOMENTS, 13), (LevelAccumulator.COV, 9)):
        acc = LevelAccumulator(Legendre(R, (-3.7190164854556804, 3.7190164854556804)), 3, mode)
        for l, f, c in chunks:
            acc.push(l, f, c)
        ref = acc.finalize()
        for _ in range(2):                       # repeated use of the same accumulator
            got = acc.estimate(chunks)
            assert all(np.array_equal(a, b) for a, b in zip(got, ref))
        acc.close()


def test_many_levels_span_several_launches(hip):
    """20 levels: more segments than one launch carries (16) -- the estimate is split over two launches; moments and
    covariance against the oracle, device-resident and host pushes."""
    import torch
    from mlmc_amd import Legendre
    from mlmc_amd.engine import LevelAccumulator
    L = 20
    steps = [0.5 * 0.8 ** l for l in range(L)]
    N = [4000 - 150 * l for l in range(L)]
    levels = level_arrays(N, steps, 1, 9)
    dom = (-3.7190164854556804, 3.7190164854556804)
    dev = torch.device("cuda", 0)
    for R, mode, rows in ((11, LevelAccumulator.MOMENTS, onp.moments_rows), (6, LevelAccumulator.COV, onp.covariance_rows)):
        b = onp.Basis(onp.LEGENDRE, R, dom)
        ref = onp.estimate_mean(to_chunks(levels), lambda x: rows(b, x))
        for resident in (True, False):
            acc = LevelAccumulator(Legendre(R, dom), L, mode)
            keep = []
            for l, (f, c) in enumerate(levels):
                if resident:
                    ft = torch.from_numpy(f[0]).to(dev)
                    ct = None if c is None else torch.from_numpy(c[0]).to(dev)
                    keep.append((ft, ct))
                    torch.cuda.synchronize()
                    acc.push(l, ft, ct)
                else:
                    acc.push(l, f[0], None if c is None else c[0])
            n, n_rm, s, sp = acc.finalize()
            acc.close()
            _check_against(n, n_rm, s, sp, ref)


def test_adaptive_sampling_loop_matches_reference_trajectory(hip):
    """The loop of the reference's test/test_run.py:93-105 with every part on the device: samples generated in HBM
    (SynthDeviceStorage), level variances + regression from the device estimator, DeviceSampler scheduling.  Against the
    reference's own run of that loop (G9): per round the variances (1e-10), n_estimated and the scheduled / collected
    counts (exact), at the end the moments."""
    import json
    from mlmc_amd import Legendre
    from mlmc_amd.estimator import Estimate, estimate_n_samples_for_target_variance
    from mlmc_amd.quantity.quantity import make_root_quantity
    from mlmc_amd.sampler import DeviceSampler
    from mlmc_amd.sim.synth_device import SynthDeviceStorage, result_format
    with open(os.path.join(os.path.dirname(__file__), "golden", "G9_sampler_loop.json")) as f:
        cases = json.load(f)["cases"]
    for case in cases:
        steps = case["level_parameters"]
        st = SynthDeviceStorage(steps, [0] * len(steps), loc=case["loc"], scale=case["scale"])
        sampler = DeviceSampler(st, level_parameters=steps)
        fn = Legendre(case["n_moments"], tuple(case["domain"]))
        sampler.set_initial_n_samples(case["initial"])
        sampler.schedule_samples()
        sampler.ask_sampling_pool_for_samples()
        value = make_root_quantity(st, result_format())['length'][1]['10'][0]
        est = Estimate(value, st, fn)
        for i, rnd in enumerate(case["rounds"]):
            variances, n_ops = est.estimate_diff_vars_regression(sampler._n_scheduled_samples)
            want = np.array(rnd["variances"])
            assert np.allclose(variances, want, rtol=1e-9, atol=1e-10 * np.max(np.abs(want))), (case["name"], i)
            assert np.allclose(n_ops, rnd["n_ops"], rtol=1e-14)
            n_est = estimate_n_samples_for_target_variance(case["target_var"], variances, n_ops, n_levels=sampler.n_levels)
            assert [int(v) for v in n_est] == rnd["n_estimated"], (case["name"], i)
            done = sampler.process_adding_samples(n_est, 0, 0.1)
            assert [int(v) for v in sampler.l_scheduled_samples()] == rnd["n_scheduled"]
            assert [int(v) for v in sampler.n_finished_samples] == rnd["n_finished"]
            assert done == rnd["done"]
        assert done
        means, vars_ = est.estimate_moments(fn)
        assert np.allclose(means, case["means"], rtol=0, atol=1e-10) and np.allclose(vars_, case["vars"], rtol=1e-9, atol=1e-16)


@pytest.mark.parametrize("R", [17, 32, 33, 40, 64, 65, 100, 128])
def test_covariance_mean_through_the_product_linearisation(hip, R, monkeypatch):
    """Covariance WITH variances of 17..128 plain polynomial moments: the matrix cores accumulate the variance Grams only, the
    means come from the level sums of the 2 R - 1 moments of the family (mlmc_hip.h, mlmc_accum_aux_kernel_time).  Against the
    oracle, and against the same library with all three Gram matrices on the matrix cores (MLMC_HIP_LINEARIZE=0): identical
    counts, bit-identical second-moment sums, means equal to rounding; host chunks, several chunks per level, a two-component
    quantity with its shared mask, and log=True moments.  The choice is made per chunk (small chunks keep all three Gram
    matrices: the sums are additive), MLMC_HIP_LINEARIZE_MIN_N sets the chunk size from which the linearised way is taken:
    0 here, and 1500 for a mix of both kinds of chunk in one estimate."""
    import torch
    from mlmc_amd import Legendre, Monomial
    from mlmc_amd.engine import LevelAccumulator
    dom = (-3.7190164854556804, 3.7190164854556804)
    levels = level_arrays([5301, 2500, 1777] if R <= 64 else [901, 500, 377], [0.5, 0.07, 0.01], 1, 19)

    def both(fn, lv, n_comp=1, device=False, split=False, min_n="0"):
        out = []
        monkeypatch.setenv("MLMC_HIP_LINEARIZE_MIN_N", min_n)
        for lin in ("1", "0"):
            monkeypatch.setenv("MLMC_HIP_LINEARIZE", lin)
            acc = LevelAccumulator(fn, len(lv), LevelAccumulator.COV, n_comp=n_comp)
            keep = []
            for l, (f, c) in enumerate(lv):
                parts = [(0, f.shape[-1])] if not split else [(0, f.shape[-1] // 3), (f.shape[-1] // 3, f.shape[-1])]
                for lo, hi in parts:
                    fa = np.ascontiguousarray(f[:, lo:hi] if n_comp > 1 else f[0, lo:hi])
                    ca = None if c is None else np.ascontiguousarray(c[:, lo:hi] if n_comp > 1 else c[0, lo:hi])
                    if device:
                        fa = torch.from_numpy(fa).cuda()
                        ca = None if ca is None else torch.from_numpy(ca).cuda()
                        keep.append((fa, ca))
                    acc.push(l, fa, ca)
            out.append(acc.finalize())
            acc.close()
        monkeypatch.delenv("MLMC_HIP_LINEARIZE")
        monkeypatch.delenv("MLMC_HIP_LINEARIZE_MIN_N")
        (n, n_rm, s, sp), (n0, n_rm0, s0, sp0) = out
        assert np.array_equal(n, n0) and np.array_equal(n_rm, n_rm0)
        # pair levels: the same matrix instructions in the same order; levels without coarse values (<= 64 moments): the second
        # moments come from the level sums of 4 R - 3 moments (a convex combination of sums instead of sums of squares)
        pair = np.array([c is not None for _, c in lv])
        if fn.size > 64:
            assert np.array_equal(sp, sp0)
        else:
            assert np.array_equal(sp[pair], sp0[pair])
            big = np.max(np.abs(sp0[~pair]), axis=1, keepdims=True)
            assert np.max(np.abs(sp[~pair] - sp0[~pair]) / np.maximum(np.abs(sp0[~pair]), 1e-3 * big)) < 1e-11
        scale = np.sqrt(np.abs(sp0) * n[:, None]) + 1e-300
        assert np.max(np.abs(s - s0) / scale) < 1e-12, np.max(np.abs(s - s0) / scale)
        return n, n_rm, s, sp

    for cls, kind in ((Legendre, onp.LEGENDRE), (Monomial, onp.MONOMIAL)):
        b = onp.Basis(kind, R, dom)
        ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.covariance_rows(b, x))
        for device, split in ((False, False), (True, False), (True, True), (False, True)):
            n, n_rm, s, sp = both(cls(R, dom), levels, device=device, split=split)
            mean, var = _check_against(n, n_rm, s, sp, ref)
            cov = mean.reshape(R, R)
            assert np.array_equal(cov, cov.T)
            S = s.reshape(len(levels), R, R)
            assert S[0, 0, 0] == float(n[0]) and not S[1:, 0, 0].any()
        # chunks of 1767 | 3534, 833 | 1667, 592 | 1185 samples: direct and linearised contributions in one level
        n, n_rm, s, sp = both(cls(R, dom), levels, device=True, split=True, min_n="1500" if R <= 64 else "300")
        _check_against(n, n_rm, s, sp, ref)
    # every level in two chunks of 400 samples and the rest, threshold 1000: the small chunk keeps its Gram matrices (and is counted
    # by the covariance kernel), the large one goes through the extended moments (level 0: counted by the moments kernel) -- sums
    # and counts add up
    if R <= 64:
        monkeypatch.setenv("MLMC_HIP_LINEARIZE_MIN_N", "1000")
        acc = LevelAccumulator(Legendre(R, dom), len(levels), LevelAccumulator.COV)
        for l, (f, c) in enumerate(levels):
            for lo, hi in ((0, 400), (400, f.shape[-1])):
                acc.push(l, np.ascontiguousarray(f[0, lo:hi]), None if c is None else np.ascontiguousarray(c[0, lo:hi]))
        n, n_rm, s, sp = acc.finalize()
        acc.close()
        monkeypatch.delenv("MLMC_HIP_LINEARIZE_MIN_N")
        b = onp.Basis(onp.LEGENDRE, R, dom)
        _check_against(n, n_rm, s, sp, onp.estimate_mean(to_chunks(levels), lambda x: onp.covariance_rows(b, x)))
    # two components: a sample is dropped when any component is masked
    lv2 = level_arrays([2800, 1100] if R <= 64 else [700, 300], [0.3, 0.02], 2, 6)
    b = onp.Basis(onp.LEGENDRE, R, dom)
    ref = onp.estimate_mean(to_chunks(lv2), lambda v: onp.covariance_rows(b, v))
    for device in (False, True):
        n, n_rm, s, sp = both(Legendre(R, dom), lv2, n_comp=2, device=device)
        _check_against(n, n_rm, s, sp, ref)
    # log=True: the keep / drop decision is the raw-value interval of the caller's basis, for both accumulators
    rng = np.random.default_rng(5)
    x = rng.lognormal(mean=0.3, sigma=0.8, size=9011 if R <= 64 else 1511)
    f1 = x * (1 + 0.01 * rng.normal(size=x.size))
    c1 = x * (1 + 0.03 * rng.normal(size=x.size))
    f1[::501] = -1.0
    lvl = [(x[None], None), (f1[None], c1[None])]
    ldom = (0.05, 30.0)
    bl = onp.Basis(onp.LEGENDRE, R, ldom, log=True)
    ref = onp.estimate_mean(to_chunks(lvl), lambda v: onp.covariance_rows(bl, v))
    n, n_rm, s, sp = both(Legendre(R, ldom, log=True), lvl)
    _check_against(n, n_rm, s, sp, ref)


def test_unclipped_moments_keep_the_direct_covariance(hip, monkeypatch):
    """safe_eval=False: values outside the domain are not masked, the high extended terms of a linearisation would overflow for
    them (and 0 * inf would poison the entries that do not need them) -- such bases keep all three Gram matrices on the matrix
    cores, whatever the chunk size, and the mean-only route of estimate_mean is the direct one too."""
    from mlmc_amd import Legendre, linearize
    from mlmc_amd.engine import LevelAccumulator
    dom = (-3.7190164854556804, 3.7190164854556804)
    levels = level_arrays([4001, 2500], [0.5, 0.07], 1, 0)
    levels[0][0][0, 17] = 40.0                      # far outside: P_k(t) ~ 10^(1.3 k) there
    out = []
    for lin in ("1", "0"):
        monkeypatch.setenv("MLMC_HIP_LINEARIZE", lin)
        monkeypatch.setenv("MLMC_HIP_LINEARIZE_MIN_N", "0")
        fn = Legendre(24, dom, safe_eval=False)
        assert linearize.extended_size(fn) is None
        out.append(_run_accum(fn, levels, mode=LevelAccumulator.COV))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b, equal_nan=True)
    s = out[0][2].reshape(2, 24, 24)
    assert np.all(np.isfinite(s[0, :4, :4]))          # the low products of the outlier are ordinary numbers
    assert linearize.extended_size(Legendre(24, dom)) == 47
