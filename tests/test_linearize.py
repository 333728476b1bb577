"""Product linearisation phi_i phi_j = sum_k c_ijk phi_k (mlmc_amd/linearize.py): the mean of the moment covariance
(reference quantity_estimate.py:131-147 + :59-65) from the level sums of ~2 R moments.

CPU: the coefficient tensors against direct products of the oracle's basis values, and the covariance level means of the
oracle (reference-form per-sample outer products) against the contraction of the oracle's moment sums.
GPU: estimate_mean(covariance(q, fn), variance=False) through the linearised pass against the matrix-core pass
(MLMC_HIP_LINEARIZE=0) and the oracle; the mean-only term-split kernel (65..128 terms in one pass) against the
two-pass kernels; Estimate.construct_density with one device pass."""
import os

import numpy as np
import pytest

from oracle import oracle_np as onp
from tests.util import level_arrays, to_chunks

DOM = (-3.7190164854556804, 3.7190164854556804)


def _product_check(kind, R, K, C, domain, n=500):
    ref = {onp.LEGENDRE: (-1.0, 1.0), onp.MONOMIAL: (0.0, 1.0), onp.FOURIER: (0.0, 2 * np.pi)}[kind]
    x = np.random.default_rng(R).uniform(domain[0], domain[1], n)
    phi = onp.eval_all(onp.Basis(kind, K, domain), x)                     # [n, K]
    lhs = (phi[:, :R, None] * phi[:, None, :R]).reshape(n, R * R)
    rhs = phi @ C.T
    scale = np.maximum(1.0, np.abs(lhs))
    return float(np.max(np.abs(lhs - rhs) / scale)), ref


@pytest.mark.parametrize("R", [1, 2, 5, 17, 64, 100])
def test_legendre_product_coefficients(R):
    from mlmc_amd import linearize
    C = linearize.legendre_products(R)
    assert C.shape == (R * R, 2 * R - 1)
    assert C.min() >= 0.0                                           # Adams: non-negative ...
    assert np.max(np.abs(C.sum(axis=1) - 1.0)) < 4e-15             # ... and P_i(1) P_j(1) = 1 = sum_k c_ijk
    C3 = C.reshape(R, R, -1)
    assert np.array_equal(C3, C3.transpose(1, 0, 2))
    err, _ = _product_check(onp.LEGENDRE, R, 2 * R - 1, C, DOM)
    assert err < 1e-12, err
    if R >= 5:                                                      # exact rationals for a few entries
        from fractions import Fraction

        def A(n):
            v = Fraction(1)
            for m in range(1, n + 1):
                v *= Fraction(2 * m - 1, m)
            return v
        for i, j, k in ((R - 1, R - 1, 2 * R - 2), (R - 1, R - 1, 0), (R - 2, 3, R - 3), (4, 4, 2)):
            s = (i + j + k) // 2
            want = float(Fraction(2 * k + 1, 2 * s + 1) * A(s - i) * A(s - j) * A(s - k) / A(s)) if (i + j + k) % 2 == 0 else 0.0
            assert abs(C3[i, j, k] - want) <= 2e-16 * max(want, 1e-300) + 1e-300, (i, j, k)


@pytest.mark.parametrize("R", [1, 2, 6, 7, 33])
def test_monomial_and_fourier_product_coefficients(R):
    from mlmc_amd import linearize
    C = linearize.monomial_products(R)
    err, _ = _product_check(onp.MONOMIAL, R, 2 * R - 1, C, (0.25, 2.0))
    assert err < 1e-13
    C = linearize.fourier_products(R)
    K = 4 * (R // 2) + 1
    assert C.shape == (R * R, K)
    err, _ = _product_check(onp.FOURIER, R, K, C, (0.0, 2 * np.pi))
    assert err < 1e-12, err


def test_covariance_level_means_from_moment_sums_match_the_reference_form():
    """The oracle's reference-form covariance (per-sample outer products) against the contraction of the oracle's sums of
    the 2 R - 1 moments: counts identical, level means within 1e-10 of the parity gate."""
    from mlmc_amd import linearize
    steps = [0.5, 0.07, 0.01]
    levels = level_arrays([3000, 2000, 1200], steps, 1, 7)
    chunks = to_chunks(levels)
    for kind, prod, R in ((onp.LEGENDRE, linearize.legendre_products, 24), (onp.MONOMIAL, linearize.monomial_products, 6)):
        b = onp.Basis(kind, R, DOM)
        bx = onp.Basis(kind, 2 * R - 1, DOM)
        ref = onp.estimate_mean(chunks, lambda x: onp.covariance_rows(b, x))
        mom = onp.estimate_mean(chunks, lambda x: onp.moments_rows(bx, x))
        assert np.array_equal(ref.n_samples, mom.n_samples) and np.array_equal(ref.n_rm_samples, mom.n_rm_samples)
        got = (mom.sums @ prod(R).T) / ref.n_samples[:, None]
        rms = np.sqrt(np.abs(ref.sums_sq) / ref.n_samples[:, None])
        assert np.all(np.abs(got - ref.l_means) <= 1e-10 * np.maximum(np.abs(ref.l_means), rms) + 1e-300)


# ---------------------------------------------------------------------------------------------------------------------
def _hip():
    from mlmc_amd import _lib
    _lib.init(0)
    return _lib


@pytest.mark.gpu
@pytest.mark.parametrize("R", [65, 80, 96, 97, 127, 128, 129, 199, 255, 256])
def test_mean_only_split_kernel_65_to_128_terms(R):
    """MOMENTS, mean only, 64 < R <= 256: one pass of k_moments_accum_split<..., SQ = false> (sp = NaN) per window of 128
    terms (the second window's head first walks 128 recurrence steps without accumulating) against the mean + variance passes
    of the same library and the oracle; ragged sizes, masks, level 0 only, monomials."""
    from mlmc_amd import Legendre, Monomial
    from mlmc_amd.engine import LevelAccumulator
    _hip()
    steps = [0.5, 0.07, 0.01]
    for N in ([20011, 9001, 130], [1], [257, 3]):
        levels = level_arrays(N, steps[:len(N)], 1, 13 if N[0] > 100 else 0)
        for cls, kind in ((Legendre, onp.LEGENDRE), (Monomial, onp.MONOMIAL)):
            fn = cls(R, DOM)
            out = []
            for mean_only in (True, False):
                acc = LevelAccumulator(fn, len(N), LevelAccumulator.MOMENTS, mean_only=mean_only)
                for l, (f, c) in enumerate(levels):
                    acc.push(l, f[0], None if c is None else c[0])
                out.append(acc.finalize())
                acc.close()
            (n1, r1, s1, sp1), (n2, r2, s2, sp2) = out
            assert np.array_equal(n1, n2) and np.array_equal(r1, r2)
            assert np.all(np.isnan(sp1)) and not np.any(np.isnan(sp2))
            scale = np.sqrt(np.abs(sp2) * np.maximum(n2[:, None], 1)) + 1e-300
            assert np.all(np.abs(s1 - s2) <= 1e-12 * np.maximum(np.abs(s2), scale)), (R, N, cls.__name__)
            if cls is Legendre and N[0] > 100 and R in (65, 127, 199):
                # two components sharing one mask (k_mask + the general, non-PLAIN loop of the split kernel)
                lv2 = level_arrays(N, steps[:len(N)], 2, 13)
                pair = []
                for mean_only in (True, False):
                    acc = LevelAccumulator(fn, len(N), LevelAccumulator.MOMENTS, n_comp=2, mean_only=mean_only)
                    for l, (f, c) in enumerate(lv2):
                        acc.push(l, f, c)
                    pair.append(acc.finalize())
                    acc.close()
                assert np.array_equal(pair[0][0], pair[1][0]) and np.array_equal(pair[0][1], pair[1][1]) and np.all(np.isnan(pair[0][3]))
                sc = np.sqrt(np.abs(pair[1][3]) * np.maximum(pair[1][0][:, None], 1)) + 1e-300
                assert np.all(np.abs(pair[0][2] - pair[1][2]) <= 1e-12 * np.maximum(np.abs(pair[1][2]), sc))
            if cls is Legendre and N[0] > 100:
                ref = onp.estimate_mean(to_chunks(levels), lambda x: onp.moments_rows(onp.Basis(kind, R, DOM), x))
                assert np.array_equal(n1, ref.n_samples) and np.array_equal(r1, ref.n_rm_samples)
                sc = np.maximum(np.abs(ref.sums), np.sqrt(np.abs(ref.sums_sq) * ref.n_samples[:, None]))
                assert np.all(np.abs(s1 - ref.sums) <= 1e-10 * sc + 1e-300)


def _storage(levels, steps, chunk_size=None, M=1):
    from mlmc_amd.quantity.quantity_spec import QuantitySpec
    from mlmc_amd.sample_storage import Memory
    spec = [QuantitySpec(name="q", unit="m", shape=(M, 1), times=[1], locations=['0'])]
    st = Memory(chunk_size=chunk_size) if chunk_size else Memory()
    st.save_global_data(result_format=spec, level_parameters=[[s] for s in steps])
    for l, (f, c) in enumerate(levels):
        st.set_level_samples(l, f.T, None if c is None else c.T)
    return st, spec


@pytest.mark.gpu
@pytest.mark.parametrize("family,R", [("Legendre", 8), ("Legendre", 33), ("Legendre", 64), ("Legendre", 100), ("Monomial", 6),
                                        ("Fourier", 9), ("Fourier", 10)])
def test_linearised_covariance_mean_equals_the_matrix_core_pass(family, R, monkeypatch):
    """estimate_mean(covariance(q, fn), variance=False): the linearised pass (moments of ~2 R terms + host contraction)
    against the matrix-core pass of the same library (MLMC_HIP_LINEARIZE=0), counts identical; Legendre also against the oracle's
    reference-form covariance."""
    import mlmc_amd
    from mlmc_amd.quantity import quantity_estimate as qe
    from mlmc_amd.quantity.quantity import make_root_quantity
    _hip()
    steps = [0.5, 0.07, 0.01]
    levels = level_arrays([6001, 3000, 1100], steps, 1, 11)
    st, spec = _storage(levels, steps, chunk_size=2000)
    q = make_root_quantity(st, spec)['q'][1]['0'][0, 0]
    fn = getattr(mlmc_amd, family)(R, DOM if family != "Fourier" else (-3.7, 3.7))
    lin = qe.estimate_mean(qe.covariance(q, fn), variance=False)
    assert "_lin_memo" in fn.__dict__ and fn.__dict__["_lin_ext"].size == (4 * (R // 2) + 1 if family == "Fourier" else 2 * R - 1)
    monkeypatch.setenv("MLMC_HIP_LINEARIZE", "0")
    mfma = qe.estimate_mean(qe.covariance(q, fn), variance=False)
    full = qe.estimate_mean(qe.covariance(q, fn))
    monkeypatch.delenv("MLMC_HIP_LINEARIZE")
    for r in (mfma, full):
        assert np.array_equal(lin.n_samples, r.n_samples) and np.array_equal(lin.n_rm_samples, r.n_rm_samples)
    assert np.all(np.isnan(lin.l_vars)) and lin.mean.shape == (R, R)
    rms = np.sqrt(np.maximum(full.l_vars, 0) + full.l_means ** 2)
    tol = 1e-10 * np.maximum(np.abs(full.l_means), rms) + 1e-300
    assert np.all(np.abs(lin.l_means - full.l_means) <= tol)
    assert np.all(np.abs(mfma.l_means - full.l_means) <= tol)
    if family == "Legendre" and R <= 33:
        b = onp.Basis(onp.LEGENDRE, R, DOM)
        chunks = to_chunks(level_arrays([1500, 1200, 900], steps, 1, 11))
        levels2 = level_arrays([1500, 1200, 900], steps, 1, 11)
        st2, _ = _storage(levels2, steps)
        q2 = make_root_quantity(st2, spec)['q'][1]['0'][0, 0]
        ref = onp.estimate_mean(chunks, lambda x: onp.covariance_rows(b, x))
        got = qe.estimate_mean(qe.covariance(q2, fn), variance=False)
        assert np.array_equal(got.n_samples, ref.n_samples) and np.array_equal(got.n_rm_samples, ref.n_rm_samples)
        rms2 = np.sqrt(np.abs(ref.sums_sq) / ref.n_samples[:, None])
        assert np.all(np.abs(got.l_means.reshape(ref.l_means.shape) - ref.l_means) <= 1e-10 * np.maximum(np.abs(ref.l_means), rms2) + 1e-300)


@pytest.mark.gpu
def test_construct_density_is_one_device_pass_and_agrees_with_the_two_pass_chain(monkeypatch):
    """Estimate.construct_density (reference estimator.py:304-331): with Legendre moments the covariance mean comes from the
    linearised pass and the orthogonal-moments means from its kept sums (no second pass); same multipliers and density as
    the chain with the matrix-core covariance pass and a device pass over the transformed moments."""
    from mlmc_amd import Legendre
    from mlmc_amd.estimator import Estimate
    from mlmc_amd.quantity import quantity_estimate as qe
    from mlmc_amd.quantity.quantity import make_root_quantity
    _hip()
    steps = [0.5, 0.07, 0.01]
    levels = level_arrays([60001, 20000, 5000], steps, 1, 0)
    st, spec = _storage(levels, steps)
    q = make_root_quantity(st, spec)['q'][1]['0'][0, 0]
    fn = Legendre(21, DOM)
    est = Estimate(q, st, fn)
    qe.device_cache_clear()
    u0 = qe._device_cache.uploads
    d1, info1, res1, mo1 = est.construct_density(tol=1e-8)
    assert "_lin_memo" in fn.__dict__
    monkeypatch.setenv("MLMC_HIP_LINEARIZE", "0")
    fn2 = Legendre(21, DOM)
    d2, info2, res2, mo2 = Estimate(q, st, fn2).construct_density(tol=1e-8)
    monkeypatch.delenv("MLMC_HIP_LINEARIZE")
    assert "_lin_memo" not in fn2.__dict__
    assert res1.success and res2.success and mo1.size == mo2.size
    assert np.allclose(info1[0], info2[0], rtol=1e-9, atol=1e-13)            # eigenvalues of the covariance
    x = np.linspace(DOM[0], DOM[1], 501)
    assert np.max(np.abs(d1.density(x) - d2.density(x))) < 1e-7
    # a changed storage invalidates the kept sums
    f0 = levels[0][0].copy()
    f0[0, :100] += 0.5
    st.set_level_samples(0, f0.T, None)
    d3, _, res3, _ = est.construct_density(tol=1e-8)
    monkeypatch.setenv("MLMC_HIP_LINEARIZE", "0")
    d4, _, res4, _ = Estimate(q, st, Legendre(21, DOM)).construct_density(tol=1e-8)
    assert np.max(np.abs(d3.density(x) - d4.density(x))) < 1e-7 and np.max(np.abs(d3.density(x) - d1.density(x))) > 1e-9


@pytest.mark.gpu
def test_linearised_covariance_mean_of_a_vector_quantity(monkeypatch):
    """Four components, one mask for all (a sample is dropped when any component is masked), both row layouts
    (cov_at_bottom True / False): the linearised pass against the matrix-core pass, and 100 moments (a 199-term pass of the
    run-time-window kernels) for a two-component quantity."""
    from mlmc_amd import Legendre
    from mlmc_amd.quantity import quantity_estimate as qe
    from mlmc_amd.quantity.quantity import make_root_quantity
    _hip()
    steps = [0.5, 0.07, 0.01]
    levels = level_arrays([5001, 2500, 1200], steps, 4, 9)
    st, spec = _storage(levels, steps, chunk_size=1500, M=4)
    root = make_root_quantity(st, spec)['q'][1]['0']
    for R, q in ((9, root), (70, root[:2])):
        fn = Legendre(R, DOM)
        for at_bottom in (True, False):
            lin = qe.estimate_mean(qe.covariance(q, fn, cov_at_bottom=at_bottom), variance=False)
            monkeypatch.setenv("MLMC_HIP_LINEARIZE", "0")
            full = qe.estimate_mean(qe.covariance(q, fn, cov_at_bottom=at_bottom))
            monkeypatch.delenv("MLMC_HIP_LINEARIZE")
            assert np.array_equal(lin.n_samples, full.n_samples) and np.array_equal(lin.n_rm_samples, full.n_rm_samples)
            assert lin.l_means.shape == full.l_means.shape and lin.mean.shape == full.mean.shape
            rms = np.sqrt(np.maximum(full.l_vars, 0) + full.l_means ** 2)
            assert np.all(np.abs(lin.l_means - full.l_means) <= 1e-10 * np.maximum(np.abs(full.l_means), rms) + 1e-300), (R, at_bottom)


# ---------------------------------------------------------------------------------------------------------------------
# The tables the LIBRARY builds for its own linearisations (covariance with variances: mean from 2 R - 1 moments, level 0
# from 4 R - 3): host arithmetic behind the C ABI, checked here without a GPU.
def _lib_table(kind, R, squares):
    from mlmc_amd import _lib
    K = 4 * R - 3 if squares else 2 * R - 1
    out = np.empty(K * R * R)
    _lib.check(_lib.load().mlmc_linearization_table(kind, R, int(squares), _lib.ptr(out), out.size))
    return out.reshape(K, R, R)


def _exact_legendre_products(R):
    """c_ijk as Fractions by Adams' formula with exact factorial ratios (small R only)."""
    from fractions import Fraction
    from math import factorial

    def A(n):                                        # (2n - 1)!! / n! = (2n)! / (2^n n!^2)
        return Fraction(factorial(2 * n), 2 ** n * factorial(n) ** 2)
    c = {}
    for i in range(R):
        for j in range(R):
            for k in range(abs(i - j), i + j + 1, 2):
                s = (i + j + k) // 2
                c[i, j, k] = Fraction(2 * k + 1, 2 * s + 1) * A(s - i) * A(s - j) * A(s - k) / A(s)
    return c


@pytest.mark.parametrize("R", [1, 2, 5, 9])
def test_library_tables_against_exact_rationals(R):
    from fractions import Fraction
    from mlmc_amd import _lib
    c = _exact_legendre_products(2 * R)              # products of products need the table of the doubled size
    t1 = _lib_table(_lib.LEGENDRE, R, False)
    t2 = _lib_table(_lib.LEGENDRE, R, True)
    for i in range(R):
        for j in range(R):
            row = {k: c[i, j, k] for k in range(abs(i - j), i + j + 1, 2)}
            for k in range(2 * R - 1):
                assert t1[k, i, j] == float(row.get(k, Fraction(0))), (i, j, k)
            sq = {}
            for a, ca in row.items():
                for b, cb in row.items():
                    for k in range(abs(a - b), a + b + 1, 2):
                        sq[k] = sq.get(k, Fraction(0)) + ca * cb * c[a, b, k]
            for k in range(4 * R - 3):
                exact = float(sq.get(k, Fraction(0)))
                assert abs(t2[k, i, j] - exact) <= 4e-16 * max(exact, 1e-3), (i, j, k, t2[k, i, j], exact)
                if k not in sq:
                    assert t2[k, i, j] == 0.0


@pytest.mark.parametrize("R", [17, 64])
def test_library_tables_properties_and_python_twin(R):
    """Full sizes: the product table is bit for bit the one of mlmc_amd/linearize.py (same recurrences in the same extended
    precision); both tables are non-negative, symmetric in (i, j), their rows sum to one (phi_k(1) = 1), and the squares' table
    has zeros at odd k and beyond 2 (i + j)."""
    from mlmc_amd import linearize
    t1 = _lib_table(0, R, False)
    py = linearize.legendre_products(R).reshape(R, R, 2 * R - 1).transpose(2, 0, 1)
    assert np.array_equal(t1, py)
    t2 = _lib_table(0, R, True)
    for t in (t1, t2):
        assert np.all(t >= 0) and np.array_equal(t, t.transpose(0, 2, 1))
        assert np.max(np.abs(t.sum(axis=0) - 1.0)) < 5e-15
    assert not t2[1::2].any()
    # (P_0 P_j)^2 = P_j^2: rows with i = 0 are the product table's, exactly -- c2_00k = delta_k0 carries the exact counts
    assert t2[0, 0, 0] == 1.0 and not t2[1:, 0, 0].any()
    for j in range(R):
        assert np.array_equal(t2[:2 * R - 1, 0, j], t1[:, j, j]) and not t2[2 * R - 1:, 0, j].any()
    I, J = np.meshgrid(np.arange(R), np.arange(R), indexing="ij")
    for k in range(4 * R - 3):
        assert not t2[k][2 * (I + J) < k].any()
    # squares' table = the product table applied twice (in double: agreement to rounding)
    big = linearize.legendre_products(2 * R - 1).reshape(2 * R - 1, 2 * R - 1, 4 * R - 3)
    i, j = R - 1, R // 2
    v = t1[:, i, j]
    ref = np.einsum("a,b,abk->k", v, v, big)
    assert np.max(np.abs(t2[:, i, j] - ref)) < 1e-14
    # monomials: exponents add
    m1, m2 = _lib_table(1, 5, False), _lib_table(1, 5, True)
    assert m1.sum() == 25 and m2.sum() == 25 and m1[3 + 4, 3, 4] == 1.0 and m2[2 * (3 + 4), 3, 4] == 1.0


def test_library_table_argument_checks():
    from mlmc_amd import _lib
    lib = _lib.load()
    out = np.empty(16)
    for args in ((2, 4, 0, _lib.ptr(out), 16),          # Fourier: no table here
                 (0, 0, 0, _lib.ptr(out), 16),          # size out of range
                 (0, 129, 0, _lib.ptr(out), 16),
                 (0, 65, 1, _lib.ptr(out), 16),         # squares: at most 64 moments
                 (0, 3, 0, _lib.ptr(out), 16),          # needs 5 * 9 = 45 doubles
                 (0, 2, 0, None, 16)):
        assert lib.mlmc_linearization_table(*args) != 0
        assert lib.mlmc_last_error().decode() != ""
    assert lib.mlmc_linearization_table(0, 2, 0, _lib.ptr(out), 16) == 0       # 3 * 4 = 12 doubles fit
    assert np.array_equal(out[:12].reshape(3, 2, 2)[:, 1, 1], [1.0 / 3.0, 0.0, 2.0 / 3.0])   # P_1^2 = 1/3 P_0 + 2/3 P_2


def test_no_linearisation_where_extended_terms_could_overflow():
    """Unclipped bases and monomials on a reference domain beyond [-1, 1]: the high extended terms are not bounded by one there."""
    from mlmc_amd import Legendre, Monomial, linearize
    assert linearize.extended_size(Legendre(10, (-2.0, 2.0))) == 19
    assert linearize.extended_size(Legendre(10, (-2.0, 2.0), safe_eval=False)) is None
    assert linearize.extended_size(Monomial(10, (0.0, 5.0))) == 19
    assert linearize.extended_size(Monomial(10, (0.0, 5.0), ref_domain=(0, 3))) is None
    assert linearize.extended_size(Monomial(10, (0.0, 5.0), safe_eval=False)) is None
