"""Samples on the edges of a log domain (Moments(log=True, safe_eval=True), reference moments.py:27-39,58-73).

The reference keeps a sample when t = (np.log(x) - shift) * scale + ref0 lies in [ref0, ref1]; on the edge that hinges on
the last bit of NumPy's log, which the device log() does not share.  The product therefore decides on the RAW value
against thresholds the host bisects with NumPy's own log (mlmc_amd.moments.log_keep_interval -> mlmc_basis_desc.x_lo / x_hi).
Fixture: tests/golden/G10_log_edges.npz (oracle/gen_golden.py::g10_log_edges, the reference's own classes).

CPU tests: the oracle and the host bisection against the fixture.  GPU tests: eval_all NaN patterns and the sample counts of
an estimate whose samples sit on those edges -- bit-exact."""
import os

import numpy as np
import pytest

from oracle import oracle_np as onp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = "abcd"


@pytest.fixture(scope="module")
def g10():
    return np.load(os.path.join(GOLDEN, "G10_log_edges.npz"))


def _same_host_log(g10, tag):
    """NumPy's log is not the same function on every host (AVX-512 / AVX2 / scalar loops differ in the last bit for about
    one argument in a thousand).  The fixture records np.log(grid) of the container that generated it; where this host
    disagrees the fixture's masks are not this host's reference answers."""
    with np.errstate(all="ignore"):
        return np.array_equal(np.log(g10[tag + "_grid"]), g10[tag + "_log"], equal_nan=True)


@pytest.mark.parametrize("tag", CASES)
def test_oracle_and_host_thresholds_against_fixture(g10, tag):
    from mlmc_amd.moments import log_keep_interval
    if not _same_host_log(g10, tag):
        pytest.skip("this host's np.log differs from the fixture's on the edge grid")
    ldom, ref, grid = tuple(g10[tag + "_ldom"]), tuple(g10[tag + "_ref"]), g10[tag + "_grid"]
    b = onp.Basis(onp.LEGENDRE, 5, ldom, ref, log=True)
    assert np.array_equal(onp.eval_all(b, grid), g10[tag + "_legendre5"], equal_nan=True)
    assert np.array_equal(onp.eval_all(onp.Basis(onp.MONOMIAL, 3, ldom, ref if tag == "b" else None, log=True), grid),
                          g10[tag + "_monomial3"], equal_nan=True)
    x_lo, x_hi = log_keep_interval(b.shift, b.scale, b.ref_domain[0], b.ref_domain[1])
    assert np.array_equal(np.array([x_lo, x_hi]), g10[tag + "_keep_interval"])
    kept_ref = ~np.isnan(g10[tag + "_legendre5"][:, 0])
    assert np.array_equal((grid >= x_lo) & (grid <= x_hi), kept_ref)


@pytest.mark.parametrize("tag", CASES)
def test_threshold_rule_equals_direct_evaluation_near_the_edges(g10, tag):
    """x -> t is monotone: 257 consecutive doubles around each threshold, decided by the thresholds and by the reference's
    arithmetic (this host's np.log -- no fixture involved), agree."""
    from mlmc_amd.moments import log_keep_interval
    ldom, ref = tuple(g10[tag + "_ldom"]), tuple(g10[tag + "_ref"])
    b = onp.Basis(onp.LEGENDRE, 2, ldom, ref, log=True)
    x_lo, x_hi = log_keep_interval(b.shift, b.scale, b.ref_domain[0], b.ref_domain[1])
    for centre in (x_lo, x_hi):
        bits = np.array([centre]).view(np.int64)[0] + np.arange(-128, 129)
        x = bits[bits > 0].astype(np.int64).view(np.float64)
        direct = ~np.isnan(onp.transform(b, x))
        assert np.array_equal((x >= x_lo) & (x <= x_hi), direct)


def test_empty_keep_interval():
    from mlmc_amd.moments import log_keep_interval
    # ref1 < ref0: nothing can be kept
    assert log_keep_interval(0.0, 1.0, 1.0, -1.0) == (float("inf"), 0.0)


def _hip():
    from mlmc_amd import _lib
    _lib.init(0)
    return _lib


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_device_masks_and_counts_on_log_edges(g10, tag):
    from mlmc_amd import Legendre, Monomial
    from mlmc_amd.engine import LevelAccumulator, level_stats
    _hip()
    ldom, ref, grid = tuple(g10[tag + "_ldom"]), tuple(g10[tag + "_ref"]), g10[tag + "_grid"]
    fn = Legendre(5, ldom, ref_domain=ref, log=True)
    # the live oracle on this host is the reference answer here; the fixture is too wherever this host's np.log agrees
    b = onp.Basis(onp.LEGENDRE, 5, ldom, ref, log=True)
    want = [onp.eval_all(b, grid)]
    if _same_host_log(g10, tag):
        want.append(g10[tag + "_legendre5"])
    got = fn.eval_all(grid)
    for w in want:
        assert np.array_equal(np.isnan(got), np.isnan(w)), tag
        m = ~np.isnan(w)
        assert np.all(np.abs(got[m] - w[m]) <= 1e-10 * np.maximum(1.0, np.abs(w[m])))
    gm = Monomial(3, ldom, ref_domain=ref if tag == "b" else None, log=True).eval_all(grid)
    wm = onp.eval_all(onp.Basis(onp.MONOMIAL, 3, ldom, ref if tag == "b" else None, log=True), grid)
    assert np.array_equal(np.isnan(gm), np.isnan(wm))
    # safe_eval=False: every positive finite value is kept
    gn = Legendre(5, ldom, ref_domain=ref, log=True, safe_eval=False).eval_all(grid)
    assert np.array_equal(np.isnan(gn[:, 0]), np.isnan(g10[tag + "_legendre5_nosafe"][:, 0]))
    # the estimate: two levels of samples drawn from the edge grid -- counts bit-exact, sums 1e-10
    f0, f1, c1 = g10[tag + "_est_fine0"], g10[tag + "_est_fine1"], g10[tag + "_est_coarse1"]
    for R, mode in ((5, LevelAccumulator.MOMENTS), (5, LevelAccumulator.COV), (40, LevelAccumulator.MOMENTS), (60, LevelAccumulator.MOMENTS)):
        fnR = Legendre(R, ldom, ref_domain=ref, log=True)
        acc = LevelAccumulator(fnR, 2, mode)
        acc.push(0, f0, None)
        acc.push(1, f1, c1)
        n, n_rm, s, sp = acc.finalize()
        acc.close()
        chunks = [[f0[None, :, None]], [np.stack([f1, c1], axis=-1)[None]]]
        bR = onp.Basis(onp.LEGENDRE, R, ldom, ref, log=True)
        rows = onp.moments_rows if mode == LevelAccumulator.MOMENTS else onp.covariance_rows
        live = onp.estimate_mean(chunks, lambda x: rows(bR, x))
        assert np.array_equal(n, live.n_samples) and np.array_equal(n_rm, live.n_rm_samples), (tag, R, n, live.n_samples)
        if _same_host_log(g10, tag):
            assert np.array_equal(n, g10[tag + "_est_n"]) and np.array_equal(n_rm, g10[tag + "_est_n_rm"])
        l_means, l_vars = level_stats(n, s, sp)
        rms = np.sqrt(np.abs(live.sums_sq) / np.maximum(live.n_samples[:, None], 1))
        assert np.all(np.abs(l_means - live.l_means) <= 1e-10 * np.maximum(np.abs(live.l_means), rms) + 1e-300)
        if R == 5 and mode == LevelAccumulator.MOMENTS and _same_host_log(g10, tag):
            assert np.all(np.abs(l_means - g10[tag + "_est_l_means"]) <= 1e-10 * np.maximum(np.abs(g10[tag + "_est_l_means"]), rms) + 1e-300)


@pytest.mark.gpu
def test_library_bisects_with_libm_when_no_thresholds_are_given():
    """mlmc_basis_desc.x_lo == x_hi == 0 under is_log && is_clip: the library finds the keep interval with the host C
    library's log.  glibc's log and NumPy's agree except in the last bit of about one argument in a thousand, so the interval
    must lie within a few hundred ulps of NumPy's and masks of interior / exterior points must agree."""
    import ctypes as C
    lib = _hip()
    from mlmc_amd import Legendre
    fn = Legendre(3, (0.05, 30.0), log=True)
    d = fn._desc()
    d.x_lo, d.x_hi = 0.0, 0.0
    h = C.c_void_p()
    lib.check(lib.lib().mlmc_basis_create(C.byref(d), C.byref(h)))
    x = np.array([0.04, 0.049999999, 0.0500001, 1.0, 29.99999, 30.00001, 31.0, -1.0, 0.0, np.inf, np.nan])
    out = np.empty((x.size, 3))
    lib.check(lib.lib().mlmc_basis_eval(h, lib.ptr(x), x.size, 3, lib.ptr(out), lib.HOST))
    lib.lib().mlmc_basis_destroy(h)
    assert np.array_equal(~np.isnan(out[:, 0]), np.array([0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0], dtype=bool))
