"""Maximum-entropy PDF from generalised moments (reference interface: mlmc/tool/simple_distribution.py).

`SimpleDistribution`, `construct_ortogonal_moments` and the diagnostic integrals keep the reference's names,
arguments and result fields.  The optimisation itself (functional, gradient, Hessian on a quadrature, Newton
steps with a Cholesky solve) runs on the MI355X through `mlmc_maxent_solve`; `density` and `cdf` are device
kernels as well.  The reference's SciPy trust-ncg iterates and QUADPACK sub-interval layout are not reproduced
(they are pinned by no reference test); the converged multipliers are, because the functional is
strictly convex.
"""
import ctypes as C

import numpy as np
import scipy.integrate as integrate
import scipy.linalg


class _SmallLapack:
    """Context for the dense LAPACK calls of this module (eigh, rq, eigvalsh on matrices of at most 128 x 128): one BLAS
    thread.  On a many-core host the threaded BLAS spreads such a call over every core it sees -- on the MI355X boxes of this
    pool (256 logical cores, a 16-core share per GPU) a 128 x 128 `eigh` that takes 3 ms on one thread took 85-90 ms, thirty
    times the GPU part of construct_density.  threadpoolctl is optional: without it the calls run as NumPy configures them."""
    _controller = None

    def __enter__(self):
        self._ctx = None
        try:
            if _SmallLapack._controller is None:
                from threadpoolctl import ThreadpoolController
                _SmallLapack._controller = ThreadpoolController()
            self._ctx = _SmallLapack._controller.limit(limits=1, user_api="blas")
            self._ctx.__enter__()
        except Exception:
            self._ctx = None
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False

from scipy.optimize import OptimizeResult

from .. import _lib
from .. import moments as moments_mod

EXACT_QUAD_LIMIT = 1000


def _solve_on_device(moments_fn, means, errs, domain, multipliers, tol, max_it, n_intervals=0, gauss_degree=21,
                     stab_penalty=0.0, penalty_coef=0.0, decay=(False, False), prev=None):
    size = len(multipliers)
    opts = _lib.MaxentOpts()
    opts.tol = float(tol)
    opts.max_it = int(max_it)
    opts.n_intervals = int(n_intervals)
    opts.gauss_degree = int(gauss_degree)
    opts.stab_penalty = float(stab_penalty)
    opts.penalty_coef = float(penalty_coef)
    opts.decay_left, opts.decay_right = int(bool(decay[0])), int(bool(decay[1]))
    info = _lib.MaxentInfo()
    lam = np.ascontiguousarray(multipliers, dtype=np.float64).copy()
    mu = np.ascontiguousarray(means[:size], dtype=np.float64)
    sig = np.ascontiguousarray(errs[:size], dtype=np.float64)
    hess = np.empty((size, size), dtype=np.float64)
    grad = np.empty(size, dtype=np.float64)
    prev_arr = None if prev is None else np.ascontiguousarray(prev, dtype=np.float64)
    _lib.check(_lib.lib().mlmc_maxent_solve(moments_fn._basis_handle(), _lib.ptr(mu), _lib.ptr(sig), size, float(domain[0]),
                                            float(domain[1]), C.byref(opts), _lib.ptr(prev_arr),
                                            0 if prev_arr is None else len(prev_arr), _lib.ptr(lam), _lib.ptr(grad), _lib.ptr(hess), C.byref(info)))
    return lam, grad, hess, info


def _device_density(moments_fn, multipliers, errs, value):
    value = np.atleast_1d(np.asarray(value, dtype=np.float64))
    flat = np.ascontiguousarray(value.reshape(-1))
    out = np.empty_like(flat)
    lam = np.ascontiguousarray(multipliers, dtype=np.float64)
    sig = np.ascontiguousarray(errs[:len(lam)], dtype=np.float64)
    _lib.check(_lib.lib().mlmc_density_eval(moments_fn._basis_handle(), _lib.ptr(lam), _lib.ptr(sig), len(lam), _lib.ptr(flat),
                                            flat.size, _lib.ptr(out), _lib.HOST))
    return out.reshape(value.shape)


def _device_integrals(moments_fn, multipliers, errs, lo, hi, degree):
    lo = np.ascontiguousarray(lo, dtype=np.float64)
    hi = np.ascontiguousarray(hi, dtype=np.float64)
    out = np.empty_like(lo)
    lam = np.ascontiguousarray(multipliers, dtype=np.float64)
    sig = np.ascontiguousarray(errs[:len(lam)], dtype=np.float64)
    _lib.check(_lib.lib().mlmc_density_integrate(moments_fn._basis_handle(), _lib.ptr(lam), _lib.ptr(sig), len(lam), _lib.ptr(lo),
                                                 _lib.ptr(hi), lo.size, int(degree), _lib.ptr(out)))
    return out


def _cdf(dist, values):
    """Cumulative `n`-point Gauss-Legendre between successive values, as the reference does (:108-125) -- the
    intervals are independent, so they are integrated in one device launch and prefix-summed."""
    values = np.atleast_1d(values)
    lo_edges, hi_edges, idx = [], [], []
    last_x = dist.domain[0]
    for i, val in enumerate(values):
        if dist.domain[0] < val < dist.domain[1]:
            lo_edges.append(last_x)
            hi_edges.append(val)
            idx.append(i)
            last_x = val
    pieces = _device_integrals(dist.moments_fn, dist.multipliers, dist._moment_errs, lo_edges, hi_edges, 10) if idx else []
    cdf_y = np.empty(len(values))
    last_y, k = 0, 0
    for i, val in enumerate(values):
        if val <= dist.domain[0]:
            last_y = 0
        elif val >= dist.domain[1]:
            last_y = 1
        else:
            last_y = last_y + pieces[k]
            k += 1
        cdf_y[i] = last_y
    return cdf_y


class SimpleDistribution:
    """Maximum-entropy density for given moment means (reference: simple_distribution.py:9-327)."""

    def __init__(self, moments_obj, moment_data, domain=None, force_decay=(True, True), verbose=False):
        self.moments_fn = None
        if domain is None:
            domain = moments_obj.domain
        self.domain = domain
        self.decay_penalty = force_decay
        self._verbose = verbose
        if moment_data is not None:
            self.moment_means = moment_data[:, 0]
            self.moment_errs = np.sqrt(moment_data[:, 1])
        self.multipliers = None
        self.approx_size = len(self.moment_means)
        assert moments_obj.size >= self.approx_size
        self.moments_fn = moments_obj
        self._gauss_degree = 21
        self._penalty_coef = 0
        # composite Gauss-Legendre sub-intervals of the fixed device quadrature (the reference takes them from
        # QUADPACK's adaptive bisection; 64 x 21 points integrate the Legendre-61 cases of the reference tests to 1e-13)
        self.n_intervals = 64

    def estimate_density_minimize(self, tol=1e-5, reg_param=0.01):
        """:param tol: tolerance for the norm of the gradient (moment residuals divided by their std errors)
        :param reg_param: unused, as in the reference (:50)
        :return: OptimizeResult with x, success, nit, fun, jac, fun_norm, eigvals, solver_res"""
        self._initialize_params(self.approx_size, tol)
        lam, grad, hess, info = _solve_on_device(self.moments_fn, self.moment_means, self._moment_errs, self.domain, self.multipliers,
                                           tol, max_it=100, n_intervals=self.n_intervals, gauss_degree=self._gauss_degree)
        result = OptimizeResult()
        result.x = lam.copy()
        result.fun = info.fun
        result.nit = info.nit
        result.hess = hess
        result.success = bool(info.success)
        result.status = 0 if info.success else 1
        result.message = "Optimization terminated successfully." if info.success else "Maximum number of iterations has been exceeded."
        self.multipliers = lam
        jac_norm = info.grad_norm
        result.jac = grad
        if self._verbose:
            print("size: {} nits: {} tol: {:5.3g} res: {:5.3g} msg: {}".format(self.approx_size, result.nit, tol, jac_norm, result.message))
        with _SmallLapack():
            result.eigvals = np.linalg.eigvalsh(hess)
        result.solver_res = result.jac
        # normalisation fix exactly as the reference applies it (:81-86)
        moment_0 = info.moment0
        self.multipliers[0] -= np.log(moment_0)
        if result.success or jac_norm < tol:
            result.success = True
        result.nit = max(result.nit, 1)
        result.fun_norm = jac_norm
        return result

    def density(self, value):
        """exp(clip(-sum_i phi_i(x) lambda_i / sigma_i, +-200)) (reference: :96-105)"""
        return _device_density(self.moments_fn, self.multipliers, self._moment_errs, value)

    def cdf(self, values):
        return _cdf(self, values)

    def _initialize_params(self, size, tol=None):
        assert self.domain is not None
        assert tol is not None
        self._quad_tolerance = 1e-10
        self._moment_errs = self.moment_errs
        self.multipliers = np.zeros(size)
        self.multipliers[0] = -np.log(1 / (self.domain[1] - self.domain[0]))   # uniform density to start with
        self._quad_log = []

    def eval_moments(self, x):
        return self.moments_fn.eval_all(x, self.approx_size)

    def end_point_derivatives(self):
        """One-sided difference quotients of the moments at the domain end points (reference: :240-252)."""
        eps = 1e-10
        left = right = np.zeros((1, self.approx_size))
        if self.decay_penalty[0]:
            left = self.eval_moments(self.domain[0] + eps) - self.eval_moments(self.domain[0])
        if self.decay_penalty[1]:
            right = -self.eval_moments(self.domain[1]) + self.eval_moments(self.domain[1] - eps)
        return np.stack((left[0, :], right[0, :]), axis=0) / eps / self._moment_errs[None, :]


def _composite_gauss(domain, n_intervals, degree):
    pt, w = np.polynomial.legendre.leggauss(degree)
    edges = np.linspace(domain[0], domain[1], n_intervals + 1)
    a, b = edges[:-1, None], edges[1:, None]
    return ((pt[None, :] + 1) / 2 * (b - a) + a).ravel(), (w[None, :] * (b - a) / 2).ravel()


# ----------------------------------------------------------------------------------------------------------
# diagnostics with an externally given density (Python callable): host quadrature, device moment evaluation
# ----------------------------------------------------------------------------------------------------------
def compute_exact_moments(moments_fn, density, tol=1e-10):
    """(reference: :330-346)"""
    a, b = moments_fn.domain
    out = np.zeros(moments_fn.size)
    for i in range(moments_fn.size):
        out[i] = integrate.quad(lambda x, i=i: float(np.ravel(moments_fn.eval(i, x))[0]) * density(x), a, b, epsabs=tol)[0]
    return out


def compute_semiexact_moments(moments_fn, density, tol=1e-10, n_intervals=256):
    """Moments of a given density on a composite 21-point Gauss-Legendre rule (reference: :349-378, there on
    QUADPACK's sub-intervals)."""
    pts, w = _composite_gauss(moments_fn.domain, n_intervals, 21)
    return (density(pts) * w) @ moments_fn.eval_all(pts)


def compute_exact_cov(moments_fn, density, tol=1e-10):
    """(reference: :381-399)"""
    a, b = moments_fn.domain
    size = moments_fn.size
    out = np.zeros((size, size))
    for i in range(size):
        for j in range(i + 1):
            def fn(x, i=i, j=j):
                m = np.ravel(moments_fn.eval_all(x))
                return m[i] * m[j] * density(x)
            out[j][i] = out[i][j] = integrate.quad(fn, a, b, epsabs=tol)[0]
    return out


def compute_semiexact_cov(moments_fn, density, tol=1e-10, n_intervals=256):
    """(reference: :402-438)"""
    pts, w = _composite_gauss(moments_fn.domain, n_intervals, 21)
    phi = moments_fn.eval_all(pts)
    return (phi.T * (density(pts) * w)) @ phi


def KL_divergence(prior_density, posterior_density, a, b):
    """D_KL(P | Q) with the positivity-preserving integrand of the reference (:443-459)."""
    def integrand(x):
        p = prior_density(x)
        q = max(posterior_density(x), 1e-300)
        return p * np.log(p / q) - p + q
    return max(integrate.quad(integrand, a, b, epsabs=1e-10)[0], 1e-10)


def L2_distance(prior_density, posterior_density, a, b):
    """(reference: :462-464)"""
    return np.sqrt(integrate.quad(lambda x: (posterior_density(x) - prior_density(x)) ** 2, a, b))[0]


# ----------------------------------------------------------------------------------------------------------
# orthogonalisation of the moments w.r.t. the estimated covariance
# ----------------------------------------------------------------------------------------------------------
def best_fit_all(values, range_a, range_b):
    """Best linear fit over all index windows [a, b) from the given candidates (reference: :538-556)."""
    best, best_value = None, np.inf
    for a in range_a:
        for b in range_b:
            if 0 <= a and a + 2 < b < len(values):
                fit, res, _, _, _ = np.polyfit(np.arange(a, b), values[a:b], deg=1, full=1)
                value = res / ((b - a) ** 2)
                if value < best_value:
                    best, best_value = (a, b, fit), value
    return best


def best_p1_fit(values):
    """Longest window with a small linear-fit residual, found coarse-to-fine (reference: :560-579)."""
    if len(values) > 12:
        end = len(values) - len(values) % 2
        a, b, _ = best_p1_fit(np.mean(values[:end].reshape((-1, 2)), axis=1))
        a, b = 2 * a, 2 * b
        return best_fit_all(values, [a - 1, a, a + 1], [b - 1, b, b + 1])
    idx = range(len(values))
    return best_fit_all(values, idx, idx)


def detect_treshold_slope_change(values, log=True):
    """Index from which the (log) eigenvalue sequence follows one slope; smaller ones are extrapolated
    (reference: :584-609)."""
    values = np.array(values)
    first_pos = 0
    if log:
        first_pos = int(np.argmax(values > 0))
        values[first_pos:] = np.log(values[first_pos:])
    a, b, fit = best_p1_fit(values[first_pos:])
    poly = np.poly1d(fit)
    i_treshold = a + first_pos
    mod_vals = values.copy()
    mod_vals[:i_treshold] = poly(np.arange(-first_pos, a))
    if log:
        mod_vals = np.exp(mod_vals)
    return i_treshold, mod_vals


def construct_ortogonal_moments(moments, cov, tol=None):
    """Basis orthonormal w.r.t. the (centred) covariance estimated from samples (reference: :756-841).

    centring M = I - e0-column of cov; eigen-decomposition of M cov M^T; eigenvalues under the threshold are cut;
    L = RQ-factor of M^T V diag(1/sqrt(ev)) so that the new basis is a lower-triangular combination of the old one.
    (Small dense LAPACK on the host, like the reference; R <= 128.)
    :return: TransformedMoments, (eigenvalues, threshold, L)"""
    size = moments.size
    centre = np.eye(size)
    centre[:, 0] = -cov[:, 0]
    with _SmallLapack():
        centred = centre @ cov @ centre.T
        if size >= 96:
            # LAPACK's MRRR driver (dsyevr) takes 1.9 ms for a 128 x 128 matrix on one thread where NumPy's divide-and-conquer
            # call (dsyevd) takes 4.7 ms; below ~100 rows NumPy's is the faster one.  Only eigenvalues and the SPAN of the kept
            # eigenvectors enter L (the R factor of the RQ step absorbs their signs): the result is the same to rounding.
            ev, evec = scipy.linalg.eigh(centred, driver="evr")
        else:
            ev, evec = np.linalg.eigh(centred)
    if tol is None:
        _, fixed = detect_treshold_slope_change(ev, log=True)
        threshold = int(np.argmax(ev - fixed[0] > 0))
    else:
        threshold = int(np.argmax(ev > tol))
    ev_kept = np.flip(ev[threshold:], axis=0)
    evec_kept = np.flip(evec[:, threshold:], axis=1)
    icov_sqrt_t = centre.T @ evec_kept * (1 / np.sqrt(ev_kept))[None, :]
    with _SmallLapack():
        r_nm, _ = scipy.linalg.rq(icov_sqrt_t, mode='full')
    l_mn = r_nm.T
    if l_mn[0, 0] < 0:
        l_mn = -l_mn
    return moments_mod.TransformedMoments(moments, l_mn), (ev, threshold, l_mn)
