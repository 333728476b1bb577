"""Older maximum-entropy solver with size continuation and penalties (reference interface: mlmc/tool/distribution.py).

`Distribution` keeps the reference's staged solve (odd sizes growing geometrically by 1.2, per-stage tolerances,
end-point decay penalty with coefficient 10, stabilisation toward the previous stage's multipliers); every stage
is one `mlmc_maxent_solve` call on the MI355X instead of SciPy's trust-exact on QUADPACK quadratures.
"""
import numpy as np
import scipy.integrate as integrate
from scipy.optimize import OptimizeResult

from .simple_distribution import _cdf, _device_density, _solve_on_device


class Distribution:
    def __init__(self, moments_obj, moment_data, domain=None, force_decay=(True, True), monitor=False):
        self.moments_basis = moments_obj
        self.moments_fn = None
        if domain is None:
            domain = moments_obj.domain
        self.domain = domain
        self.decay_penalty = force_decay
        self.moment_means = moment_data[:, 0]
        self.moment_errs = np.sqrt(moment_data[:, 1])
        self.multipliers = None
        self.approx_size = len(self.moment_means)
        assert moments_obj.size >= self.approx_size
        self.moments_fn = moments_obj
        self._gauss_degree = 21
        self._penalty_coef = 10
        self.monitor = monitor
        self.n_intervals = 64

    def _stage_sizes(self):
        """Odd sizes shrinking by 1.2 from the full size down to about 5 (reference: distribution.py:98-109)."""
        if self.approx_size <= 5:
            return [self.approx_size]
        size = self.approx_size
        sizes = [size]
        while size > 4:
            size /= 1.2
            odd = 2 * round((size - 1) / 2) + 1
            if odd != sizes[-1]:
                sizes.append(odd)
        sizes.reverse()
        return sizes

    def estimate_density_minimize(self, tol=1e-5, reg_param=0.01):
        """(reference: distribution.py:85-157) -> OptimizeResult with x, success, nit (total), fun_norm"""
        self._reg_param = reg_param
        sizes = self._stage_sizes()
        self.approx_size = sizes[0]
        self._initialize_params(self.approx_size, tol)
        self.extend_size(self.approx_size)
        # gradient norm of the initial (uniform) guess sets the first stage's tolerance (:114-123)
        _, grad0, _, info0 = self._solve_stage(tol=np.inf, max_it=0)
        init_error = info0.grad_norm
        if len(sizes) == 1:
            tolerances = [tol]
        else:
            t0 = max(tol, init_error / 10)
            frac = (np.array(sizes) - sizes[0]) / (sizes[-1] - sizes[0])
            tolerances = np.exp(np.log(tol) * frac + np.log(t0) * (1 - frac))
        total_nit = 0
        info = info0
        grad = grad0
        for approx_size, approx_tol in zip(sizes, tolerances):
            self.extend_size(approx_size)
            lam, grad, hess, info = self._solve_stage(tol=approx_tol, max_it=200)
            self.multipliers = lam
            total_nit += info.nit
            if self.monitor:
                print("Iteration: size: {} nits: {} tol: {:5.3g} res: {:5.3g}".format(self.approx_size, info.nit, approx_tol,
                                                                                    info.grad_norm))
        jac_norm = info.grad_norm
        result = OptimizeResult()
        result.x = self.multipliers.copy()
        result.jac = grad
        result.fun = info.fun
        result.success = bool(info.success)
        # fix normalisation as the reference does (:149-151): divide the multipliers by the zeroth moment
        self.multipliers = self.multipliers / info.moment0
        if result.success or jac_norm < tol:
            result.success = True
        result.nit = total_nit
        result.fun_norm = jac_norm
        return result

    def estimate_density(self, tol=None):
        """Root of the gradient from the uniform initial guess at the full size, no size continuation ("faster, but worse
        stability", reference: distribution.py:159-181 -- `scipy.optimize.root(fun=gradient, jac=Hessian, tol=tol)`).
        As written the reference method cannot run: it calls `_initialize_params(tol)`, which binds `tol` to the `size`
        parameter and fails its own `assert tol is not None` (:216-223), and nothing in the reference calls it.  Built here
        to its evident intent: initial guess of `_initialize_params(approx_size, tol)`, no stabilisation term (there is no
        previous stage), the decay penalties of the functional as in every other solve; the Newton iteration of
        `mlmc_maxent_solve` finds the root.  -> OptimizeResult with x, fun (the gradient), success, nit, fun_norm; the
        normalisation is not touched, as in the reference.  The result is pinned to the reference through
        `estimate_density_minimize(reg_param=0)`, whose solution is the root of the same gradient."""
        assert tol is not None
        self._reg_param = 0.0
        self._initialize_params(self.approx_size, tol)
        self._last_solved_multipliers = None
        self._stab_penalty = 0.0
        self._moment_means = self.moment_means[:self.approx_size]
        self._moment_errs = self.moment_errs[:self.approx_size]
        lam, grad, hess, info = self._solve_stage(tol=tol, max_it=200)
        self.multipliers = lam
        result = OptimizeResult()
        result.x = lam.copy()
        result.fun = grad
        result.nit = info.nit
        result.fun_norm = float(np.linalg.norm(grad))
        result.success = bool(info.success) or result.fun_norm < tol
        return result

    def _solve_stage(self, tol, max_it):
        if max_it == 0:
            # evaluate only: one Newton set-up with an unreachable iteration budget returns gradient info at the start
            return _solve_on_device(self.moments_fn, self._moment_means, self._moment_errs, self.domain, self.multipliers,
                                    tol=1e300, max_it=1, n_intervals=self.n_intervals, gauss_degree=self._gauss_degree,
                                    stab_penalty=self._stab_penalty, penalty_coef=self._penalty_coef, decay=self.decay_penalty,
                                    prev=self._last_solved_multipliers)
        return _solve_on_device(self.moments_fn, self._moment_means, self._moment_errs, self.domain, self.multipliers, tol=tol,
                                max_it=max_it, n_intervals=self.n_intervals, gauss_degree=self._gauss_degree,
                                stab_penalty=self._stab_penalty, penalty_coef=self._penalty_coef, decay=self.decay_penalty,
                                prev=self._last_solved_multipliers)

    def density(self, value, moments_fn=None):
        """exp(-sum_i phi_i(x) lambda_i / sigma_i) (reference: distribution.py:182-193; the +-200 clip of the exponent,
        which the reference applies inside its solver only, is applied here too)"""
        fn = self.moments_fn if moments_fn is None else moments_fn
        return _device_density(fn, self.multipliers, self._moment_errs, value)

    def cdf(self, values):
        return _cdf(self, values)

    def _initialize_params(self, size, tol=None):
        assert self.domain is not None
        assert tol is not None
        self._quad_tolerance = tol / 16
        self.moment_errs[0] = np.min(self.moment_errs[1:]) / 8
        self.multipliers = np.zeros(size)
        self.multipliers[0] = -np.log(1 / (self.domain[1] - self.domain[0])) * self.moment_errs[0]
        self._quad_log = []

    def extend_size(self, new_size):
        """Grow the multiplier vector; remember the last solution for the stabilisation term (reference: :234-253)."""
        self._last_solved_multipliers = self.multipliers
        self._stab_penalty = self._reg_param / np.linalg.norm(self.multipliers)
        self.approx_size = new_size
        grown = np.zeros(new_size)
        grown[:len(self.multipliers)] = self.multipliers
        self.multipliers = grown
        self._moment_means = self.moment_means[:self.approx_size]
        self._moment_errs = self.moment_errs[:self.approx_size]

    def eval_moments(self, x):
        return self.moments_fn.eval_all(x, self.approx_size)


def compute_exact_moments(moments_fn, density, tol=1e-4):
    """(reference: distribution.py:423-438)"""
    a, b = moments_fn.domain
    out = np.zeros(moments_fn.size)
    for i in range(moments_fn.size):
        out[i] = integrate.quad(lambda x, m=i: float(np.ravel(moments_fn.eval(m, x))[0]) * density(x), a, b, epsabs=tol)[0]
    return out


def KL_divergence(prior_density, posterior_density, a, b):
    """(reference: distribution.py:441-450)"""
    integrand = lambda x: prior_density(x) * max(np.log(prior_density(x) / posterior_density(x)), -1e300)
    return max(integrate.quad(integrand, a, b, epsabs=1e-10)[0], 1e-10)


def L2_distance(prior_density, posterior_density, a, b):
    """(reference: distribution.py:453-455)"""
    return np.sqrt(integrate.quad(lambda x: (posterior_density(x) - prior_density(x)) ** 2, a, b))[0]
