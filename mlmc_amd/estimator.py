"""`Estimate` facade and sample-allocation helpers (reference interface: mlmc/estimator.py:11-450).

Every pass over the samples (moments, covariance, level variances) goes through
`quantity_estimate.estimate_mean`, i.e. through the HIP accumulation kernels.  The O(L R) post-processing
(variance regression over levels, n-samples allocation) is host arithmetic, as the numbers involved are tiny.
"""
import numpy as np

from . import engine
from .quantity import quantity_estimate as qe
from .quantity.quantity_types import ScalarType


class Estimate:
    """Wrapper methods for moment estimation, sample allocation and PDF reconstruction (reference: estimator.py:11-341)."""

    def __init__(self, quantity, sample_storage, moments_fn=None):
        self._quantity = quantity
        self._sample_storage = sample_storage
        self._moments_fn = moments_fn

    @property
    def quantity(self):
        return self._quantity

    @quantity.setter
    def quantity(self, quantity):
        self._quantity = quantity

    @property
    def n_moments(self):
        return self._moments_fn.size

    # ---- estimates (device passes) -------------------------------------------------------------------
    def estimate_moments(self, moments_fn=None):
        """-> (moment means, variances of these estimates), arrays of length n_moments (reference: :32-42)."""
        if moments_fn is None:
            moments_fn = self._moments_fn
        r = qe.estimate_mean(qe.moments(self._quantity, moments_fn))
        return r.mean, r.var

    def estimate_covariance(self, moments_fn=None):
        """-> (covariance matrix of the moments, variance of its entries) (reference: :44-54)."""
        if moments_fn is None:
            moments_fn = self._moments_fn
        r = qe.estimate_mean(qe.covariance(self._quantity, moments_fn))
        self._cov_memo = (self._memo_key(moments_fn), r, moments_fn, self._quantity)
        return r.mean, r.var

    def _memo_key(self, moments_fn):
        """What a kept covariance estimate is valid for: this quantity object, these moment functions, the stamps of the
        storage's levels (samples collected + modification counts, quantity_estimate._level_stamps)."""
        try:
            stamps = qe._level_stamps(self._quantity.get_quantity_storage())
        except Exception:
            return None
        return (stamps, qe.device_cache_generation())

    def _diff_vars_from_covariance(self, moments_fn):
        """Level sums of the moments out of the last covariance estimate of the same (quantity, moments_fn, samples):
        row 0 of the covariance rows is phi_0 phi_j = phi_j, so its level means / variances ARE those of the moments
        (engine.moments_from_covariance) -- no second pass over the samples.  None when no such estimate is at hand."""
        memo = getattr(self, "_cov_memo", None)
        # the memo holds the quantity and the moment functions themselves (compared with `is`: an id() can be reused)
        if memo is None or memo[0] is None or memo[0] != self._memo_key(moments_fn) or memo[2] is not moments_fn \
                or memo[3] is not self._quantity:
            return None
        from .moments import Legendre, Monomial, Fourier, Spline
        if type(moments_fn) not in (Legendre, Monomial, Fourier, Spline):
            return None          # only the families known to start with the constant: phi_0 = 1 (moments.py:122-126,145-162,195-197)
        r, size = memo[1], moments_fn.size
        n_levels = r._l_means.shape[0]
        n_comp = r._l_means.shape[1] // (size * size)
        l_means, l_vars = engine.moments_from_covariance(r._l_means, r._l_vars, size, n_comp=n_comp)
        from .quantity.quantity import QuantityMean
        return QuantityMean(qe.moments(self._quantity, moments_fn).qtype, l_means=l_means.reshape(n_levels, -1),
                            l_vars=l_vars.reshape(n_levels, -1), n_samples=r.n_samples, n_rm_samples=r.n_rm_samples)

    def estimate_diff_vars(self, moments_fn=None):
        """-> (variances of the level differences [L, R], n_samples [L]) (reference: :76-85).  After estimate_covariance
        of the same moment functions on unchanged samples the variances are read from the covariance estimate."""
        if moments_fn is None:
            moments_fn = self._moments_fn
        r = self._diff_vars_from_covariance(moments_fn)
        if r is None:
            r = qe.estimate_mean(qe.moments(self._quantity, moments_fn))
        return r.l_vars, r.n_samples

    def estimate_diff_vars_regression(self, n_created_samples, moments_fn=None, raw_vars=None):
        """Level variances smoothed by a log-quadratic model in the level step (reference: :56-74).
        -> (vars [L, R], n_ops [L])"""
        self._n_created_samples = n_created_samples
        if raw_vars is None:
            if moments_fn is None:
                moments_fn = self._moments_fn
            raw_vars, _ = self.estimate_diff_vars(moments_fn)
        sim_steps = np.squeeze(self._sample_storage.get_level_parameters())
        return self._all_moments_variance_regression(raw_vars, sim_steps), self._sample_storage.get_n_ops()

    def _all_moments_variance_regression(self, raw_vars, sim_steps):
        """Per-moment regression of the level variances (reference: :87-93), all moments in ONE least-squares solve with
        several right-hand sides (the design matrix [1, log h, log^2 h] is the same for every moment)."""
        raw_vars = np.asarray(raw_vars, dtype=np.float64)
        reg_vars = np.array(raw_vars, copy=True)
        n_levels = raw_vars.shape[0]
        if n_levels >= 3:
            # moments whose level variances are all (close to) zero are left alone, as np.allclose(raw_vars[:, m], 0) decides it
            cols = 1 + np.flatnonzero(~np.all(np.isclose(raw_vars[:, 1:], 0), axis=0))
            if cols.size:
                log_h = np.log(np.asarray(sim_steps, dtype=np.float64)[1:])
                design = np.stack([np.ones(n_levels - 1), log_h, log_h ** 2], axis=1)
                with np.errstate(all="ignore"):
                    params = np.linalg.lstsq(design, np.log(raw_vars[1:][:, cols]), rcond=None)[0]
                    reg_vars[1:, cols] = np.exp(design @ params)
        assert np.allclose(reg_vars[:, 0], 0.0)
        return reg_vars

    def _moment_variance_regression(self, raw_vars, sim_steps):
        """log var_l = A + B log h_l + C log^2 h_l fitted on levels 1..L-1, unweighted; level 0 is kept
        (reference: :95-134 -- its chi-square weights are computed and then overwritten by ones, :111-113)."""
        raw_vars = np.asarray(raw_vars, dtype=np.float64)
        n_levels = raw_vars.shape[0]
        if n_levels < 3 or np.allclose(raw_vars, 0):
            return raw_vars
        log_h = np.log(np.asarray(sim_steps, dtype=np.float64)[1:])
        design = np.stack([np.ones(n_levels - 1), log_h, log_h ** 2], axis=1)
        params = np.linalg.lstsq(design, np.log(raw_vars[1:]), rcond=None)[0]
        fitted = raw_vars.copy()
        fitted[1:] = np.exp(design @ params)
        return fitted

    def _variance_of_variance(self, n_samples=None):
        """Variance of log(chi^2_{n-1} / (n-1)) per level (reference: :136-169).  Closed form instead of the reference's
        two adaptive quadratures per level: Var[log X], X ~ chi2_df / df, equals trigamma(df / 2)."""
        from scipy.special import polygamma
        if n_samples is None:
            n_samples = self._n_created_samples
        return np.array([float(polygamma(1, (ns - 1) / 2.0)) for ns in n_samples])

    # ---- bootstrap ------------------------------------------------------------------------------------
    def est_bootstrap(self, n_subsamples=100, sample_vector=None, moments_fn=None):
        """Bootstrap statistics of the estimates over random sub-samples (reference: :171-205)."""
        if moments_fn is not None:
            self._moments_fn = moments_fn
        else:
            moments_fn = self._moments_fn
        sample_vector = determine_sample_vec(n_collected_samples=self._sample_storage.get_n_collected(),
                                             n_levels=self._sample_storage.get_n_levels(), sample_vector=sample_vector)
        bs_mean, bs_var, bs_l_means, bs_l_vars = [], [], [], []
        for _ in range(n_subsamples):
            # The reference writes quantity.select(quantity.subsample(...)) here (estimator.py:186), which indexes the
            # chunk with the picked *values* and raises IndexError; its own test (test_quantity_concept.py:630-648) uses
            # the sub-sampled quantity directly, as done here.
            sub = self.quantity.subsample(sample_vec=sample_vector)
            q_mean = qe.estimate_mean(qe.moments(sub, moments_fn=moments_fn, mom_at_bottom=False))
            bs_mean.append(q_mean.mean)
            bs_var.append(q_mean.var)
            bs_l_means.append(q_mean.l_means)
            bs_l_vars.append(q_mean.l_vars)
        self.mean_bs_mean = np.mean(bs_mean, axis=0)
        self.mean_bs_var = np.mean(bs_var, axis=0)
        self.mean_bs_l_means = np.mean(bs_l_means, axis=0)
        self.mean_bs_l_vars = np.mean(bs_l_vars, axis=0)
        self.var_bs_mean = np.var(bs_mean, axis=0, ddof=1)
        self.var_bs_var = np.var(bs_var, axis=0, ddof=1)
        self.var_bs_l_means = np.var(bs_l_means, axis=0, ddof=1)
        self.var_bs_l_vars = np.var(bs_l_vars, axis=0, ddof=1)
        n_coll = np.array(self._sample_storage.get_n_collected())
        # [L, R] for scalar quantities (reference: `[:, None]`); array-typed quantities carry extra trailing axes
        self._bs_level_mean_variance = self.var_bs_l_means * n_coll.reshape((-1,) + (1,) * (self.var_bs_l_means.ndim - 1))

    def bs_target_var_n_estimated(self, target_var, sample_vec=None):
        sample_vec = determine_sample_vec(n_collected_samples=self._sample_storage.get_n_collected(),
                                          n_levels=self._sample_storage.get_n_levels(), sample_vector=sample_vec)
        self.est_bootstrap(n_subsamples=300, sample_vector=sample_vec)
        variances, n_ops = self.estimate_diff_vars_regression(sample_vec, raw_vars=self.mean_bs_l_vars)
        return estimate_n_samples_for_target_variance(target_var, variances, n_ops,
                                                      n_levels=self._sample_storage.get_n_levels())

    # ---- domain ------------------------------------------------------------------------------------------
    @staticmethod
    def estimate_domain(quantity, sample_storage, quantile=None):
        """Moments domain from sample quantiles of the fine samples (reference: :275-302).  As in the reference,
        `chunks(n_samples=...)` without a level id yields the level-0 chunk for every level."""
        if quantile is None:
            quantile = 0.01
        ranges = []
        for level_id in range(sample_storage.get_n_levels()):
            chunk_spec = next(sample_storage.chunks(n_samples=sample_storage.get_n_collected()[level_id]))
            fine = qe.fine_samples_for_device(quantity, chunk_spec)
            # NaN removal + np.percentile of the reference (:298-299) as one device call (exact radix select)
            ranges.append(engine.percentiles(fine, [100 * quantile, 100 * (1 - quantile)]))
        ranges = np.array(ranges)
        return np.min(ranges[:, 0]), np.max(ranges[:, 1])

    # ---- PDF ---------------------------------------------------------------------------------------------
    def construct_density(self, tol=1e-8, reg_param=0.0, orth_moments_tol=1e-4, exact_pdf=None):
        """Maximum-entropy density from the estimated moments (reference: :304-331).
        -> (distr_obj, info, result, moments_obj)"""
        from .tool import simple_distribution
        if not isinstance(self._quantity.qtype, ScalarType):
            raise NotImplementedError("Currently, we only support ScalarType quantities")
        # only the means of the two estimates are used (the reference overwrites the variances with ones, :323)
        cov_mat = qe.estimate_mean(qe.covariance(self._quantity, self._moments_fn), variance=False).mean
        moments_obj, info = simple_distribution.construct_ortogonal_moments(self._moments_fn, cov_mat, tol=orth_moments_tol)
        est_moments = qe.estimate_mean(qe.moments(self._quantity, moments_obj), variance=False).mean
        est_vars = np.ones(moments_obj.size)        # the reference discards the estimated variances (:323)
        moments_data = np.stack((est_moments, est_vars), axis=1)
        distr_obj = simple_distribution.SimpleDistribution(moments_obj, moments_data, domain=moments_obj.domain)
        result = distr_obj.estimate_density_minimize(tol, reg_param)
        return distr_obj, info, result, moments_obj

    def get_level_samples(self, level_id, n_samples=None):
        chunk_spec = next(self._sample_storage.chunks(level_id=level_id, n_samples=n_samples))
        return self._quantity.samples(chunk_spec=chunk_spec)


def estimate_domain(quantity, sample_storage, quantile=None):
    """Module-level variant (reference: :344-363): percentile range of the fine samples of every level -- the first
    n_collected[0] samples of each (`ChunkSpec(level_id, n_samples=n_collected()[0])`), NaNs not removed (np.percentile
    then yields NaN, :360).  The samples stay where the device wants them (quantity_estimate.fine_samples_for_device)."""
    if quantile is None:
        quantile = 0.01
    ranges = []
    for level_id in range(sample_storage.get_n_levels()):
        n0 = sample_storage.get_n_collected()[0]
        chunk_spec = next(sample_storage.chunks(level_id=level_id, n_samples=n0))
        fine = qe.fine_samples_for_device(quantity, chunk_spec)
        ranges.append(engine.percentiles(fine, [100 * quantile, 100 * (1 - quantile)], nan_policy="propagate"))
    ranges = np.array(ranges)
    return np.min(ranges[:, 0]), np.max(ranges[:, 1])


def estimate_n_samples_for_target_variance(target_variance, prescribe_vars, n_ops, n_levels):
    """Optimal samples per level for a target variance of every moment (reference: :366-385).

    n_l = round( sqrt(V_l / C_l) * sum_k sqrt(V_k C_k) / target ), capped by V_l L / target, at least 2; max over moments.
    :return: int array [L]"""
    variances = np.asarray(prescribe_vars, dtype=np.float64)
    n_ops = np.asarray(n_ops, dtype=np.float64)
    sqrt_var_cost = np.sqrt(variances.T * n_ops)                    # [R, L]
    total = np.sum(sqrt_var_cost, axis=1)                           # [R]
    n_est = np.round((sqrt_var_cost / n_ops).T * total / target_variance).astype(int)     # [L, R]
    n_safe = np.maximum(np.minimum(n_est, variances * n_levels / target_variance), 2)
    return np.max(n_safe, axis=1).astype(int)


def _geometric_level_params(step_range, n_levels):
    assert step_range[0] > step_range[1]
    params = []
    for i in range(n_levels):
        frac = 1 if n_levels == 1 else i / (n_levels - 1)
        params.append([step_range[0] ** (1 - frac) * step_range[1] ** frac])
    return params


def calc_level_params(step_range, n_levels):
    """Geometric sequence of level steps (reference: :388-398)."""
    return _geometric_level_params(step_range, n_levels)


def determine_level_parameters(n_levels, step_range):
    """Geometric sequence of level steps (reference: :409-426)."""
    return _geometric_level_params(step_range, n_levels)


def determine_sample_vec(n_collected_samples, n_levels, sample_vector=None):
    """(reference: :401-406)"""
    if sample_vector is None:
        sample_vector = n_collected_samples
    if len(sample_vector) > n_levels:
        sample_vector = sample_vector[:n_levels]
    return np.array(sample_vector)


def determine_n_samples(n_levels, n_samples=None):
    """Initial number of samples per level: geometric interpolation between n0 and nL (reference: :429-450)."""
    if n_samples is None:
        n_samples = [100, 3]
    n_samples = np.atleast_1d(n_samples)
    if len(n_samples) == 1:
        n_samples = np.array([n_samples[0], 3])
    if len(n_samples) == 2:
        n0, n_last = n_samples
        n_samples = np.round(np.exp2(np.linspace(np.log2(n0), np.log2(n_last), n_levels))).astype(int)
    return n_samples
