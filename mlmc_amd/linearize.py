"""Product linearisation of the moment families: phi_i(t) phi_j(t) = sum_k c_ijk phi_k(t), k < K.

Why: the MEAN of the moment covariance (reference quantity_estimate.py:131-147 + :59-65 -- per sample the outer products
f f^T and c c^T, then the level sums of their difference) is linear in the per-sample products,

    sum_n (f_i f_j - c_i c_j)  =  sum_k c_ijk  sum_n (phi_k(fine_n) - phi_k(coarse_n)),

so for Legendre, monomial and Fourier moments the level means of the R x R covariance follow from the level sums of
K ~ 2 R MOMENTS -- one pass of the (vector-pipe) moments kernel, 7 instructions per term and pair, instead of R^2 / 16 matrix
instructions per four pairs -- and an R^2 x K host contraction.  The keep / drop decision of a sample is the one of the
domain transform, identical for every size of a family, so the sample counts are the covariance pass's own.  The identity is
exact for polynomials / trigonometric polynomials; Legendre coefficients are non-negative and sum to one (Adams 1878,
Neumann), so the contraction is as well conditioned as a convex combination.  Only the mean: the VARIANCE of the
covariance (squares of f_i f_j - c_i c_j) mixes fine and coarse values of one sample and stays on the matrix cores.

Used by quantity_estimate._estimate_mean for `estimate_mean(covariance(q, fn), variance=False)` -- the first pass of
Estimate.construct_density (reference estimator.py:304-331, which reads only the means)."""
import functools

import numpy as np

MAX_R = 128          # dense [R^2, K] coefficient matrix: 33 MB at R = 128


def extended_size(fn):
    """Moments needed to linearise the products of `fn`'s functions, or None when the family has no such identity here."""
    from .moments import Legendre, Monomial, Fourier
    R = fn.size
    if R > MAX_R:
        return None
    if type(fn) in (Legendre, Monomial):
        # Without clipping to the domain (safe_eval=False) a value outside it makes the HIGH extended terms overflow (P_k grows
        # like (x + sqrt(x^2 - 1))^k beyond 1) while the low products it is not needed for stay finite -- and 0 * inf in the
        # contraction would turn those into NaN.  The direct form touches, per entry, only its own two factors.
        if not getattr(fn, "_is_clip", True):
            return None
        if max(abs(float(fn.ref_domain[0])), abs(float(fn.ref_domain[1]))) > 1.0:      # monomials on a wider reference domain
            return None
        return 2 * R - 1
    if type(fn) is Fourier:
        return 4 * (R // 2) + 1
    return None


@functools.lru_cache(maxsize=8)
def legendre_products(R):
    """C [R * R, 2 R - 1]:  P_i P_j = sum_k C[i R + j, k] P_k.  Adams' formula with A(n) = (2n - 1)!! / n!:
    c_ijk = (2k + 1) / (2s + 1) * A(s - i) A(s - j) A(s - k) / A(s),  2s = i + j + k,  |i - j| <= k <= i + j, i + j + k even.
    Evaluated by ratio recurrences in extended precision (every factor is O(1): no factorial is ever formed), rounded once."""
    ld = np.longdouble
    K = 2 * R - 1
    I, J = np.meshgrid(np.arange(R), np.arange(R), indexing="ij")
    hi, lo = np.maximum(I, J), np.minimum(I, J)
    diff = (hi - lo).astype(ld)
    # first term k = |i - j| (s = max): c = (2 diff + 1) / (2 hi + 1) * A(diff) A(lo) / A(hi)
    #   A(diff) A(lo) / A(hi) = prod_{t = 1..lo} (2t - 1) / t * (diff + t) / (2 (diff + t) - 1)
    c = (2 * diff + 1) / (2 * hi.astype(ld) + 1)
    for t in range(1, R):
        tt = ld(t)
        fac = (2 * tt - 1) / tt * (diff + tt) / (2 * (diff + tt) - 1)
        c = np.where(t <= lo, c * fac, c)
    out = np.zeros((R, R, K), dtype=np.float64)
    for t in range(R):                       # term t: k = diff + 2t, s = hi + t
        active = t <= lo
        if not active.any():
            break
        k = (hi - lo) + 2 * t
        a_i, a_j = active.nonzero()
        out[a_i, a_j, k[active]] = c[active].astype(np.float64)
        # step t -> t + 1:  k += 2, s += 1;  A(n + 1) / A(n) = (2n + 1) / (n + 1)
        s = hi.astype(ld) + t
        kk = k.astype(ld)
        si, sj, sk = s - hi, s - lo, s - kk               # arguments of the three A's in the numerator (max / min of (i, j): the
                                                          # result is bit for bit symmetric in i, j)
        ratio = ((2 * kk + 5) / (2 * s + 3)) / ((2 * kk + 1) / (2 * s + 1))
        ratio = ratio * ((2 * si + 1) / (si + 1)) * ((2 * sj + 1) / (sj + 1))
        with np.errstate(all="ignore"):
            ratio = ratio * (sk / (2 * sk - 1))           # A(sk - 1) / A(sk); sk >= 1 while the term after this one exists
            ratio = ratio * ((s + 1) / (2 * s + 1))       # A(s) / A(s + 1)
        c = np.where(t + 1 <= lo, c * ratio, c)
    return out.reshape(R * R, K)


@functools.lru_cache(maxsize=8)
def monomial_products(R):
    """t^i t^j = t^(i + j)."""
    K = 2 * R - 1
    out = np.zeros((R, R, K))
    I, J = np.meshgrid(np.arange(R), np.arange(R), indexing="ij")
    out[I, J, I + J] = 1.0
    return out.reshape(R * R, K)


@functools.lru_cache(maxsize=8)
def fourier_products(R):
    """Basis of mlmc/moments.py:145-162: phi_0 = 1, phi_(2m-1) = cos(m t), phi_(2m) = sin(m t).  Product-to-sum formulas."""
    M = R // 2
    K = 4 * M + 1
    out = np.zeros((R, R, K))

    def kind(idx):                  # -> (frequency m, is_sin)
        return ((idx + 1) // 2, idx > 0 and idx % 2 == 0)

    def add(i, j, m, is_sin, coeff):
        if m < 0:                   # cos(-x) = cos x, sin(-x) = -sin x
            m = -m
            if is_sin:
                coeff = -coeff
        if m == 0:
            if not is_sin:
                out[i, j, 0] += coeff
            return
        out[i, j, 2 * m if is_sin else 2 * m - 1] += coeff

    for i in range(R):
        a, sa = kind(i)
        for j in range(R):
            b, sb = kind(j)
            if not sa and not sb:        # cos a cos b = 1/2 [cos(a - b) + cos(a + b)]
                add(i, j, a - b, False, 0.5)
                add(i, j, a + b, False, 0.5)
            elif sa and sb:              # sin a sin b = 1/2 [cos(a - b) - cos(a + b)]
                add(i, j, a - b, False, 0.5)
                add(i, j, a + b, False, -0.5)
            elif sa:                     # sin a cos b = 1/2 [sin(a + b) + sin(a - b)]
                add(i, j, a + b, True, 0.5)
                add(i, j, a - b, True, 0.5)
            else:                        # cos a sin b = 1/2 [sin(a + b) - sin(a - b)]
                add(i, j, a + b, True, 0.5)
                add(i, j, a - b, True, -0.5)
    return out.reshape(R * R, K)


def product_matrix(fn):
    """C [R * R, K] with phi_i phi_j = sum_k C[i R + j, k] phi_k, K = extended_size(fn)."""
    from .moments import Legendre, Monomial, Fourier
    if type(fn) is Legendre:
        return legendre_products(fn.size)
    if type(fn) is Monomial:
        return monomial_products(fn.size)
    if type(fn) is Fourier:
        return fourier_products(fn.size)
    raise TypeError("no product linearisation for {}".format(type(fn).__name__))


def covariance_sums_from_moment_sums(fn, s_mom, n_comp=1):
    """Level sums of the moment covariance from the level sums of the extended moments: s_mom [L, n_comp * K] (rows
    m * K + k) -> [L, n_comp * R * R] (rows m * R^2 + i * R + j, the cov_at_bottom layout)."""
    C = product_matrix(fn)
    R2, K = C.shape
    s = np.asarray(s_mom, dtype=np.float64)
    L = s.shape[0]
    return (s.reshape(L * n_comp, K) @ C.T).reshape(L, n_comp * R2)
