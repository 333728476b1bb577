"""Generalised moment functions -- host-side mirror of the reference interface mlmc/moments.py.

Same class names, constructor arguments, attributes and error behaviour as the reference
(`Moments` :6-108, `Monomial` :111-130, `Fourier` :133-171, `Legendre` :174-229,
`TransformedMoments` :232-274), but every evaluation runs on the MI355X through libmlmc_hip.so
(`mlmc_basis_eval`); inside the estimators the moment functions are never materialised at all --
they are evaluated in registers by the fused accumulation kernels (see quantity/quantity_estimate.py).
"""
import ctypes as C

import numpy as np

from . import _lib


def log_keep_interval(shift, scale, ref0, ref1):
    """Raw-value keep interval of a log-domain moments object with safe_eval: (x_lo, x_hi) = the smallest / largest positive
    double x whose t = (np.log(x) - shift) * scale + ref0 -- exactly the reference's arithmetic, moments.py:27-39,69-70 --
    lies in [ref0, ref1] (Moments.clip keeps the closed interval, moments.py:58-67).  x -> t is monotone, so both ends are
    found by bisection over the bit patterns of the positive doubles (ordered like the values), <= 64 evaluations of
    NumPy's own log each.  The device then compares the RAW sample value with these thresholds (mlmc_basis_desc.x_lo /
    x_hi), so the sample counts equal the NumPy path's bit for bit whatever the last bit of the device's log is.
    Nothing kept: (inf, 0)."""
    shift, scale, ref0, ref1 = (np.float64(v) for v in (shift, scale, ref0, ref1))
    buf = np.empty(1, dtype=np.uint64)

    def t_of(bits):
        buf[0] = bits
        with np.errstate(all="ignore"):
            return ((np.log(buf.view(np.float64)) - shift) * scale + ref0)[0]      # array arithmetic, like the reference

    top = 0x7ff0000000000000                  # +inf; 1 .. top - 1 are the positive finite doubles
    lo, hi = 1, top
    while lo < hi:                            # first x with t >= ref0
        mid = lo + (hi - lo) // 2
        if t_of(mid) >= ref0:
            hi = mid
        else:
            lo = mid + 1
    first = lo
    lo, hi = 0, top - 1
    while lo < hi:                            # last x with t <= ref1
        mid = lo + (hi - lo + 1) // 2
        if t_of(mid) <= ref1:
            lo = mid
        else:
            hi = mid - 1
    last = lo
    if first >= top or last == 0 or first > last:
        return float("inf"), 0.0
    pair = np.array([first, last], dtype=np.uint64).view(np.float64)
    return float(pair[0]), float(pair[1])


class Moments:
    """Base class: domain transform parameters + device handle (reference: moments.py:6-108)."""
    _kind = None
    _default_ref_domain = None

    def __init__(self, size, domain, log=False, safe_eval=True):
        assert size > 0
        self.size = size
        self.domain = domain
        self._is_log = log
        self._is_clip = safe_eval
        lo, hi = (np.log(domain[0]), np.log(domain[1])) if log else (domain[0], domain[1])
        width = hi - lo
        assert width > 0
        width = max(width, 1e-15)
        self._linear_scale = (self.ref_domain[1] - self.ref_domain[0]) / width
        self._linear_shift = lo
        self._handle = None
        self._aux_handle = None

    # ---- device plumbing -------------------------------------------------------------------
    def _desc(self, kind=None, size=None, matrix=None):
        d = _lib.BasisDesc()
        d.kind = self._kind if kind is None else kind
        d.size = int(self.size if size is None else size)
        d.shift = float(self._linear_shift)
        d.scale = float(self._linear_scale)
        d.ref0 = float(self.ref_domain[0])
        d.ref1 = float(self.ref_domain[1])
        d.is_log = int(bool(self._is_log))
        d.is_clip = int(bool(self._is_clip))
        d.out_size = 0
        d.matrix = None
        d.x_lo, d.x_hi = 0.0, 0.0
        if self._is_log and self._is_clip:
            if getattr(self, "_log_keep", None) is None:
                self._log_keep = log_keep_interval(d.shift, d.scale, d.ref0, d.ref1)
            d.x_lo, d.x_hi = self._log_keep
        if matrix is not None:
            d.out_size = int(matrix.shape[0])
            d.matrix = matrix.ctypes.data_as(C.POINTER(C.c_double))
        return d

    def _basis_handle(self):
        """Opaque mlmc_basis* for the accumulation / PDF kernels (created on first use)."""
        if self._handle is None:
            h = C.c_void_p()
            _lib.check(_lib.lib().mlmc_basis_create(C.byref(self._desc()), C.byref(h)))
            self._handle = h
        return self._handle

    def __del__(self):
        try:
            for name in ("_handle", "_aux_handle"):
                h = getattr(self, name, None)
                if h is not None and _lib._lib is not None:
                    _lib._lib.mlmc_basis_destroy(h)
                    setattr(self, name, None)
        except Exception:
            pass

    def _device_eval(self, handle, value, size):
        value = np.atleast_1d(np.asarray(value, dtype=np.float64))
        flat = np.ascontiguousarray(value.reshape(-1))
        out = np.empty((flat.size, size), dtype=np.float64)
        _lib.check(_lib.lib().mlmc_basis_eval(handle, _lib.ptr(flat), flat.size, int(size), _lib.ptr(out), _lib.HOST))
        return out.reshape(value.shape + (size,))

    def _eval_all(self, value, size):
        if size > self.size:
            # the reference evaluates any requested size; keep that by building a larger twin
            return self.change_size(size)._eval_all(value, size)
        return self._device_eval(self._basis_handle(), value, size)

    # ---- reference interface -----------------------------------------------------------------
    def __eq__(self, other):
        return type(self) is type(other) and self.size == other.size and np.all(self.domain == other.domain) \
            and self._is_log == other._is_log and self._is_clip == other._is_clip

    __hash__ = object.__hash__

    def change_size(self, size):
        return self.__class__(size, self.domain, log=self._is_log, safe_eval=self._is_clip)

    def transform(self, value):
        """Map values to the reference domain; out-of-domain values become NaN when safe_eval (device evaluated:
        the monomial t^1 of the same transform)."""
        if self._aux_handle is None:
            h = C.c_void_p()
            _lib.check(_lib.lib().mlmc_basis_create(C.byref(self._desc(kind=_lib.MONOMIAL, size=2)), C.byref(h)))
            self._aux_handle = h
        return self._device_eval(self._aux_handle, value, 2)[..., 1]

    def inv_transform(self, ref):
        lin = (np.asarray(ref) - self.ref_domain[0]) / self._linear_scale + self._linear_shift
        return np.exp(lin) if self._is_log else lin

    def linear(self, value):
        return (value - self._linear_shift) * self._linear_scale + self.ref_domain[0]

    def inv_linear(self, value):
        return (value - self.ref_domain[0]) / self._linear_scale + self._linear_shift

    def clip(self, value):
        value = np.array(value, dtype=np.float64, copy=True)
        value[(value < self.ref_domain[0]) | (value > self.ref_domain[1])] = np.nan
        return value

    def __call__(self, value):
        return self._eval_all(value, self.size)

    def eval(self, i, value):
        return self._eval_all(value, i + 1)[:, -1]

    def eval_single_moment(self, i, value):
        return self._eval_all(value, i + 1)[..., i]

    def eval_all(self, value, size=None):
        return self._eval_all(value, self.size if size is None else size)

    def eval_all_der(self, value, size=None, degree=1):
        return self._eval_all_der(value, self.size if size is None else size, degree)

    def eval_diff(self, value, size=None):
        return self._eval_diff(value, self.size if size is None else size)

    def eval_diff2(self, value, size=None):
        return self._eval_diff2(value, self.size if size is None else size)


class Monomial(Moments):
    """t^k, k < size, on ref_domain (0, 1) by default (reference: moments.py:111-130)."""
    _kind = _lib.MONOMIAL

    def __init__(self, size, domain=(0, 1), ref_domain=None, log=False, safe_eval=True):
        self.ref_domain = ref_domain if ref_domain is not None else (0, 1)
        super().__init__(size, domain, log=log, safe_eval=safe_eval)

    def change_size(self, size):
        return Monomial(size, self.domain, ref_domain=self.ref_domain, log=self._is_log, safe_eval=self._is_clip)

    def eval(self, i, value):
        return self._eval_all(value, i + 1)[..., i]


class Fourier(Moments):
    """[1, cos t, sin t, cos 2t, sin 2t, ...] on ref_domain (0, 2 pi) (reference: moments.py:133-171).
    Unlike the reference (np.outer, 1-D only) any input shape is accepted."""
    _kind = _lib.FOURIER

    def __init__(self, size, domain=(0, 2 * np.pi), ref_domain=None, log=False, safe_eval=True):
        self.ref_domain = ref_domain if ref_domain is not None else (0, 2 * np.pi)
        super().__init__(size, domain, log=log, safe_eval=safe_eval)

    def change_size(self, size):
        return Fourier(size, self.domain, ref_domain=self.ref_domain, log=self._is_log, safe_eval=self._is_clip)

    def _eval_all(self, value, size):
        out = super()._eval_all(value, size)
        out[..., 0] = 1         # the reference fills column 0 with the constant, also in rows of masked values (moments.py:156)
        return out

    def eval(self, i, value):
        # the reference's Fourier.eval (:164-171) disagrees with its own _eval_all; column i of eval_all is the truth
        return self._eval_all(value, i + 1)[..., i]


class Legendre(Moments):
    """Legendre polynomials P_0..P_{size-1} on ref_domain (-1, 1) (reference: moments.py:174-229)."""
    _kind = _lib.LEGENDRE

    def __init__(self, size, domain, ref_domain=None, log=False, safe_eval=True):
        self.ref_domain = ref_domain if ref_domain is not None else (-1, 1)
        # P_m' = sum_{n = m-1, m-3, ...} (2n + 1) P_n
        self.diff_mat = np.zeros((size, size))
        for n in range(size - 1):
            self.diff_mat[n, n + 1::2] = 2 * n + 1
        self.diff2_mat = self.diff_mat @ self.diff_mat
        super().__init__(size, domain, log, safe_eval)

    def change_size(self, size):
        return Legendre(size, self.domain, ref_domain=self.ref_domain, log=self._is_log, safe_eval=self._is_clip)

    def _eval_value(self, x, size):
        """Legendre values of already transformed x (reference :190-193)."""
        ident = Legendre(size, (self.ref_domain[0], self.ref_domain[1]), ref_domain=self.ref_domain, safe_eval=False)
        return ident._eval_all(x, size)

    def _eval_diff(self, value, size):
        return self._eval_all(value, size) @ self.diff_mat[:size, :size]

    def _eval_diff2(self, value, size):
        return self._eval_all(value, size) @ self.diff2_mat[:size, :size]

    def _eval_all_der(self, value, size, degree=1):
        return self._eval_all(value, size) @ np.linalg.matrix_power(self.diff_mat[:size, :size], degree)


class Spline(Moments):
    """Cubic B-spline moments on ref_domain (0, 1): phi_0 = 1, phi_r = B_r (r = 1..size-1) of the clamped uniform cubic
    B-spline basis B_0..B_{size-1} (size - 3 knot spans).  NOT part of this reference version (it only imports
    scipy.interpolate.BSpline, moments.py:3): defined here and pinned against scipy's BSpline."""
    _kind = _lib.SPLINE

    def __init__(self, size, domain, ref_domain=None, log=False, safe_eval=True):
        assert size >= 4, "cubic spline moments need size >= 4"
        self.ref_domain = ref_domain if ref_domain is not None else (0, 1)
        super().__init__(size, domain, log=log, safe_eval=safe_eval)

    def change_size(self, size):
        return Spline(size, self.domain, ref_domain=self.ref_domain, log=self._is_log, safe_eval=self._is_clip)

    def knots(self):
        """Knot vector on the unit interval (the one to hand to scipy.interpolate.BSpline)."""
        ns = self.size - 3
        return np.clip(np.arange(-3, ns + 4), 0, ns) / ns

    def eval(self, i, value):
        return self._eval_all(value, i + 1)[..., i]


class TransformedMoments(Moments):
    """new_moments = matrix . old_moments (reference: moments.py:232-274)."""

    def __init__(self, other_moments, matrix):
        matrix = np.asarray(matrix, dtype=np.float64)
        n, m = matrix.shape
        assert m == other_moments.size
        self.size = n
        self.domain = other_moments.domain
        self._origin = other_moments
        self._transform = matrix
        self._handle = None
        self._aux_handle = None
        # flatten a chain of transforms down to the underlying polynomial family
        base, mat = other_moments, matrix
        while isinstance(base, TransformedMoments):
            mat = mat @ base._transform
            base = base._origin
        self._base = base
        self._base_matrix = np.ascontiguousarray(mat)
        self.ref_domain = base.ref_domain
        self._is_log = base._is_log
        self._is_clip = base._is_clip
        self._linear_scale = base._linear_scale
        self._linear_shift = base._linear_shift
        self._kind = base._kind

    def _basis_handle(self):
        if self._handle is None:
            h = C.c_void_p()
            d = self._base._desc(matrix=self._base_matrix)
            _lib.check(_lib.lib().mlmc_basis_create(C.byref(d), C.byref(h)))
            self._handle = h
        return self._handle

    def __eq__(self, other):
        return type(self) is type(other) and self.size == other.size and self._origin == other._origin \
            and np.all(self._transform == other._transform)

    __hash__ = object.__hash__

    def change_size(self, size):
        return TransformedMoments(self._origin, self._transform[:size])

    def transform(self, value):
        return self._base.transform(value)

    def inv_transform(self, ref):
        return self._base.inv_transform(ref)

    def _eval_all(self, value, size):
        return self._device_eval(self._basis_handle(), value, size)

    def _eval_all_der(self, value, size, degree=1):
        return np.matmul(self._origin._eval_all_der(value, self._origin.size, degree=degree), self._transform.T)[..., :size]

    def _eval_diff(self, value, size):
        return np.matmul(self._origin.eval_diff(value, self._origin.size), self._transform.T)[..., :size]

    def _eval_diff2(self, value, size):
        return np.matmul(self._origin.eval_diff2(value, self._origin.size), self._transform.T)[..., :size]
