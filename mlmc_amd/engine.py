"""Level accumulators: thin Python face of mlmc_accum_* (include/mlmc_hip.h).

One `LevelAccumulator` corresponds to one `estimate_mean()` call of the reference
(mlmc/quantity/quantity_estimate.py:22-80): chunks of raw fine/coarse samples are pushed level by
level, the HIP kernels evaluate the moment functions, mask out-of-domain / NaN samples and accumulate
the level-difference sums; `finalize()` returns per-level counts and sums.

Multi-GPU: every rank (one process per GPU) pushes its shard of each level's samples; `finalize()`
then all-reduces the packed partial sums with torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests).  No other collective exists on this path.
"""
import ctypes as C
import os

import numpy as np

from . import _lib


class _IdentityBasis:
    """Plain quantity (no moments node): the single 'moment' is the value itself, mask = isnan."""
    size = 1

    def __init__(self):
        self._handle = None

    def _basis_handle(self):
        if self._handle is None:
            d = _lib.BasisDesc()
            d.kind, d.size, d.shift, d.scale, d.ref0, d.ref1 = _lib.IDENTITY, 1, 0.0, 1.0, 0.0, 0.0
            d.is_log = d.is_clip = d.out_size = 0
            d.matrix = None
            h = C.c_void_p()
            _lib.check(_lib.lib().mlmc_basis_create(C.byref(d), C.byref(h)))
            self._handle = h
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None and _lib._lib is not None:
                _lib._lib.mlmc_basis_destroy(self._handle)
        except Exception:
            pass


def _dist_group_active(group):
    try:
        import torch.distributed as dist
    except Exception:
        return False
    if not (dist.is_available() and dist.is_initialized()):
        return False
    # MLMC_HIP_FORCE_DIST=1 exercises the all-reduce path with a single rank (used by the 1-GPU test of the RCCL path)
    return dist.get_world_size(group) > 1 or os.environ.get("MLMC_HIP_FORCE_DIST") == "1"


class LevelAccumulator:
    MOMENTS = _lib.MODE_MOMENTS
    COV = _lib.MODE_COV

    def __init__(self, moments_fn, n_levels, mode=_lib.MODE_MOMENTS, n_comp=1, mean_only=False):
        self._moments_fn = moments_fn if moments_fn is not None else _IdentityBasis()
        self.n_levels = int(n_levels)
        self.mode = int(mode)
        self.n_comp = int(n_comp)
        r = self._moments_fn.size
        self.rows_per_comp = r if mode == _lib.MODE_MOMENTS else r * r
        self.K = self.n_comp * self.rows_per_comp
        h = C.c_void_p()
        # mean_only: only the level means will be read; passes that exist for the variances alone are skipped (sp = NaN)
        _lib.check(_lib.lib().mlmc_accum_create(self._moments_fn._basis_handle(), self.n_levels,
                                                self.mode | (_lib.MODE_MEAN_ONLY if mean_only else 0), self.n_comp, C.byref(h)))
        self._h = h
        self._keepalive = []

    def close(self):
        # at interpreter shutdown the module globals may already be gone (`_lib` is None then): nothing left to free
        if getattr(self, "_h", None) is not None and _lib is not None and _lib._lib is not None:
            _lib._lib.mlmc_accum_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        _lib.check(_lib.lib().mlmc_accum_reset(self._h))
        self._keepalive = []

    def push(self, level, fine, coarse=None):
        """fine / coarse: [n] or [M, n] float64, NumPy arrays (host) or torch CUDA tensors (HBM resident)."""
        if isinstance(fine, np.ndarray):
            fine = _lib.as_f64(fine)
            coarse = None if coarse is None else _lib.as_f64(coarse)
        else:
            assert fine.is_contiguous() and (coarse is None or coarse.is_contiguous())
            self._keepalive.append((fine, coarse))   # launches are asynchronous
        n = fine.shape[-1]
        m = 1 if fine.ndim == 1 else fine.shape[0]
        if m != self.n_comp:
            raise ValueError("push: expected {} components, got {}".format(self.n_comp, m))
        if coarse is not None and tuple(coarse.shape) != tuple(fine.shape):
            raise ValueError("push: fine and coarse shapes differ")
        _lib.check(_lib.lib().mlmc_accum_push(self._h, int(level), _lib.ptr(fine), _lib.ptr(coarse), int(n),
                                              _lib.mem_kind(fine)))

    def estimate(self, chunks, group=None, reduce=True):
        """reset + push + finalize in one call of the C ABI (`mlmc_accum_estimate`): chunks = [(level, fine, coarse | None)]
        of torch CUDA tensors, shape [n] ([M, n] for vector quantities).  With an initialised process group of more than
        one rank the chunks are this rank's shard and the packed partial sums are all-reduced once
        (`mlmc_accum_estimate_packed` + allreduce_partials).  -> n[L], n_rm[L], s[L, K], sp[L, K]"""
        k = len(chunks)
        key = tuple((int(c[0]), c[1].data_ptr(), 0 if c[2] is None else c[2].data_ptr(), int(c[1].shape[-1])) for c in chunks)
        cached = getattr(self, "_est_args", None)
        if cached is None or cached[0] != key:              # the argument arrays of a repeated estimate are reused
            cached = self._est_args = (key, (C.c_int32 * k)(*[c[0] for c in key]), (C.c_void_p * k)(*[c[1] for c in key]),
                                       (C.c_void_p * k)(*[c[2] or None for c in key]), (C.c_int64 * k)(*[c[3] for c in key]))
        _, levels, fine, coarse, ns = cached
        L, K = self.n_levels, self.K
        self._keepalive = list(chunks)
        if reduce and _dist_group_active(group):
            packed, on_gpu = self._packed_buffer(group)
            _lib.check(_lib.lib().mlmc_accum_estimate_packed(self._h, k, levels, fine, coarse, ns, _lib.DEVICE, _lib.ptr(packed),
                                                             _lib.DEVICE if on_gpu else _lib.HOST))
            out = unpack_partials(allreduce_partials(packed, group), L, K)
            self._keepalive = []
            return out
        bufs = getattr(self, "_est_out", None)
        if bufs is None:
            arrays = (np.empty(L, dtype=np.int64), np.empty(L, dtype=np.int64),
                      np.empty((L, K), dtype=np.float64), np.empty((L, K), dtype=np.float64))
            bufs = self._est_out = arrays + tuple(_lib.ptr(a) for a in arrays)
        n, n_rm, s, sp, p_n, p_rm, p_s, p_sp = bufs
        _lib.check(_lib.lib().mlmc_accum_estimate(self._h, k, levels, fine, coarse, ns, _lib.DEVICE, p_n, p_rm, p_s, p_sp))
        self._keepalive = []
        return n.copy(), n_rm.copy(), s.copy(), sp.copy()

    def _packed_buffer(self, group):
        """(packed fp64 buffer n | n_rm | s | sp where the collective wants it, lives on the GPU?).  With RCCL the library's
        kernels move onto torch's stream: finalize -> all-reduce is stream-ordered, no host sync in between."""
        import torch
        import torch.distributed as dist
        L, K = self.n_levels, self.K
        on_gpu = dist.get_backend(group) == "nccl"
        if on_gpu and not getattr(_lib, "_on_torch_stream", False):
            torch.cuda.set_device(_lib._bound_device)
            _lib.use_torch_stream()
            _lib._on_torch_stream = True
        dev = torch.device("cuda", _lib._bound_device) if on_gpu else torch.device("cpu")
        packed = getattr(self, "_packed", None)
        if packed is None or packed.device != dev:
            packed = self._packed = torch.empty(2 * L + 2 * L * K, dtype=torch.float64, device=dev)
        return packed, on_gpu

    def finalize(self, group=None, reduce=True):
        """-> n[L], n_rm[L] (int64), s[L, K], sp[L, K] (float64); all-reduced over ranks when distributed."""
        L, K = self.n_levels, self.K
        if reduce and _dist_group_active(group):
            n, n_rm, s, sp = self._finalize_distributed(group)
        else:
            n = np.empty(L, dtype=np.int64)
            n_rm = np.empty(L, dtype=np.int64)
            s = np.empty((L, K), dtype=np.float64)
            sp = np.empty((L, K), dtype=np.float64)
            _lib.check(_lib.lib().mlmc_accum_finalize(self._h, _lib.ptr(n), _lib.ptr(n_rm), _lib.ptr(s), _lib.ptr(sp), _lib.HOST))
        self._keepalive = []
        return n, n_rm, s, sp

    def _finalize_distributed(self, group):
        packed, on_gpu = self._packed_buffer(group)
        _lib.check(_lib.lib().mlmc_accum_finalize_packed(self._h, _lib.ptr(packed), _lib.DEVICE if on_gpu else _lib.HOST))
        return unpack_partials(allreduce_partials(packed, group), self.n_levels, self.K)

    def kernel_time(self):
        """(ms, launches, algorithmic bytes) of the accumulation kernels since create or the previous call
        (needs FLAG_TIMING; returns and clears the totals)."""
        ms = C.c_double()
        launches = C.c_int64()
        nbytes = C.c_int64()
        _lib.check(_lib.lib().mlmc_accum_kernel_time(self._h, C.byref(ms), C.byref(launches), C.byref(nbytes)))
        return ms.value, launches.value, nbytes.value

    def aux_kernel_time(self):
        """(ms, launches, algorithmic bytes) of the auxiliary moments pass of a covariance accumulator whose means come from
        the product linearisation (`mlmc_accum_aux_kernel_time`; zeros when there is none); returns and clears the totals."""
        ms = C.c_double()
        launches = C.c_int64()
        nbytes = C.c_int64()
        _lib.check(_lib.lib().mlmc_accum_aux_kernel_time(self._h, C.byref(ms), C.byref(launches), C.byref(nbytes)))
        return ms.value, launches.value, nbytes.value

    def kernel_flops(self):
        """Matrix-core flops the timed covariance launches executed since create or the previous call (symmetric Gram tiles
        counted once: `mlmc_accum_kernel_flops`); returns and clears the total."""
        fl = C.c_int64()
        _lib.check(_lib.lib().mlmc_accum_kernel_flops(self._h, C.byref(fl)))
        return fl.value


def allreduce_partials(packed, group=None):
    """The only exchange step of the path: ONE all-reduce (sum) of the packed fp64 partials
    [n(L) | n_rm(L) | s(L*K) | sp(L*K)] over the ranks (RCCL over xGMI with backend "nccl"; "gloo" in CPU tests).
    The sample counts travel as doubles (exact below 2^53).  Takes a torch tensor, returns a NumPy array."""
    import torch
    import torch.distributed as dist
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    if not packed.is_cuda:
        return packed.numpy()
    host = _pinned_like(packed)                         # one pinned landing buffer per size, reused
    host.copy_(packed, non_blocking=True)
    if getattr(_lib, "_on_torch_stream", False):        # the library shares torch's stream: its bounded busy poll is the
        _lib.check(_lib.lib().mlmc_synchronize())        # cheaper wait (the runtime's sleeps on the completion interrupt)
    else:
        torch.cuda.current_stream(packed.device).synchronize()
    return host.numpy().copy()


_pinned = {}


def _pinned_like(t):
    import torch
    buf = _pinned.get(t.numel())
    if buf is None:
        buf = torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)
        _pinned[t.numel()] = buf
    return buf


def unpack_partials(packed, L, K):
    n = np.rint(packed[:L]).astype(np.int64)
    n_rm = np.rint(packed[L:2 * L]).astype(np.int64)
    s = packed[2 * L:2 * L + L * K].reshape(L, K).copy()
    sp = packed[2 * L + L * K:].reshape(L, K).copy()
    return n, n_rm, s, sp


def shard_bounds(n, rank, world_size):
    """Contiguous slice [lo, hi) of a level's n samples owned by `rank`."""
    lo = (n * rank) // world_size
    hi = (n * (rank + 1)) // world_size
    return lo, hi


def level_stats(n, s, sp):
    """Per-level mean and variance of the differences (quantity_estimate.py:70-77): mean_l = s / n,
    var_l = (sp - s^2 / n) / (n - 1), inf where n <= 1.  n[L] int, s / sp [L, K]."""
    n = np.asarray(n)
    nf = n.astype(np.float64)[:, None]
    s = np.asarray(s)
    if n.min() > 1:                                   # the usual case: no division by zero to silence, no level to patch
        l_vars = s * s                                # (sp - s * s / n) / (n - 1), same operations in the same order, one temporary
        l_vars /= nf
        np.subtract(sp, l_vars, out=l_vars)
        l_vars /= nf - 1.0
        return s / nf, l_vars
    with np.errstate(all="ignore"):
        l_means = s / nf
        l_vars = (sp - (s * s / nf)) / (nf - 1.0)
    l_vars[n <= 1] = np.inf
    return l_means, l_vars


def moments_from_covariance(s, sp, size, n_comp=1):
    """Level sums of the moments read out of the level sums of their covariance: s / sp [L, n_comp * size * size] (rows
    m * size^2 + i * size + j, the cov_at_bottom layout of mlmc_accum_finalize) -> (s_mom, sp_mom) [L, n_comp * size].
    Row i = 0 of the per-sample outer products (quantity_estimate.py:131-147) is phi_0 phi_j with phi_0 = 1 for every
    moment family of mlmc/moments.py, so  sum_n (f_0 f_j - c_0 c_j) = sum_n d_j  and  sum_n (f_0 f_j - c_0 c_j)^2 =
    sum_n d_j^2: an estimate that has the covariance accumulators of a (quantity, moments_fn) pair needs no second pass
    over the samples for the level means / variances of the moments (estimator.py:56-85)."""
    L = s.shape[0]
    s3 = np.asarray(s).reshape(L, n_comp, size, size)
    sp3 = np.asarray(sp).reshape(L, n_comp, size, size)
    return s3[:, :, 0, :].reshape(L, n_comp * size).copy(), sp3[:, :, 0, :].reshape(L, n_comp * size).copy()


def percentiles(values, q_percent, nan_policy="omit"):
    """np.percentile(values[~isnan(values)], q_percent) evaluated on the device (radix select), bit-identical to NumPy's
    "linear" method.  values: NumPy array or torch CUDA tensor (flattened).
    nan_policy="propagate": NaN for every percentile when a NaN is present, as plain np.percentile(values, ...)."""
    q = _lib.as_f64(np.atleast_1d(q_percent))
    out = np.empty(q.size, dtype=np.float64)
    if isinstance(values, np.ndarray):
        values = _lib.as_f64(values.reshape(-1))
        n = values.size
    else:
        values = values.reshape(-1).contiguous()
        n = values.numel()
    n_valid = C.c_int64()
    _lib.check(_lib.lib().mlmc_percentiles(_lib.ptr(values), int(n), _lib.ptr(q), int(q.size), _lib.ptr(out), C.byref(n_valid),
                                           _lib.mem_kind(values)))
    if nan_policy == "propagate" and n_valid.value < n:
        out[:] = np.nan
    return out
