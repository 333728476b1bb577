"""ctypes binding of libmlmc_hip.so (C ABI declared in include/mlmc_hip.h).

The product has no CPU fallback: if the shared library is missing, or no MI355X is visible when a
compute entry point is first used, an exception is raised.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MLMC_HIP_LIB", os.path.join(_HERE, "libmlmc_hip.so"))   # env override: development builds

ABI_VERSION = 6      # MLMC_ABI_VERSION of include/mlmc_hip.h
LEGENDRE, MONOMIAL, FOURIER, IDENTITY, SPLINE = 0, 1, 2, 3, 4
MODE_MOMENTS, MODE_COV = 0, 1
MODE_MEAN_ONLY = 0x100
HOST, DEVICE = 0, 1
FLAG_TIMING = 1


class MlmcHipError(RuntimeError):
    pass


class BasisDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("size", C.c_int32), ("shift", C.c_double), ("scale", C.c_double),
                ("ref0", C.c_double), ("ref1", C.c_double), ("is_log", C.c_int32), ("is_clip", C.c_int32),
                ("out_size", C.c_int32), ("reserved", C.c_int32), ("matrix", C.POINTER(C.c_double)),
                ("x_lo", C.c_double), ("x_hi", C.c_double)]


class MaxentOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_it", C.c_int32), ("n_intervals", C.c_int32), ("gauss_degree", C.c_int32),
                ("reserved", C.c_int32), ("stab_penalty", C.c_double), ("penalty_coef", C.c_double),
                ("decay_left", C.c_int32), ("decay_right", C.c_int32)]


class MaxentInfo(C.Structure):
    _fields_ = [("nit", C.c_int32), ("success", C.c_int32), ("fun", C.c_double), ("grad_norm", C.c_double),
                ("moment0", C.c_double), ("n_quad", C.c_int32), ("reserved", C.c_int32)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol declared in include/mlmc_hip.h
SIGNATURES = {
    "mlmc_init": (C.c_int, [C.c_int, C.c_int]),
    "mlmc_shutdown": (None, []),
    "mlmc_set_stream": (C.c_int, [_vp]),
    "mlmc_synchronize": (C.c_int, []),
    "mlmc_wait_event": (C.c_int, [_vp]),
    "mlmc_last_error": (C.c_char_p, []),
    "mlmc_abi_version": (C.c_int, []),
    "mlmc_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), _ip]),
    "mlmc_basis_create": (C.c_int, [C.POINTER(BasisDesc), C.POINTER(_vp)]),
    "mlmc_basis_destroy": (None, [_vp]),
    "mlmc_basis_eval": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, _vp, C.c_int]),
    "mlmc_accum_create": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "mlmc_accum_destroy": (None, [_vp]),
    "mlmc_accum_reset": (C.c_int, [_vp]),
    "mlmc_accum_push": (C.c_int, [_vp, C.c_int32, _vp, _vp, C.c_int64, C.c_int]),
    "mlmc_accum_finalize": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int]),
    "mlmc_accum_finalize_packed": (C.c_int, [_vp, _vp, C.c_int]),
    "mlmc_accum_estimate": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "mlmc_accum_estimate_packed": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int]),
    "mlmc_accum_kernel_time": (C.c_int, [_vp, _dp, _ip, _ip]),
    "mlmc_accum_kernel_flops": (C.c_int, [_vp, _ip]),
    "mlmc_accum_aux_kernel_time": (C.c_int, [_vp, _dp, _ip, _ip]),
    "mlmc_linearization_table": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int64]),
    "mlmc_percentiles": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, _vp, _ip, C.c_int]),
    "mlmc_maxent_solve": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_double, C.c_double, C.POINTER(MaxentOpts), _vp,
                                    C.c_int32, _vp, _vp, _vp, C.POINTER(MaxentInfo)]),
    "mlmc_density_eval": (C.c_int, [_vp, _vp, _vp, C.c_int32, _vp, C.c_int64, _vp, C.c_int]),
    "mlmc_density_integrate": (C.c_int, [_vp, _vp, _vp, C.c_int32, _vp, _vp, C.c_int64, C.c_int32, _vp]),
    "mlmc_expr_create": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "mlmc_expr_destroy": (None, [_vp]),
    "mlmc_expr_eval": (C.c_int, [_vp, _vp, C.c_int32, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _ip]),
    "mlmc_expr_state": (C.c_int, [_vp, C.POINTER(C.c_int32), _ip]),
    "mlmc_expr_kernel_time": (C.c_int, [_vp, _dp, _ip, _ip]),
    "mlmc_synth_generate": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_int32, _vp, _vp]),
    "mlmc_synth_seeds": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, _vp]),
    "mlmc_subsample_gather": (C.c_int, [_vp, _vp, C.c_int32, C.c_int64, C.c_int64, C.c_uint64, _vp, _vp]),
}

_lock = threading.Lock()
_lib = None
_bound_device = None


def load():
    """Load the shared library (no GPU needed) and declare the prototypes."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise MlmcHipError(
                    "libmlmc_hip.so not found at {} -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C mlmc_amd/csrc`; mlmc_amd has no CPU fallback".format(LIB_PATH))
            try:
                # PyTorch bundles its own HIP runtime under the same soname (libamdhip64.so.7); import it first so
                # that the process holds ONE runtime and torch device pointers are valid in our kernels.
                import torch  # noqa: F401
            except ImportError:
                pass
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            if lib.mlmc_abi_version() != ABI_VERSION:
                raise MlmcHipError("{} has ABI version {}, this package binds version {}: rebuild it "
                                   "(make -C mlmc_amd/csrc)".format(LIB_PATH, lib.mlmc_abi_version(), ABI_VERSION))
            _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise MlmcHipError(load().mlmc_last_error().decode("utf-8", "replace"))


def init(device=None, flags=None):
    """Bind the process to one GPU (LOCAL_RANK by default). Idempotent."""
    global _bound_device
    lib = load()
    if device is None:
        device = _bound_device if _bound_device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    if flags is None:
        flags = FLAG_TIMING if os.environ.get("MLMC_HIP_TIMING") else 0
    check(lib.mlmc_init(int(device), int(flags)))
    _bound_device = int(device)
    return lib


def lib():
    """The loaded library with a device bound (binds LOCAL_RANK / device 0 on first use)."""
    if _bound_device is None:
        return init()
    return _lib


def use_torch_stream():
    """Enqueue all kernels on torch's current CUDA/HIP stream (the one torch.distributed collectives use)."""
    import torch
    check(lib().mlmc_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def device_info():
    l = lib()
    name = C.create_string_buffer(256)
    n_cu, wave = C.c_int(), C.c_int()
    hbm = C.c_int64()
    check(l.mlmc_device_info(name, 256, C.byref(n_cu), C.byref(wave), C.byref(hbm)))
    return dict(name=name.value.decode(), n_cu=n_cu.value, wave_size=wave.value, hbm_bytes=hbm.value, device=_bound_device)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    """void* of a C-contiguous numpy array, a torch tensor (host or device) or None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))


def mem_kind(a):
    if isinstance(a, np.ndarray):
        return HOST
    if hasattr(a, "is_cuda"):
        return DEVICE if a.is_cuda else HOST
    raise TypeError(type(a))
