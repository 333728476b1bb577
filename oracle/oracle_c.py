"""TEST INFRASTRUCTURE ONLY -- ctypes face of oracle/oracle_c.c (the C restatement of the hot loop)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


class _Basis(C.Structure):
    _fields_ = [("kind", C.c_int), ("size", C.c_int), ("shift", C.c_double), ("scale", C.c_double),
                ("ref0", C.c_double), ("ref1", C.c_double), ("is_log", C.c_int), ("is_clip", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        for name in ("oracle_moments_level", "oracle_cov_level"):
            fn = getattr(_lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(_Basis), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                           C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    return _lib


def _run(fn_name, b, fine, coarse, K):
    lib = _load()
    cb = _Basis(b.kind, b.size, b.shift, b.scale, b.ref_domain[0], b.ref_domain[1], int(b.log), int(b.safe_eval))
    fine = np.ascontiguousarray(fine, dtype=np.float64)
    coarse = None if coarse is None else np.ascontiguousarray(coarse, dtype=np.float64)
    s = np.zeros(K)
    sp = np.zeros(K)
    nk, nr = C.c_int64(), C.c_int64()
    getattr(lib, fn_name)(C.byref(cb), fine.ctypes.data, None if coarse is None else coarse.ctypes.data, fine.size,
                          s.ctypes.data, sp.ctypes.data, C.byref(nk), C.byref(nr))
    return nk.value, nr.value, s, sp


def moments_level(b, fine, coarse):
    """-> n_keep, n_rm, s[R], sp[R] for one level of a scalar quantity (oracle_np.Basis b, no transform matrix)."""
    return _run("oracle_moments_level", b, fine, coarse, b.size)


def cov_level(b, fine, coarse):
    return _run("oracle_cov_level", b, fine, coarse, b.size * b.size)
