"""Deterministic float64 products for the handful of tiny host-side matrices.

The reference builds its per-frame constants (look-at, MVP, frustum planes) with NumPy
``@``, i.e. whatever FMA order the BLAS behind NumPy uses on the machine at hand.  The
device kernels need those constants bit-for-bit identical to the ones the golden frames
were rendered with, on any host, so the products are spelled out here as explicit
fused-multiply-add chains (ascending k, first term a plain rounded product): the order
measured for the reference's stack, SURVEY.md Appendix D.
"""
import ctypes
import ctypes.util
from fractions import Fraction

import numpy as np


def _load_fma():
    try:
        libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        f = libm.fma
        f.restype = ctypes.c_double
        f.argtypes = (ctypes.c_double, ctypes.c_double, ctypes.c_double)
        if f(0.1, 10.0, -1.0) == 5.551115123125783e-17:      # really fused
            return f
    except (OSError, AttributeError):
        pass

    def slow(a, b, c):                                      # exact rational, one rounding
        try:
            return float(Fraction(a) * Fraction(b) + Fraction(c))
        except (ValueError, OverflowError):
            return a * b + c
    return slow


fma = _load_fma()


def dot_chain(a, b):
    """``sum_k a[k]*b[k]`` as ``rn(a0*b0)`` followed by ascending ``fma`` steps."""
    a = [float(x) for x in a]
    b = [float(x) for x in b]
    acc = a[0] * b[0]
    for k in range(1, len(a)):
        acc = fma(a[k], b[k], acc)
    return acc


_native_matmul = None        # mr_host_matmul_chain of the HIP library (plain host C: the same chains, ~100x faster), or False


def _fast_matmul():
    """The library's host helper, if the library is built (it loads without a GPU); else the loops below."""
    global _native_matmul
    if _native_matmul is None:
        try:
            from ._native import load_library
            _native_matmul = load_library().mr_host_matmul_chain
        except Exception:           # not built / not loadable here: the pure-Python chains give the same bits
            _native_matmul = False
    return _native_matmul


_native_camera = None        # mr_host_camera_constants of the HIP library, or False


def camera_constants():
    """The library's one-call look-at / MVP / frustum planes (``core.TransformationMatrixMixin``), if it is built."""
    global _native_camera
    if _native_camera is None:
        try:
            from ._native import load_library
            _native_camera = load_library().mr_host_camera_constants
        except Exception:
            _native_camera = False
    return _native_camera


def matmul_chain(a, b):
    """(M,K) @ (K,P) in float64 with every output element an ascending-k fma chain."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    squeeze = a.ndim == 1
    a2 = np.atleast_2d(a)
    m, k = a2.shape
    k2, p = b.shape
    if k != k2:
        raise ValueError(f"matmul: shapes {a.shape} and {b.shape} not aligned")
    out = np.empty((m, p), dtype=np.float64)
    fast = _fast_matmul()
    if fast:
        a2, b = np.ascontiguousarray(a2), np.ascontiguousarray(b)
        fast(a2.ctypes.data, b.ctypes.data, out.ctypes.data, m, k, p)
        return out[0] if squeeze else out
    for i in range(m):
        row = a2[i]
        for j in range(p):
            out[i, j] = dot_chain(row, b[:, j])
    return out[0] if squeeze else out
