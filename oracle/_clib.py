"""ORACLE (test infrastructure only): build + load the plain-C NN restatement."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmmk_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "nn_search.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "_build/libmmk_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        f32p = ctypes.POINTER(ctypes.c_float)
        i32p = ctypes.POINTER(ctypes.c_int32)
        _lib.mmk_oracle_nn_search.argtypes = [f32p, f32p, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, i32p, f32p]
        _lib.mmk_oracle_nn_search.restype = None
        _lib.mmk_oracle_transform.argtypes = [f32p, f32p, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, f32p]
        _lib.mmk_oracle_transform.restype = None
        f64p = ctypes.POINTER(ctypes.c_double)
        _lib.mmk_oracle_blend4_f64.argtypes = [f64p] * 8 + [ctypes.c_long, f64p]
        _lib.mmk_oracle_blend4_f64.restype = None
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def nn_search(p, t):
    """p: (B,N,d) f32, t: (B,M,d) f32 -> idx (B,N) int32, d2 (B,N) f32."""
    p, pp = _f32(p)
    t, tp = _f32(t)
    B, N, d = p.shape
    M = t.shape[1]
    assert t.shape == (B, M, d) and d in (2, 3)
    idx = np.empty((B, N), dtype=np.int32)
    d2 = np.empty((B, N), dtype=np.float32)
    lib().mmk_oracle_nn_search(pp, tp, B, N, M, d,
                               idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                               d2.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return idx, d2


def transform(src, T, d):
    """src: (B,N,3) f32, T: (B,4,4) f32 -> (B,N,d) f32 (stage I1)."""
    src, sp = _f32(src)
    T, Tp = _f32(np.asarray(T).reshape(-1, 16))
    B, N, _ = src.shape
    out = np.empty((B, N, d), dtype=np.float32)
    lib().mmk_oracle_transform(sp, Tp, B, N, d, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def blend4_f64(taps, weights):
    """fma(t3, w3, fma(t2, w2, fma(t1, w1, t0 * w0))) element-wise in fp64 (see nn_search.c)."""
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in list(taps) + list(weights)]
    out = np.empty_like(arrs[0])
    ptrs = [a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) for a in arrs]
    lib().mmk_oracle_blend4_f64(*ptrs, out.size, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out
