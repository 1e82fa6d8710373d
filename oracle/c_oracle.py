"""ctypes wrapper of oracle/scatter_ref.c (TEST INFRASTRUCTURE ONLY: see hgnn_oracle.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def _lib():
    if not os.path.exists(_SO):
        build()
    return ctypes.CDLL(_SO)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def scatter_add(src, index, dim_size, weight=None):
    src = np.ascontiguousarray(src, np.float32)
    index = np.ascontiguousarray(index, np.int64)
    w = np.ascontiguousarray(weight, np.float32).reshape(-1) if weight is not None else None
    M, F = src.shape
    out = np.empty((dim_size, F), np.float32)
    rc = _lib().oracle_scatter_add_f32(_p(src, ctypes.c_float), _p(index, ctypes.c_int64), _p(w, ctypes.c_float),
                                       ctypes.c_int64(M), ctypes.c_int64(F), _p(out, ctypes.c_float),
                                       ctypes.c_int64(dim_size))
    if rc:
        raise IndexError("index out of range")
    return out


def gather_scale_scatter(X, gather, dst, dim_size, weight=None, row_scale=None):
    X = np.ascontiguousarray(X, np.float32)
    gather = np.ascontiguousarray(gather, np.int64)
    dst = np.ascontiguousarray(dst, np.int64)
    w = np.ascontiguousarray(weight, np.float32).reshape(-1) if weight is not None else None
    rs = np.ascontiguousarray(row_scale, np.float32).reshape(-1) if row_scale is not None else None
    out = np.empty((dim_size, X.shape[1]), np.float32)
    rc = _lib().oracle_gather_scale_scatter_f32(
        _p(X, ctypes.c_float), ctypes.c_int64(X.shape[0]), _p(gather, ctypes.c_int64), _p(dst, ctypes.c_int64),
        _p(w, ctypes.c_float), _p(rs, ctypes.c_float), ctypes.c_int64(len(dst)), ctypes.c_int64(X.shape[1]),
        _p(out, ctypes.c_float), ctypes.c_int64(dim_size))
    if rc:
        raise IndexError("index out of range")
    return out
