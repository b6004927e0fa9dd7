"""ctypes wrapper of oracle/als_oracle.c (TEST INFRASTRUCTURE, see als_oracle.py)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(path):
            raise ImportError('oracle/liboracle.so missing: run `make -C oracle`')
        lib = ctypes.CDLL(path)
        vp, i64, i32, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
        lib.oracle_half_sweep.argtypes = [vp, vp, vp, vp, i32, dbl, i64, i64, vp, i32]
        lib.oracle_half_sweep.restype = None
        lib.oracle_sse.argtypes = [vp, vp, vp, vp, vp, i32, i64, i64, i32]
        lib.oracle_sse.restype = dbl
        lib.oracle_imc_eval.argtypes = [i32, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32]
        lib.oracle_imc_eval.restype = dbl
        lib.oracle_max_threads.restype = i32
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def max_threads():
    return int(load().oracle_max_threads())


def half_sweep(indptr, indices, vals, src_rows, alpha, row_begin=0, row_end=None, threads=0):
    """src_rows: [n_src x k] row-major.  Returns [(row_end-row_begin) x k]."""
    indptr, indices, vals = _c(indptr, np.int64), _c(indices, np.int32), _c(vals, np.float64)
    src_rows = _c(src_rows, np.float64)
    if row_end is None:
        row_end = indptr.size - 1
    k = src_rows.shape[1]
    out = np.zeros((row_end - row_begin, k))
    load().oracle_half_sweep(_p(indptr), _p(indices), _p(vals), _p(src_rows), k, float(alpha),
                             int(row_begin), int(row_end), _p(out), int(threads))
    return out


def sse(indptr, indices, vals, U_rows, V_rows, row_begin=0, row_end=None, threads=0):
    indptr, indices, vals = _c(indptr, np.int64), _c(indices, np.int32), _c(vals, np.float64)
    U_rows, V_rows = _c(U_rows, np.float64), _c(V_rows, np.float64)
    if row_end is None:
        row_end = indptr.size - 1
    return float(load().oracle_sse(_p(indptr), _p(indices), _p(vals), _p(U_rows), _p(V_rows),
                                   U_rows.shape[1], int(row_begin), int(row_end), int(threads)))


def imc_eval(mode, indptr, items, ratings, X, Y, M, threads=0):
    """Returns (X^T W [p x q], sse) — see oracle_imc_eval."""
    indptr, items = _c(indptr, np.int64), _c(items, np.int32)
    ratings, X, Y, M = (_c(a, np.float64) for a in (ratings, X, Y, M))
    p, q = M.shape
    out = np.zeros((p, q))
    s = load().oracle_imc_eval(int(mode), indptr.size - 1, p, q, _p(indptr), _p(items),
                               _p(ratings), _p(X), _p(Y), _p(M), _p(out), int(threads))
    return out, float(s)
