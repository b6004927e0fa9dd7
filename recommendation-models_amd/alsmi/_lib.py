"""ctypes binding of libalsmi.so (the HIP kernels behind every estimator).

There is no CPU fallback: if the shared library is missing or a call fails,
an exception is raised.  Declarations mirror ``include/alsmi.h``.
"""
import ctypes
import os

F32 = 0
F64 = 1

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ALSMI_LIB', os.path.join(_HERE, 'libalsmi.so'))

_lib = None


class AlsmiError(RuntimeError):
    """A libalsmi entry point returned a non-zero status."""


def load():
    """Load libalsmi.so once; raise ImportError with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libalsmi.so not found at %s — build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` or '
            '`make -C recommendation-models_amd/csrc` (needs hipcc, gfx950)' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    c_int, c_i64, c_vp, c_dbl = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_double
    sigs = {
        'alsmi_version': (c_int, []),
        'alsmi_last_error': (ctypes.c_char_p, []),
        'alsmi_padded_rank': (c_int, [c_int]),
        'alsmi_plan_create': (c_int, [ctypes.POINTER(c_vp), c_vp, c_i64, c_i64, c_i64, c_int,
                                      c_int, c_int]),
        'alsmi_plan_destroy': (c_int, [c_vp]),
        'alsmi_plan_info': (c_int, [c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64),
                                    ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
        'alsmi_half_sweep': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_dbl, c_vp]),
        'alsmi_half_sweep_err': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_dbl,
                                         c_vp, c_vp]),
        'alsmi_normal_equations': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_dbl,
                                           c_vp]),
        'alsmi_sddmm_workspace_bytes': (ctypes.c_size_t, []),
        'alsmi_sddmm': (c_int, [c_int, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                c_vp, c_vp]),
        'alsmi_sddmm_noalloc': (c_int, [c_int, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_vp, c_vp]),
        'alsmi_topk_workspace_bytes': (ctypes.c_size_t, [c_int, c_i64, c_i64]),
        'alsmi_topk_items': (c_int, [c_int, c_int, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_vp,
                                     c_vp, c_vp, c_vp]),
        'alsmi_expand_rows': (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp]),
        'alsmi_sparse_workspace_bytes': (ctypes.c_size_t, [c_i64]),
        'alsmi_build_csr': (c_int, [c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                    c_vp, c_vp, c_vp, c_vp]),
        'alsmi_csr_to_csc': (c_int, [c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                     c_vp, c_vp, c_vp]),
        'alsmi_imc_padded_width': (c_int, [c_int]),
        'alsmi_imc_workspace_bytes': (ctypes.c_size_t, [c_int, c_i64, c_i64, c_int, c_int]),
        'alsmi_imc_eval': (c_int, [c_int, c_int, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp,
                                   c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
        'alsmi_imc_project': (c_int, [c_int, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTS = ('alsmi_version', 'alsmi_last_error', 'alsmi_padded_rank', 'alsmi_plan_create',
           'alsmi_plan_destroy', 'alsmi_plan_info', 'alsmi_half_sweep',
           'alsmi_half_sweep_err',
           'alsmi_normal_equations', 'alsmi_sddmm_workspace_bytes', 'alsmi_sddmm',
           'alsmi_sddmm_noalloc', 'alsmi_topk_workspace_bytes', 'alsmi_topk_items',
           'alsmi_expand_rows', 'alsmi_sparse_workspace_bytes', 'alsmi_build_csr',
           'alsmi_csr_to_csc', 'alsmi_imc_padded_width', 'alsmi_imc_workspace_bytes',
           'alsmi_imc_eval', 'alsmi_imc_project')


def check(status, what):
    if status != 0:
        msg = load().alsmi_last_error().decode('utf-8', 'replace')
        raise AlsmiError('%s failed (%d): %s' % (what, status, msg))


def dtype_code(np_dtype):
    import numpy as np
    dt = np.dtype(np_dtype)
    if dt == np.float32:
        return F32
    if dt == np.float64:
        return F64
    raise ValueError("dtype must be 'float32' or 'float64', got %r" % (np_dtype,))
