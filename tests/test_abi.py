"""The C-ABI library loads and exports every symbol include/alsmi.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

from alsmi import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'alsmi.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(alsmi_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 10
    for name in names:
        assert hasattr(lib, name), name


def test_binding_covers_header():
    _lib.load()
    assert set(_lib.EXPORTS) == set(declared_symbols())


def test_host_only_entry_points():
    lib = _lib.load()
    assert lib.alsmi_version() >= 100
    assert [lib.alsmi_padded_rank(r) for r in (0, 1, 10, 16, 17, 33, 64, 65, 128, 129)] == \
        [0, 16, 16, 16, 32, 64, 64, 128, 128, 0]
    assert lib.alsmi_sddmm_workspace_bytes() >= 8
    # argument errors are reported through the status code and the message
    st = lib.alsmi_sddmm(7, 4, 0, None, None, None, None, None, None, None, None, None)
    assert st == -1
    assert b'dtype' in lib.alsmi_last_error()
