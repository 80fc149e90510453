"""The C-ABI library loads without a GPU and exports exactly what include/tdg.h declares."""
import os
import re
import subprocess

from conftest import pkg, ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'tdg.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return set(re.findall(r'\b(tdg_[a-z0-9_]+)\s*\(', text))


def test_header_python_and_library_agree():
    L = pkg('_lib')
    lib = L.load()                                   # raises if the .so is missing: there is no CPU fallback
    decl = declared_symbols()
    assert decl == set(L.SIGNATURES), decl ^ set(L.SIGNATURES)
    out = subprocess.check_output(['nm', '-D', '--defined-only', L.LIB_PATH]).decode()
    exported = set(re.findall(r' T (tdg_[a-z0-9_]+)', out))
    assert decl <= exported, decl - exported
    assert lib.tdg_version() >= 100


def test_argument_errors_are_reported_not_thrown():
    """Status + tdg_last_error(), never an exception across the boundary (no GPU work is enqueued)."""
    import ctypes as C
    L = pkg('_lib')
    lib = L.load()
    d = L.ConvDesc(1, 8, 8, 4, 4, 4, 4, 8, 8, 5, 5, 3, 1, 1, 0)          # stride 3: unsupported
    assert lib.tdg_packed_filter_fwd_bytes(C.byref(d)) == 0
    assert b'stride' in lib.tdg_last_error()
    rc = lib.tdg_adam_step(None, None, None, None, 16, 0.1, 0.9, 0.999, 1e-8, 1.0, None)
    assert rc == -1 and b'tdg_adam_step' in lib.tdg_last_error()
    d = L.ConvDesc(1, 8, 8, 4, 2, 4, 4, 8, 8, 5, 5, 2, 1, 1, 0)          # channel stride < channels
    assert lib.tdg_packed_filter_bwd_bytes(C.byref(d)) == 0
    # sizes for a valid descriptor: K = 25*8 = 200 f32 -> padded to 224 per row of the 16 output channels
    d = L.ConvDesc(2, 8, 8, 8, 8, 4, 4, 16, 16, 5, 5, 2, 1, 1, 0)
    assert lib.tdg_packed_filter_fwd_bytes(C.byref(d)) == 16 * 224 * 4
    assert lib.tdg_conv2d_bwd_filter_workspace_bytes(C.byref(d), 2) >= 25 * 8 * 16 * 4


def test_missing_library_fails_loudly(monkeypatch):
    L = pkg('_lib')
    import pytest
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'LIB_PATH', '/nonexistent/lib3dgan_hip.so')
    with pytest.raises(L.TdgError):
        L.load()
