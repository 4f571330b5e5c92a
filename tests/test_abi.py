"""The C-ABI libraries load and export every symbol the headers declare (no compute, no GPU)."""
import ctypes as C
import os
import re

from amplisolve_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ampli_[A-Za-z0-9_]+)\s*\(", txt)))


def test_hip_header_symbols_exported():
    names = declared("amplisolve_hip.h")
    assert len(names) >= 25
    lib = C.CDLL(_lib.HIP_LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/amplisolve_hip.h but not exported"
    assert set(names) == set(_lib.HIP_SYMBOLS), set(names) ^ set(_lib.HIP_SYMBOLS)


def test_host_header_symbols_exported():
    names = declared("amplisolve_host.h")
    lib = C.CDLL(_lib.HOST_LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/amplisolve_host.h but not exported"
    assert set(names) == set(_lib.HOST_SYMBOLS), set(names) ^ set(_lib.HOST_SYMBOLS)


def test_abi_version_and_strerror():
    lib = _lib.hip_lib()
    assert lib.ampli_abi_version() == 5  # AMPLI_ABI_VERSION (5: ampli_error_sums_inorder, round 6)
    assert lib.ampli_strerror(0) == b"ok"
    assert b"HIP" in lib.ampli_strerror(-2)


def test_acc_layout_is_plane_aligned():
    lib = _lib.hip_lib()
    P = 1000
    n = lib.ampli_acc_bytes(P)
    assert n >= P * (64 + 64 + 16 + 4 + 16 * 4)
    t = _lib.AccTable()
    base = 1 << 20
    assert lib.ampli_acc_bind(C.c_void_p(base), P, C.byref(t)) == 0
    ptrs = [t.snt, t.srd, t.cnt, t.nrec, t.gm_n, t.gm_first_af, t.gm_rest, t.gm_first]  # buffer order
    assert all(p % 256 == 0 for p in ptrs) and ptrs == sorted(ptrs) and ptrs[-1] + 16 * P <= base + n
    a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    assert lib.ampli_acc_regions(P, C.byref(a), C.byref(b), C.byref(c)) == 0
    assert b.value == t.gm_n - base and b.value + c.value == t.gm_first - base and a.value == t.gm_first_af - base


def test_no_gpu_means_loud_failure():
    """Without a device the product must refuse to compute rather than fall back."""
    import torch

    if torch.cuda.is_available():
        return
    lib = _lib.hip_lib()
    h = C.c_void_p()
    assert lib.ampli_ctx_create(0, None, C.byref(h)) == -2
    import pytest

    from amplisolve_amd import AmpliError, Context

    with pytest.raises(AmpliError):
        Context(0)
