"""The C-ABI shared library loads on a CPU-only box and exports exactly what include/icvideo.h declares;
the ctypes table in infinicube_amd/native.py covers every declared symbol (no compute calls here)."""
import ctypes
import os
import re

from infinicube_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "icvideo.h"), encoding="utf-8").read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(icv_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert len(names) >= 12
    assert sorted(native.SIGNATURES) == names, "native.SIGNATURES must list every symbol of icvideo.h"
    lib = ctypes.CDLL(native.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"libicvideo.so does not export {n}"


def test_lib_loads_and_reports_version():
    lib = native.lib()
    assert lib.icv_abi_version() == 1
    assert isinstance(lib.icv_last_error(), bytes)


def test_argument_validation_without_gpu():
    """Shape checks run on the host before any launch, so they are testable without a GPU."""
    lib = native.lib()
    rc = lib.icv_gemm_bf16(1, 64, 1, 64, None, 8, 8, 100, 0, 1, 8, 8, 0, None, 0, None, None)   # K=100
    assert rc != 0 and b"multiple of 64" in lib.icv_last_error()
    rc = lib.icv_ln_modulate(1, 100, None, None, None, None, 1, 100, 4, 100, 1e-6, None)           # d=100
    assert rc != 0 and b"multiple of 256" in lib.icv_last_error()
    rc = lib.icv_attention_fwd(1, 128, 1, 128, 1, 128, 1, 128, 0, 5, 1, 0.1, None)                  # Sq=0
    assert rc != 0 and b"empty problem" in lib.icv_last_error()
