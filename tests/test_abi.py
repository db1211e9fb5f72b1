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
    assert lib.icv_abi_version() == native.ABI_VERSION == 5
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "icvideo.h")).read()
    assert f"#define ICV_ABI_VERSION {native.ABI_VERSION}" in hdr, "binding and header disagree on the ABI version"
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


def test_context_entry_points_validate_without_gpu():
    """icv_dit_* / icv_comm_* argument checks return before any HIP or RCCL call."""
    lib = native.lib()
    h = ctypes.c_void_p()
    bad = native.DitConfig(dim=100, ffn_dim=64, heads=1, layers=1, n_tok=4, tok0=0, T=1, Hp=2, Wp=2, k_patch=64, out_cols=64, eps=1e-6)
    assert lib.icv_dit_create(ctypes.byref(bad), ctypes.byref(h)) != 0 and b"heads * 128" in lib.icv_last_error()
    shard = native.DitConfig(dim=256, ffn_dim=512, heads=2, layers=2, n_tok=8, tok0=3, T=1, Hp=2, Wp=4, k_patch=64, out_cols=64, eps=1e-6)
    assert lib.icv_dit_create(ctypes.byref(shard), ctypes.byref(h)) != 0 and b"token shard outside" in lib.icv_last_error()
    ok = native.DitConfig(dim=256, ffn_dim=512, heads=2, layers=2, n_tok=8, tok0=0, T=1, Hp=2, Wp=4, k_patch=64, out_cols=64, eps=1e-6)
    assert lib.icv_dit_create(ctypes.byref(ok), ctypes.byref(h)) == 0 and h.value
    try:
        assert lib.icv_dit_bind(h, b"wqkv", 2, 8) != 0 and b"out of range" in lib.icv_last_error()
        assert lib.icv_dit_bind(h, b"nonsense", -1, 8) != 0 and b"unknown tensor" in lib.icv_last_error()
        assert lib.icv_dit_bind(h, b"nonsense", 0, 8) != 0 and b"unknown per-layer tensor" in lib.icv_last_error()
        assert lib.icv_dit_bind(h, b"wqkv", 0, 8) == 0
        # forward with unbound tensors is refused before anything is enqueued
        rc = lib.icv_dit_forward(h, 8, 16, 4, 8, 8, 8, 8, 8, 1, 0, None, None, 0, 0, None, 8, -1, 0, 1.0, None)
        assert rc != 0 and b"bind" in lib.icv_last_error()
        assert lib.icv_dit_forward(h, 8, 16, 4, 8, 8, 8, 8, 8, 1, 0, None, None, 0, 0, None, 8, 5, 0, 1.0, None) != 0
        assert b"num_layers" in lib.icv_last_error()
        ms, n = ctypes.c_double(-1.0), ctypes.c_int64(-1)
        assert lib.icv_dit_profile(h, 1) == 0 and lib.icv_dit_profile_read(h, ctypes.byref(ms), ctypes.byref(n)) == 0
        assert (ms.value, n.value) == (0.0, 0)
    finally:
        lib.icv_dit_destroy(h)
    assert lib.icv_comm_create(b"\0" * native.COMM_ID_BYTES, 2, 2, ctypes.byref(h)) != 0 and b"bad (rank, world)" in lib.icv_last_error()
    assert lib.icv_allgather_kv(None, 8, 8, 4, 16, None) != 0 and b"null argument" in lib.icv_last_error()


def test_header_is_plain_c(tmp_path):
    """include/icvideo.h is the drop-in boundary: it must compile as C99 (no C++ / torch types) and as C++."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no gcc")
    src = tmp_path / "h.c"
    src.write_text('#include "icvideo.h"\nint main(void) { int (*f)(void) = icv_abi_version; return f == 0 ? 1 : 0; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)], check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, "-x", "c++", str(src)], check=True)
