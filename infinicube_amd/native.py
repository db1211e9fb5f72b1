"""ctypes binding of libicvideo.so (C ABI declared in include/icvideo.h).

This is the stub a maintainer of the reference would add next to
``infinicube/videogen/inference.py``.  There is NO fallback: if the shared
library is missing or fails to load, importing ``lib()`` raises — the product path never routes
through PyTorch eager or the oracle.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# ICV_LIB_PATH: load another build of the same ABI (e.g. csrc/build/libicvideo_experiments.so, which adds the A/B kernels)
LIB_PATH = os.environ.get("ICV_LIB_PATH") or os.path.join(_HERE, "csrc", "libicvideo.so")

EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32, EPI_F32 = 0, 1, 2, 3
ACT_NONE, ACT_SILU = 0, 1

# name -> (restype, argtypes); must list EVERY symbol include/icvideo.h declares
# (tests/test_abi.py parses the header and checks this table and the .so against it).
_P, _I, _F = c_void_p, c_int64, c_float
SIGNATURES = {
    "icv_abi_version": (c_int, []),
    "icv_last_error": (c_char_p, []),
    "icv_device_info": (c_int, [c_int, ctypes.POINTER(c_int64)]),
    "icv_set_option": (c_int, [c_char_p, c_int]),
    "icv_gemm_bf16": (c_int, [_P, _I, _P, _I, _P, _I, _I, _I, c_int, _P, _I, _I, _I, _P, _I, _P, _P]),
    "icv_gemv_f32": (c_int, [_P, _P, _P, _P, _I, _I, _I, c_int, c_int, _P]),
    "icv_sinusoidal_embedding": (c_int, [c_double, _I, _P, _P]),
    "icv_bcast_add_f32": (c_int, [_P, _P, _P, _I, _I, _P]),
    "icv_ln_modulate": (c_int, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "icv_rmsnorm_rope": (c_int, [_P, _P, _P, _P, _I, _I, _I, _F, _P, _I, _I, _I, _I, _P]),
    "icv_quantize_rows_fp8": (c_int, [_P, c_int, _I, _I, _I, _P, _I, _P, _P]),
    "icv_ln_modulate_fp8": (c_int, [_P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _F, _P]),
    "icv_gemm_fp8": (c_int, [_P, _I, _P, _P, _I, _P, _P, _I, _I, _I, c_int, _P, _I, _I, _I, _P, _I, _P, _P]),
    "icv_attention_fwd": (c_int, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _F, _P]),
    "icv_attention_fwd_add": (c_int, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _F, _P]),
    "icv_attention_fp8_vt_bytes": (ctypes.c_int64, [_I, _I]),
    "icv_attention_fp8_prepare": (c_int, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P]),
    "icv_attention_fp8_fwd": (c_int, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "icv_attention_fp8_fwd_chunk": (c_int, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _P, _I, _I, _I, c_int, c_int, _P]),
    "icv_attention_fp8_blob_bytes": (ctypes.c_int64, [_I, _I]),
    "icv_attention_fp8_kv_amax": (c_int, [_P, _I, _P, _I, _I, _I, _P, _P]),
    "icv_attention_fp8_quantize_kv": (c_int, [_P, _I, _P, _I, _I, _I, _P, _P, _P]),
    "icv_attention_fp8_fwd_pieces": (c_int, [_P, _I, _P, _I, _I, _P, _P, _I, _P, _I, _P, _I, _I, c_int, c_int, _P]),
    "icv_attention_fp8_fwd_pieces_gated": (c_int, [_P, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, c_int, c_int, _P]),
    "icv_attention_trace": (c_int, [_P, _I]),
    "icv_attention_fwd_chunk": (c_int, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _F, c_int, c_int, _P]),
    "icv_patchify": (c_int, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P]),
    "icv_unpatchify_cfg_euler": (c_int, [_P, _P, _P, _P, _I, _F, _F, _I, _I, _I, _I, _I, _I, c_int, _P]),
    "icv_cast_f32_to_bf16": (c_int, [_P, _P, _I, _P]),
    "icv_semantic_to_color": (c_int, [_P, _I, _P, c_int, _P, _P, _P]),
    "icv_instance_overlay_u8": (c_int, [_P, _P, _I, _P, _P, _P]),
    "icv_voxel_scatter": (c_int, [_P, _I, ctypes.POINTER(c_int), ctypes.POINTER(c_int), _P, _P, _P]),
    "icv_voxel_raycast": (c_int, [_P, _P, ctypes.POINTER(c_int), ctypes.POINTER(c_float), ctypes.POINTER(c_float), _P, _P, _I, _I,
                                  _F, _F, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "icv_depth_to_u16": (c_int, [_P, _I, _F, _P, _P]),
    "icv_rmsnorm_act_rows": (c_int, [_P, _P, _P, _I, _I, _F, _F, c_int, _P]),
    "icv_rmsnorm_act_volume": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, c_int, _P]),
    "icv_conv3d_ndhwc": (c_int, [_P, _I, _I, _I, _P, _P, ctypes.POINTER(c_int64), _I, _I, _I, _I, _I, _P, _I, _P, _I, _P]),
    "icv_coord_valid_mask": (c_int, [_P, ctypes.POINTER(c_float), _P, _I, _I, _I, _P, _P]),
    "icv_coord_gather_points": (c_int, [_P, ctypes.POINTER(c_float), _P, _I, _I, _I, _P, _I, _P, _P]),
    "icv_coord_normalize": (c_int, [_P, ctypes.POINTER(c_float), _P, _I, _I, _I, ctypes.POINTER(c_float),
                                    ctypes.POINTER(c_float), c_int, _P, _P, _P]),
}

class DitConfig(ctypes.Structure):
    """icv_dit_config of include/icvideo.h."""
    _fields_ = [(n, c_int64) for n in ("dim", "ffn_dim", "heads", "layers", "n_tok", "tok0", "T", "Hp", "Wp", "k_patch", "out_cols")] + [("eps", c_float)]


SIGNATURES.update({
    "icv_dit_create": (c_int, [ctypes.POINTER(DitConfig), ctypes.POINTER(c_void_p)]),
    "icv_dit_destroy": (None, [c_void_p]),
    "icv_dit_bind": (c_int, [c_void_p, c_char_p, _I, _P]),
    "icv_dit_set_fp8": (c_int, [c_void_p, c_int]),
    "icv_dit_set_seqpar": (c_int, [c_void_p, c_void_p, _I, _I, ctypes.POINTER(c_int64), _P]),
    "icv_dit_forward": (c_int, [c_void_p, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _P, _P, _I, c_int, _F, _P]),
    "icv_comm_unique_id": (c_int, [c_char_p]),
    "icv_comm_create": (c_int, [c_char_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "icv_comm_destroy": (None, [c_void_p]),
    "icv_allgather_kv": (c_int, [c_void_p, _P, _P, _I, _I, _P]),
    "icv_ipc_create": (c_int, [c_char_p, c_int, c_int, _P, _I, ctypes.POINTER(c_void_p)]),
    "icv_ipc_destroy": (None, [c_void_p]),
    "icv_ipc_shm_unlink": (c_int, [c_char_p]),
    "icv_ipc_heap": (c_int, [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64)]),
    "icv_ipc_export": (c_int, [c_void_p, c_char_p]),
    "icv_ipc_open_peer": (c_int, [c_void_p, c_int, c_char_p]),
    "icv_ipc_gather_start": (c_int, [c_void_p, _I, _I, _P, _P, ctypes.POINTER(c_int64)]),
    "icv_ipc_gather_wait": (c_int, [c_void_p, _I, _P]),
    "icv_ipc_acquire": (c_int, [c_void_p, _P]),
    "icv_ipc_tickets": (c_int64, [c_void_p]),
    "icv_ipc_abort": (c_int, [c_void_p]),
    "icv_ipc_arrival": (c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    "icv_ipc_gather_consumed": (c_int, [c_void_p, _I]),
    "icv_ipc_configure": (c_int, [c_void_p, c_int]),
    "icv_ipc_check": (c_int, [c_void_p]),
    "icv_ipc_drain": (c_int, [c_void_p, c_int]),
    "icv_ipc_probe_copy": (c_int, [c_void_p, c_int, _I, ctypes.POINTER(c_int), ctypes.POINTER(c_double)]),
    "icv_probe_copy_path": (c_int, [_P, _P, _I, ctypes.POINTER(c_int), ctypes.POINTER(c_double)]),
    "icv_attention_fwd_pieces": (c_int, [_P, _I, _P, _I, _I, _I, _P, _I, _I, _I, c_float, _P, _P, _I, _P, _P]),
    "icv_flag_write": (c_int, [_P, _I, ctypes.c_uint32, _I, _P]),
    "icv_dit_profile": (c_int, [c_void_p, c_int]),
    "icv_dit_profile_read": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
})

class KVPiece(ctypes.Structure):
    """icv_kv_piece of include/icvideo.h (icv_attention_fwd_pieces)."""
    _fields_ = [("k", c_void_p), ("v", c_void_p), ("rows", c_int64), ("flag", ctypes.c_int32), ("value", ctypes.c_uint32)]


ATTN_MAX_PIECES = 64   # ICV_ATTN_MAX_PIECES
COMM_ID_BYTES = 128   # ICV_COMM_ID_BYTES
IPC_HANDLE_BYTES = 72  # ICV_IPC_HANDLE_BYTES
IPC_SLOTS = 32         # ICV_IPC_SLOTS
ABI_VERSION = 5       # ICV_ABI_VERSION of include/icvideo.h

_lib: Optional[ctypes.CDLL] = None


class NativeError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load libicvideo.so (once).  Raises NativeError loudly if it is absent or broken."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} not found: build it with infinicube_amd/csrc/build.sh "
            "(or python -c 'import __graft_entry__ as g; g.build()').  There is no CPU/eager fallback.")
    try:
        l = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the box
        raise NativeError(f"failed to load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise NativeError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    got = l.icv_abi_version()
    if got != ABI_VERSION:
        raise NativeError(f"libicvideo ABI version {got}, binding expects {ABI_VERSION} (rebuild with infinicube_amd/csrc/build.sh)")
    _lib = l
    return l


N_CALLS = [0]   # C-ABI calls issued by this process (one call = one kernel launch, or a few); bench.py reports it


def check(rc: int, what: str = "") -> None:
    N_CALLS[0] += 1
    if rc != 0:
        msg = lib().icv_last_error()
        raise NativeError(f"{what or 'libicvideo'} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
