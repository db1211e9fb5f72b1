"""Operator layer of the DiT host code: thin, checked wrappers that hand torch tensors' device
pointers to the C ABI (include/icvideo.h).  PyTorch is plumbing here (HBM allocation, streams);
every arithmetic op of the denoising loop runs in libicvideo's HIP kernels.

The host driver (dit.py) is written against this small interface so that its sharding / caching
logic can be exercised on CPU by the tests, which inject their own checker implementation
(tests/oracle_ops.py).  The product never does that: ``HipOps`` is the only implementation in this
package and it raises if the native library or a GPU is missing.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import os

import ctypes

import torch

from .. import native
from ..native import EPI_BF16, EPI_F32, EPI_GELU_BF16, EPI_RESID_F32  # noqa: F401 (re-export)

BF16, F32 = torch.bfloat16, torch.float32
FP8 = torch.float8_e4m3fn   # OCP e4m3 (gfx950's fp8), one f32 scale per row next to it


@dataclass
class RopeTable:
    """Per-axis (cos, sin) f32 tables, concatenated [T][22] ++ [Hp][21] ++ [Wp][21] pairs
    (SURVEY.md Appendix A.3); angles are computed in fp64 on the host."""
    table: torch.Tensor  # f32 [(T*22 + Hp*21 + Wp*21), 2]
    T: int
    Hp: int
    Wp: int

    @staticmethod
    def build(T: int, Hp: int, Wp: int, device, head_dim: int = 128, theta: float = 10000.0) -> "RopeTable":
        hw = head_dim // 3
        dims = (head_dim - 2 * hw, hw, hw)
        parts = []
        for axis_dim, n in zip(dims, (T, Hp, Wp)):
            inv = 1.0 / (theta ** (torch.arange(0, axis_dim, 2, dtype=torch.float64)[: axis_dim // 2] / axis_dim))
            ang = torch.outer(torch.arange(n, dtype=torch.float64), inv)
            parts.append(torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).reshape(-1, 2))
        tab = torch.cat(parts, dim=0).to(torch.float32).contiguous().to(device)
        return RopeTable(tab, T, Hp, Wp)


def _chk(t: torch.Tensor, dtype, name: str, last_contig: bool = True):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if last_contig and t.dim() >= 1 and t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")


class HipOps:
    """The product operator set: every method enqueues one libicvideo kernel on the current
    HIP stream of ``device``."""

    name = "hip"

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise native.NativeError(
                f"HipOps needs a ROCm GPU device ('cuda:N'), got {device!r}; there is no CPU fallback")
        if not torch.cuda.is_available():
            raise native.NativeError("HipOps: no GPU visible to PyTorch-ROCm; there is no CPU fallback")
        self.lib = native.lib()
        # one process drives one GPU: kernels are launched on this device's current stream and the library keeps
        # per-process (not per-device) launch attributes, so make it the process's current HIP device
        torch.cuda.set_device(self.device)
        # kernel A/B switches (icv_set_option) from the environment, e.g. ICV_OPTIONS="attn2_variant=4,gemm256=1"
        for item in filter(None, os.environ.get("ICV_OPTIONS", "").split(",")):
            name, _, val = item.partition("=")
            native.check(self.lib.icv_set_option(name.strip().encode(), int(val)), f"icv_set_option({item})")

    # -- helpers ---------------------------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def alloc(self, shape, dtype) -> torch.Tensor:
        return torch.empty(shape, dtype=dtype, device=self.device)

    def to_device(self, t: torch.Tensor, dtype) -> torch.Tensor:
        return t.detach().to(device=self.device, dtype=dtype).contiguous()

    # -- kernels ---------------------------------------------------------------------------
    def gemm(self, a, w, bias, out, epilogue, resid=None, gate=None, nsplit=None):
        """out = epilogue(a @ w.T + bias).  a [M,K] bf16, w [N,K] bf16, bias f32[N] | None.
        ``out`` is [M,N] (bf16 or f32 by epilogue), or [N/nsplit, M, nsplit] when nsplit is set."""
        _chk(a, BF16, "gemm.a"); _chk(w, BF16, "gemm.w")
        M, K = a.shape
        N = w.shape[0]
        if w.shape[1] != K:
            raise ValueError(f"gemm: K mismatch {a.shape} x {w.shape}")
        want = BF16 if epilogue in (EPI_BF16, EPI_GELU_BF16) else F32
        _chk(out, want, "gemm.out")
        if nsplit is None:
            if tuple(out.shape) != (M, N):
                raise ValueError(f"gemm: out shape {tuple(out.shape)} != {(M, N)}")
            ldo, ns, sstride = out.stride(0), N, 0
        else:
            if tuple(out.shape) != (N // nsplit, M, nsplit):
                raise ValueError(f"gemm: split out shape {tuple(out.shape)} != {(N // nsplit, M, nsplit)}")
            ldo, ns, sstride = out.stride(1), nsplit, out.stride(0)
        if bias is not None:
            _chk(bias, F32, "gemm.bias")
        if resid is not None:
            _chk(resid, F32, "gemm.resid")
        if gate is not None:
            _chk(gate, F32, "gemm.gate")
        native.check(self.lib.icv_gemm_bf16(
            a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), native.ptr(bias), M, N, K, epilogue,
            out.data_ptr(), ldo, ns, sstride, native.ptr(resid),
            resid.stride(0) if resid is not None else 0, native.ptr(gate), self._stream()), "icv_gemm_bf16")

    def gemv(self, x, w, bias, out, in_act=0, out_act=0):
        _chk(x, F32, "gemv.x"); _chk(w, BF16, "gemv.w"); _chk(out, F32, "gemv.out")
        M, K = x.shape
        N = w.shape[0]
        assert x.is_contiguous() and w.is_contiguous() and out.is_contiguous() and tuple(out.shape) == (M, N)
        native.check(self.lib.icv_gemv_f32(x.data_ptr(), w.data_ptr(), native.ptr(bias), out.data_ptr(),
                                           M, N, K, in_act, out_act, self._stream()), "icv_gemv_f32")

    def sinusoidal(self, timestep: float, out):
        _chk(out, F32, "sinusoidal.out")
        native.check(self.lib.icv_sinusoidal_embedding(float(timestep), out.numel(), out.data_ptr(),
                                                       self._stream()), "icv_sinusoidal_embedding")

    def bcast_add(self, a, b, out):
        _chk(a, F32, "bcast_add.a"); _chk(b, F32, "bcast_add.b"); _chk(out, F32, "bcast_add.out")
        assert a.is_contiguous() and b.is_contiguous() and out.is_contiguous()
        rows, n = a.numel() // b.numel(), b.numel()
        native.check(self.lib.icv_bcast_add_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), rows, n,
                                                self._stream()), "icv_bcast_add_f32")

    def ln_modulate(self, x, out, weight=None, bias=None, shift=None, scale=None, eps=1e-6):
        _chk(x, F32, "ln.x"); _chk(out, BF16, "ln.out")
        rows, d = x.shape
        native.check(self.lib.icv_ln_modulate(
            x.data_ptr(), x.stride(0), native.ptr(weight), native.ptr(bias), native.ptr(shift),
            native.ptr(scale), out.data_ptr(), out.stride(0), rows, d, eps, self._stream()), "icv_ln_modulate")

    # ---- fp8 path (BASELINE.json config #5): e4m3 rows + one f32 scale per row ----------------------
    def quantize_rows(self, src, out_q, out_scale):
        """src bf16|f32 [rows, K] -> out_q e4m3 [rows, K], out_scale f32 [rows] (row abs-max / 448)."""
        if src.dtype not in (BF16, F32):
            raise TypeError(f"quantize_rows.src: expected bf16 or f32, got {src.dtype}")
        _chk(src, src.dtype, "quantize_rows.src"); _chk(out_q, FP8, "quantize_rows.out"); _chk(out_scale, F32, "quantize_rows.scale")
        rows, K = src.shape
        assert tuple(out_q.shape) == (rows, K) and out_scale.numel() == rows and out_scale.is_contiguous()
        native.check(self.lib.icv_quantize_rows_fp8(
            src.data_ptr(), int(src.dtype == F32), src.stride(0), rows, K, out_q.data_ptr(), out_q.stride(0),
            out_scale.data_ptr(), self._stream()), "icv_quantize_rows_fp8")

    def ln_modulate_fp8(self, x, out_q, out_scale, weight=None, bias=None, shift=None, scale=None, eps=1e-6):
        _chk(x, F32, "ln8.x"); _chk(out_q, FP8, "ln8.out"); _chk(out_scale, F32, "ln8.scale")
        rows, d = x.shape
        assert tuple(out_q.shape) == (rows, d) and out_scale.numel() == rows and out_scale.is_contiguous()
        native.check(self.lib.icv_ln_modulate_fp8(
            x.data_ptr(), x.stride(0), native.ptr(weight), native.ptr(bias), native.ptr(shift),
            native.ptr(scale), out_q.data_ptr(), out_q.stride(0), out_scale.data_ptr(), rows, d, eps,
            self._stream()), "icv_ln_modulate_fp8")

    def gemm_fp8(self, a_q, a_scale, w_q, w_scale, bias, out, epilogue, resid=None, gate=None, nsplit=None):
        """out = epilogue((a_q @ w_q.T) * a_scale[:, None] * w_scale[None, :] + bias); shapes as ``gemm``."""
        _chk(a_q, FP8, "gemm8.a"); _chk(w_q, FP8, "gemm8.w"); _chk(a_scale, F32, "gemm8.a_scale"); _chk(w_scale, F32, "gemm8.w_scale")
        M, K = a_q.shape
        N = w_q.shape[0]
        if w_q.shape[1] != K:
            raise ValueError(f"gemm_fp8: K mismatch {a_q.shape} x {w_q.shape}")
        assert a_scale.numel() == M and w_scale.numel() == N and a_scale.is_contiguous() and w_scale.is_contiguous()
        want = BF16 if epilogue in (EPI_BF16, EPI_GELU_BF16) else F32
        _chk(out, want, "gemm8.out")
        if nsplit is None:
            if tuple(out.shape) != (M, N):
                raise ValueError(f"gemm_fp8: out shape {tuple(out.shape)} != {(M, N)}")
            ldo, ns, sstride = out.stride(0), N, 0
        else:
            if tuple(out.shape) != (N // nsplit, M, nsplit):
                raise ValueError(f"gemm_fp8: split out shape {tuple(out.shape)} != {(N // nsplit, M, nsplit)}")
            ldo, ns, sstride = out.stride(1), nsplit, out.stride(0)
        for t, nm in ((bias, "bias"), (resid, "resid"), (gate, "gate")):
            if t is not None:
                _chk(t, F32, f"gemm8.{nm}")
        native.check(self.lib.icv_gemm_fp8(
            a_q.data_ptr(), a_q.stride(0), a_scale.data_ptr(), w_q.data_ptr(), w_q.stride(0), w_scale.data_ptr(),
            native.ptr(bias), M, N, K, epilogue, out.data_ptr(), ldo, ns, sstride, native.ptr(resid),
            resid.stride(0) if resid is not None else 0, native.ptr(gate), self._stream()), "icv_gemm_fp8")

    def rmsnorm_rope(self, x0, w0, x1=None, w1=None, eps=1e-6, rope: Optional[RopeTable] = None, tok0=0):
        _chk(x0, BF16, "rms.x0"); _chk(w0, F32, "rms.w0")
        rows, d = x0.shape
        ld = x0.stride(0)
        if x1 is not None:
            _chk(x1, BF16, "rms.x1"); _chk(w1, F32, "rms.w1")
            assert x1.shape == x0.shape and x1.stride(0) == ld
        native.check(self.lib.icv_rmsnorm_rope(
            x0.data_ptr(), w0.data_ptr(), native.ptr(x1), native.ptr(w1), ld, rows, d, eps,
            rope.table.data_ptr() if rope is not None else None,
            rope.T if rope else 0, rope.Hp if rope else 0, rope.Wp if rope else 0, tok0,
            self._stream()), "icv_rmsnorm_rope")

    def attention(self, q, k, v, o, heads: int, scale: float):
        for t, nm in ((q, "q"), (k, "k"), (v, "v"), (o, "o")):
            _chk(t, BF16, f"attention.{nm}")
        native.check(self.lib.icv_attention_fwd(
            q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
            o.data_ptr(), o.stride(0), q.shape[0], k.shape[0], heads, scale, self._stream()),
            "icv_attention_fwd")

    def attention_add(self, q, k, v, o, heads: int, scale: float):
        """o += softmax(q k^T * scale) v  (i2v image cross-attention summed onto the text one)."""
        for t, nm in ((q, "q"), (k, "k"), (v, "v"), (o, "o")):
            _chk(t, BF16, f"attention_add.{nm}")
        native.check(self.lib.icv_attention_fwd_add(
            q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
            o.data_ptr(), o.stride(0), q.shape[0], k.shape[0], heads, scale, self._stream()),
            "icv_attention_fwd_add")

    # ---- fp8 attention (fp8 mode only): e4m3 Q/K/V/P on the K=64 scaled MFMA ----------------------------------
    def attention_fp8_buffers(self, Sq: int, Skv: int, d: int, heads: int):
        """Workspace of attention_fp8: (qq [Sq,d], kq [Skv,d], vt bytes, amax f32 [3, heads])."""
        nbytes = int(self.lib.icv_attention_fp8_vt_bytes(Skv, heads))
        return (self.alloc((Sq, d), FP8), self.alloc((Skv, d), FP8), self.alloc((nbytes,), torch.uint8), self.alloc((3, heads), F32))

    def attention_fp8_prepare(self, ws, heads: int, q=None, k=None, v=None):
        """Quantise the queries and / or the keys + values of one attention into the workspace (per-head power-of-two
        scales from the abs-max, e4m3 rows, transposed key-permuted V tiles)."""
        qq, kq, vt, amax = ws
        if q is not None:
            _chk(q, BF16, "attention_fp8.q"); assert qq.shape[0] >= q.shape[0] and qq.shape[1] == q.shape[1]
        if k is not None:
            _chk(k, BF16, "attention_fp8.k"); _chk(v, BF16, "attention_fp8.v")
            assert kq.shape[0] >= k.shape[0] and kq.shape[1] == k.shape[1] and v.shape == k.shape
            assert vt.numel() >= int(self.lib.icv_attention_fp8_vt_bytes(k.shape[0], heads))
        native.check(self.lib.icv_attention_fp8_prepare(
            native.ptr(q), q.stride(0) if q is not None else 0, native.ptr(k), k.stride(0) if k is not None else 0,
            native.ptr(v), v.stride(0) if v is not None else 0, q.shape[0] if q is not None else 0,
            k.shape[0] if k is not None else 0, heads, qq.data_ptr() if q is not None else None, qq.stride(0),
            kq.data_ptr() if k is not None else None, kq.stride(0), vt.data_ptr() if k is not None else None,
            amax.data_ptr(), self._stream()), "icv_attention_fp8_prepare")

    def attention_fp8(self, q, k, v, o, heads: int, ws):
        """o = softmax2(q k^T) v with e4m3 operands; K must carry scale * log2(e) (the DiT's unit-scale convention).
        ``ws`` = attention_fp8_buffers(...) sized for these shapes."""
        _chk(o, BF16, "attention_fp8.o")
        self.attention_fp8_prepare(ws, heads, q=q, k=k, v=v)
        qq, kq, vt, amax = ws
        native.check(self.lib.icv_attention_fp8_fwd(
            qq.data_ptr(), qq.stride(0), kq.data_ptr(), kq.stride(0), vt.data_ptr(), amax.data_ptr(), o.data_ptr(), o.stride(0),
            q.shape[0], k.shape[0], heads, self._stream()), "icv_attention_fp8_fwd")

    def attention_fp8_chunk(self, ws, Sq: int, Skv: int, o, acc, ml, heads: int, first: bool, last: bool):
        """fp8 attention of the prepared queries over the prepared chunk of keys / values, carried state as
        attention_chunk (acc f32 [Sq, H*128], ml f32 [Sq, H, 2])."""
        qq, kq, vt, amax = ws
        if o is not None:
            _chk(o, BF16, "attention_fp8_chunk.o")
        native.check(self.lib.icv_attention_fp8_fwd_chunk(
            qq.data_ptr(), qq.stride(0), kq.data_ptr(), kq.stride(0), vt.data_ptr(), amax.data_ptr(), native.ptr(o),
            o.stride(0) if o is not None else 0, native.ptr(acc), acc.stride(0) if acc is not None else 0, native.ptr(ml),
            Sq, Skv, heads, int(first), int(last), self._stream()), "icv_attention_fp8_fwd_chunk")

    # e4m3 K|V on the wire (sequence parallel): quantise the LOCAL rows once, exchange e4m3 blobs, attend over the pieces
    def attention_fp8_blob_bytes(self, rows: int, heads: int) -> int:
        return int(self.lib.icv_attention_fp8_blob_bytes(rows, heads))

    def attention_fp8_kv_amax(self, k, v, heads: int, amax):
        """amax f32 [3, H]: rows 1 / 2 <- per-head abs-max of this rank's k / v rows (row 0, the queries', is left alone)."""
        _chk(k, BF16, "attention_fp8_kv_amax.k"); _chk(v, BF16, "attention_fp8_kv_amax.v"); _chk(amax, F32, "attention_fp8_kv_amax.amax")
        native.check(self.lib.icv_attention_fp8_kv_amax(k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), k.shape[0], heads, amax.data_ptr(),
                                                        self._stream()), "icv_attention_fp8_kv_amax")

    def attention_fp8_quantize_kv(self, k, v, heads: int, amax, blob):
        """rows of k / v -> one e4m3 blob (kq rows | transposed V tiles) with the per-head scales of ``amax``."""
        assert blob.dtype == torch.uint8 and blob.is_contiguous() and blob.numel() >= self.attention_fp8_blob_bytes(k.shape[0], heads)
        native.check(self.lib.icv_attention_fp8_quantize_kv(k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), k.shape[0], heads, amax.data_ptr(),
                                                            blob.data_ptr(), self._stream()), "icv_attention_fp8_quantize_kv")

    def attention_fp8_pieces(self, ws, amax, blobs, piece_rows: int, n_pieces: int, Sq: int, o, acc, ml, heads: int, first: bool, last: bool, gate=None):
        """fp8 attention of the prepared queries (ws) over ``n_pieces`` blobs back to back (the gathered chunk), carried state as
        attention_chunk.  ``gate`` (arrival-driven, icv_attention_fp8_fwd_pieces_gated): dict(seq=[(piece, flag, value), ...] - the order in
        which the pieces are walked, this rank's own first; flag < 0 = there now, else readable once (int32)(flags[flag] - value) >= 0 -,
        flags=device words, own=(uint8 tensor, index) or None: that piece is read from the tensor instead of its slot in ``blobs``,
        err=int32 device word or None, timeout_us)."""
        qq = ws[0]
        assert blobs.dtype == torch.uint8 and blobs.is_contiguous() and blobs.numel() >= n_pieces * self.attention_fp8_blob_bytes(piece_rows, heads)
        if gate is not None:
            seq = gate["seq"]
            assert len(seq) == n_pieces
            sp = (ctypes.c_int32 * n_pieces)(*[int(e[0]) for e in seq])
            sf = (ctypes.c_int32 * n_pieces)(*[int(e[1]) for e in seq])
            sv = (ctypes.c_uint32 * n_pieces)(*[int(e[2]) & 0xffffffff for e in seq])
            own = gate.get("own")
            own_t, own_i = own if own is not None else (None, -1)
            if own_t is not None:
                assert own_t.dtype == torch.uint8 and own_t.is_contiguous() and own_t.numel() >= self.attention_fp8_blob_bytes(piece_rows, heads)
            native.check(self.lib.icv_attention_fp8_fwd_pieces_gated(
                qq.data_ptr(), qq.stride(0), blobs.data_ptr(), piece_rows, n_pieces, native.ptr(own_t), int(own_i), ctypes.cast(sp, ctypes.c_void_p),
                ctypes.cast(sf, ctypes.c_void_p), ctypes.cast(sv, ctypes.c_void_p), native.ptr(gate.get("flags")), native.ptr(gate.get("err")),
                int(gate.get("timeout_us", 0)), amax.data_ptr(), native.ptr(o), o.stride(0) if o is not None else 0, native.ptr(acc),
                acc.stride(0) if acc is not None else 0, native.ptr(ml), Sq, heads, int(first), int(last), self._stream()),
                "icv_attention_fp8_fwd_pieces_gated")
            return
        native.check(self.lib.icv_attention_fp8_fwd_pieces(
            qq.data_ptr(), qq.stride(0), blobs.data_ptr(), piece_rows, n_pieces, amax.data_ptr(), native.ptr(o), o.stride(0) if o is not None else 0,
            native.ptr(acc), acc.stride(0) if acc is not None else 0, native.ptr(ml), Sq, heads, int(first), int(last), self._stream()),
            "icv_attention_fp8_fwd_pieces")

    def attention_fp8_with_amax(self, ws, amax):
        """The same workspace with another abs-max table (one per CFG branch: the branches' K / V scales differ)."""
        return (ws[0], ws[1], ws[2], amax)

    def attention_chunk(self, q, k, v, o, acc, ml, heads: int, scale: float, first: bool, last: bool):
        """Attention over one chunk of keys with carried softmax state (acc f32 [Sq, H*128], ml f32
        [Sq, H, 2]); ``first`` starts from the empty state, ``last`` normalises into ``o``."""
        for t, nm in ((q, "q"), (k, "k"), (v, "v")):
            _chk(t, BF16, f"attention_chunk.{nm}")
        if o is not None:
            _chk(o, BF16, "attention_chunk.o")
        if acc is not None:
            _chk(acc, F32, "attention_chunk.acc"); _chk(ml, F32, "attention_chunk.ml")
            assert ml.is_contiguous() and tuple(ml.shape) == (q.shape[0], heads, 2)
        native.check(self.lib.icv_attention_fwd_chunk(
            q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
            native.ptr(o), o.stride(0) if o is not None else 0, native.ptr(acc),
            acc.stride(0) if acc is not None else 0, native.ptr(ml), q.shape[0], k.shape[0], heads, scale,
            int(first), int(last), self._stream()), "icv_attention_fwd_chunk")

    def attention_pieces(self, q, pieces, o, heads: int, scale: float, flags=None, err=None, timeout_us: int = 0, trace=None):
        """K6, arrival-driven (csrc/attn7p.hip): ONE launch over the K|V ``pieces`` = [(k, v, flag, value), ...] - k, v bf16 row views
        with one common row stride each; flag < 0: the rows are in place now (this rank's own rows: list them FIRST); otherwise the rows
        are there once (int32)(flags[flag] - value) >= 0 (``flags`` int32 / uint32 device tensor written behind the transfer).  A piece
        that is late costs a bounded in-kernel wait; after ``timeout_us`` the kernel stores 0x80000000 | piece into ``err`` and goes on."""
        _chk(q, BF16, "attention_pieces.q"); _chk(o, BF16, "attention_pieces.o")
        arr = (native.KVPiece * len(pieces))()
        ldk = ldv = None
        for i, (k, v, flag, value) in enumerate(pieces):
            if k.shape[0] == 0:
                arr[i] = native.KVPiece(None, None, 0, -1, 0)
                continue
            _chk(k, BF16, "attention_pieces.k"); _chk(v, BF16, "attention_pieces.v")
            assert ldk in (None, k.stride(0)) and ldv in (None, v.stride(0)), "attention_pieces: the pieces must share their row strides"
            ldk, ldv = k.stride(0), v.stride(0)
            arr[i] = native.KVPiece(k.data_ptr(), v.data_ptr(), k.shape[0], int(flag), int(value) & 0xffffffff)
        native.check(self.lib.icv_attention_fwd_pieces(
            q.data_ptr(), q.stride(0), ctypes.cast(arr, ctypes.c_void_p), len(pieces), ldk or 0, ldv or 0, o.data_ptr(), o.stride(0), q.shape[0], heads, scale,
            native.ptr(flags), native.ptr(err), int(timeout_us), native.ptr(trace), self._stream()), "icv_attention_fwd_pieces")

    def flag_write(self, flags, index: int, value: int, delay_us: int = 0, stream: Optional[int] = None):
        """flags[index] <- value (system-scope release) on ``stream`` (default: the current one), optionally after holding it delay_us."""
        assert flags.dtype in (torch.int32, torch.uint32) and flags.is_contiguous()
        native.check(self.lib.icv_flag_write(flags.data_ptr(), index, int(value) & 0xffffffff, int(delay_us),
                                             self._stream() if stream is None else stream), "icv_flag_write")

    def patchify(self, latent, out, tok0: int, n_tok: int):
        _chk(latent, F32, "patchify.latent"); _chk(out, BF16, "patchify.out")
        assert latent.is_contiguous()
        C, T, H8, W8 = latent.shape
        native.check(self.lib.icv_patchify(latent.data_ptr(), C, T, H8, W8, out.data_ptr(),
                                           out.stride(0), tok0, n_tok, self._stream()), "icv_patchify")

    def unpatchify_cfg_euler(self, latent, hc, hu, cfg_scale, dsigma, tok0, n_tok, vel_out=None, round_bf16=False):
        _chk(latent, F32, "euler.latent"); _chk(hc, F32, "euler.hc")
        assert latent.is_contiguous()
        C, T, H8, W8 = latent.shape
        native.check(self.lib.icv_unpatchify_cfg_euler(
            latent.data_ptr(), native.ptr(vel_out), hc.data_ptr(), native.ptr(hu), hc.stride(0),
            cfg_scale, dsigma, C, T, H8, W8, tok0, n_tok, int(bool(round_bf16)), self._stream()), "icv_unpatchify_cfg_euler")

    def cast_bf16(self, src, out):
        _chk(src, F32, "cast.src"); _chk(out, BF16, "cast.out")
        assert src.is_contiguous() and out.is_contiguous() and src.numel() == out.numel()
        native.check(self.lib.icv_cast_f32_to_bf16(src.data_ptr(), out.data_ptr(), src.numel(),
                                                   self._stream()), "icv_cast_f32_to_bf16")
