"""TEST-ONLY operator set: the ops.HipOps interface implemented with the CPU oracle's arithmetic.

Lets the `-m "not gpu"` tests drive the product's host logic (infinicube_amd/videogen/dit.py:
weight packing, caches, split-QKV layout, sequence-parallel sharding, the fused unpatchify/CFG/
Euler indexing) on CPU, and gives the GPU tests a rounding-point-exact emulation of the HIP
pipeline (fp32 math, bf16 storage where the C ABI stores bf16).  The product never imports this.
"""

from __future__ import annotations

import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wan_ref as R  # noqa: E402

BF16, F32, FP8 = torch.bfloat16, torch.float32, torch.float8_e4m3fn
EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32, EPI_F32 = 0, 1, 2, 3


class OracleOps:
    name = "oracle-cpu"

    def __init__(self, device="cpu"):
        self.device = torch.device(device)

    def alloc(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def to_device(self, t, dtype):
        return t.detach().to(device=self.device, dtype=dtype).contiguous()

    def gemm(self, a, w, bias, out, epilogue, resid=None, gate=None, nsplit=None):
        assert a.dtype == BF16 and w.dtype == BF16
        acc = a.float() @ w.float().t()
        if bias is not None:
            acc = acc + bias
        if epilogue == EPI_GELU_BF16:
            acc = F.gelu(acc, approximate="tanh")
        if epilogue == EPI_RESID_F32:
            acc = resid + (gate * acc if gate is not None else acc)
        if nsplit is None:
            out.copy_(acc.to(out.dtype))
        else:
            M, N = acc.shape
            out.copy_(acc.reshape(M, N // nsplit, nsplit).permute(1, 0, 2).to(out.dtype))

    def quantize_rows(self, src, out_q, out_scale):
        q, sc = R.quantize_rows_fp8(src.float())
        out_q.copy_(q.to(FP8)); out_scale.copy_(sc)

    def ln_modulate_fp8(self, x, out_q, out_scale, weight=None, bias=None, shift=None, scale=None, eps=1e-6):
        y = R.layer_norm(x, weight, bias, eps)
        if scale is not None:
            y = y * (1.0 + scale)
        if shift is not None:
            y = y + shift
        self.quantize_rows(y, out_q, out_scale)

    def gemm_fp8(self, a_q, a_scale, w_q, w_scale, bias, out, epilogue, resid=None, gate=None, nsplit=None):
        assert a_q.dtype == FP8 and w_q.dtype == FP8
        acc = (a_q.float() @ w_q.float().t()) * a_scale[:, None] * w_scale[None, :]
        if bias is not None:
            acc = acc + bias
        if epilogue == EPI_GELU_BF16:
            acc = F.gelu(acc, approximate="tanh")
        if epilogue == EPI_RESID_F32:
            acc = resid + (gate * acc if gate is not None else acc)
        if nsplit is None:
            out.copy_(acc.to(out.dtype))
        else:
            M, N = acc.shape
            out.copy_(acc.reshape(M, N // nsplit, nsplit).permute(1, 0, 2).to(out.dtype))

    def gemv(self, x, w, bias, out, in_act=0, out_act=0):
        xi = F.silu(x) if in_act == 1 else x
        y = xi @ w.float().t()
        if bias is not None:
            y = y + bias
        out.copy_(F.silu(y) if out_act == 1 else y)

    def sinusoidal(self, timestep, out):
        out.copy_(R.sinusoidal_embedding_1d(out.numel(), torch.tensor([timestep], dtype=torch.float64))
                  .to(F32).reshape(out.shape))

    def bcast_add(self, a, b, out):
        out.copy_((a.reshape(-1, b.numel()) + b.reshape(1, -1)).reshape(out.shape))

    def ln_modulate(self, x, out, weight=None, bias=None, shift=None, scale=None, eps=1e-6):
        y = R.layer_norm(x, weight, bias, eps)
        if scale is not None:
            y = y * (1.0 + scale)
        if shift is not None:
            y = y + shift
        out.copy_(y.to(BF16))

    def rmsnorm_rope(self, x0, w0, x1=None, w1=None, eps=1e-6, rope=None, tok0=0):
        for x, w in ((x0, w0), (x1, w1)):
            if x is None:
                continue
            y = R.rms_norm(x.float(), w, eps)
            if rope is not None:
                freqs = R.rope_freqs_3d(128, rope.T, rope.Hp, rope.Wp)[tok0: tok0 + x.shape[0]]
                y = R.rope_apply(y, freqs, x.shape[1] // 128)
            x.copy_(y.to(BF16))

    def attention(self, q, k, v, o, heads, scale):
        o.copy_(R.attention(q.float(), k.float(), v.float(), heads, scale=scale).to(BF16))

    def attention_fp8_buffers(self, Sq, Skv, d, heads):
        return {}

    def attention_fp8(self, q, k, v, o, heads, ws):
        o.copy_(R.attention_fp8(q.float(), k.float(), v.float(), heads).to(BF16))

    def attention_fp8_prepare(self, ws, heads, q=None, k=None, v=None):
        def qd(x):   # per-head power-of-two scale + e4m3 rounding, dequantised
            S, d = x.shape
            xh = x.float().reshape(S, heads, -1)
            amax = xh.abs().amax(dim=(0, 2))
            e = torch.where(amax > 0, torch.ceil(torch.log2(amax / R.FP8_MAX)), torch.zeros_like(amax))
            return ((xh * torch.exp2(-e)[None, :, None]).to(FP8).float() * torch.exp2(e)[None, :, None]).reshape(S, d)
        if q is not None:
            ws["q"] = qd(q)
        if k is not None:
            ws["k"], ws["v"] = qd(k), qd(v)

    def attention_fp8_chunk(self, ws, Sq, Skv, o, acc, ml, heads, first, last):
        """Online softmax in log2 units over the prepared chunk (P rounded to e4m3 against the running max)."""
        q, k, v = ws["q"][:Sq], ws["k"][:Skv], ws["v"][:Skv]
        d = q.shape[1]
        qh = q.reshape(Sq, heads, -1).transpose(0, 1)
        kh = k.reshape(Skv, heads, -1).transpose(0, 1)
        vh = v.reshape(Skv, heads, -1).transpose(0, 1)
        s = qh @ kh.transpose(1, 2)
        if first:
            m_old = torch.full((heads, Sq), -1e30); l_old = torch.zeros((heads, Sq)); a_old = torch.zeros((heads, Sq, d // heads))
        else:
            m_old, l_old = ml[:, :, 0].t().clone(), ml[:, :, 1].t().clone()
            a_old = acc.reshape(Sq, heads, -1).transpose(0, 1).clone()
        m_new = torch.maximum(m_old, s.max(dim=-1).values)
        alpha = torch.exp2(m_old - m_new)
        pmat = torch.exp2(s - m_new[:, :, None])
        l_new = l_old * alpha + pmat.sum(-1)
        a_new = a_old * alpha[:, :, None] + pmat.to(FP8).float() @ vh
        if last:
            o.copy_((a_new / l_new[:, :, None]).transpose(0, 1).reshape(Sq, d).to(BF16))
        else:
            acc.copy_(a_new.transpose(0, 1).reshape(Sq, d))
            ml[:, :, 0] = m_new.t(); ml[:, :, 1] = l_new.t()

    # e4m3 K|V on the wire: the same three steps as HipOps (the blob layout is private to an operator set: here kq rows | vq rows)
    def attention_fp8_blob_bytes(self, rows, heads):
        return 2 * ((rows + 63) // 64 * 64) * heads * 128

    def attention_fp8_kv_amax(self, k, v, heads, amax):
        for row, x in ((1, k), (2, v)):
            amax[row] = x.float().reshape(x.shape[0], heads, -1).abs().amax(dim=(0, 2))

    @staticmethod
    def _exp(amax_row):
        return torch.where(amax_row > 0, torch.ceil(torch.log2(amax_row / R.FP8_MAX)), torch.zeros_like(amax_row))

    def attention_fp8_quantize_kv(self, k, v, heads, amax, blob):
        m, d = k.shape
        mp = (m + 63) // 64 * 64
        blob[: 2 * mp * d].zero_()
        for part, (row, x) in enumerate(((1, k), (2, v))):
            e = self._exp(amax[row])
            q8 = (x.float().reshape(m, heads, -1) * torch.exp2(-e)[None, :, None]).to(FP8).reshape(m, d)
            blob[part * mp * d: part * mp * d + m * d] = q8.view(torch.uint8).reshape(-1)

    def attention_fp8_pieces(self, ws, amax, blobs, piece_rows, n_pieces, Sq, o, acc, ml, heads, first, last, gate=None):
        d = heads * 128
        mp = (piece_rows + 63) // 64 * 64
        ks, vs = [], []
        order = [e[0] for e in gate["seq"]] if gate is not None else list(range(n_pieces))      # CPU twin: the rows are there (the host waited)
        own_t, own_i = (gate.get("own") or (None, -1)) if gate is not None else (None, -1)
        for i in order:
            b = own_t.reshape(-1)[: 2 * mp * d] if (own_t is not None and i == own_i) else blobs.reshape(-1)[i * 2 * mp * d: (i + 1) * 2 * mp * d]
            for part, (row, dst) in enumerate(((1, ks), (2, vs))):
                x8 = b[part * mp * d: part * mp * d + piece_rows * d].view(FP8).float().reshape(piece_rows, heads, -1)
                dst.append((x8 * torch.exp2(self._exp(amax[row]))[None, :, None]).reshape(piece_rows, d))
        tmp = dict(ws, k=torch.cat(ks, 0), v=torch.cat(vs, 0))
        self.attention_fp8_chunk(tmp, Sq, n_pieces * piece_rows, o, acc, ml, heads, first, last)

    def attention_fp8_with_amax(self, ws, amax):
        return ws

    def attention_add(self, q, k, v, o, heads, scale):
        o.copy_((o.float() + R.attention(q.float(), k.float(), v.float(), heads, scale=scale)).to(BF16))

    def attention_chunk(self, q, k, v, o, acc, ml, heads, scale, first, last):
        """Online-softmax over one chunk of keys with carried (acc, m, l) state, fp32."""
        Sq, d = q.shape
        qh = q.float().reshape(Sq, heads, -1).transpose(0, 1)
        kh = k.float().reshape(k.shape[0], heads, -1).transpose(0, 1)
        vh = v.float().reshape(v.shape[0], heads, -1).transpose(0, 1)
        s = qh @ kh.transpose(1, 2) * scale                                  # [H, Sq, Sk]
        if first:
            m_old = torch.full((heads, Sq), -1e30)
            l_old = torch.zeros((heads, Sq))
            a_old = torch.zeros((heads, Sq, d // heads))
        else:
            m_old, l_old = ml[:, :, 0].t().clone(), ml[:, :, 1].t().clone()
            a_old = acc.reshape(Sq, heads, -1).transpose(0, 1).clone()
        m_new = torch.maximum(m_old, s.max(dim=-1).values)
        alpha = torch.exp(m_old - m_new)
        pmat = torch.exp(s - m_new[:, :, None])
        l_new = l_old * alpha + pmat.sum(-1)
        a_new = a_old * alpha[:, :, None] + pmat @ vh
        if last:
            o.copy_((a_new / l_new[:, :, None]).transpose(0, 1).reshape(Sq, d).to(BF16))
        else:
            acc.copy_(a_new.transpose(0, 1).reshape(Sq, d))
            ml[:, :, 0] = m_new.t()
            ml[:, :, 1] = l_new.t()

    def attention_pieces(self, q, pieces, o, heads, scale, flags=None, err=None, timeout_us=0, trace=None):
        """The arrival-driven launch on CPU: the pieces are simply there (flags were waited for by the caller's work handles)."""
        ks = [k for k, v, _, _ in pieces if k.shape[0]]
        vs = [v for k, v, _, _ in pieces if k.shape[0]]
        o.copy_(R.attention(q.float(), torch.cat(ks).float(), torch.cat(vs).float(), heads, scale=scale).to(BF16))

    def patchify(self, latent, out, tok0, n_tok):
        C, T, H8, W8 = latent.shape
        Hp, Wp = H8 // 2, W8 // 2
        tok = latent.reshape(C, T, Hp, 2, Wp, 2).permute(1, 2, 4, 0, 3, 5).reshape(T * Hp * Wp, C * 4)
        out[:, : C * 4].copy_(tok[tok0: tok0 + n_tok].to(BF16))

    def unpatchify_cfg_euler(self, latent, hc, hu, cfg_scale, dsigma, tok0, n_tok, vel_out=None, round_bf16=False):
        C, T, H8, W8 = latent.shape
        Hp, Wp = H8 // 2, W8 // 2
        if round_bf16:
            rb = lambda t: t.to(torch.bfloat16).to(F32)   # noqa: E731
            v = rb(hc) if hu is None else rb(rb(hu) + rb(cfg_scale * rb(rb(hc) - rb(hu))))
            full = torch.zeros((T * Hp * Wp, 4 * C), dtype=F32)
            full[tok0: tok0 + n_tok] = v
            vel = R.unpatchify(full, (T, Hp, Wp), C)
            mask = torch.zeros((T * Hp * Wp, 4 * C), dtype=F32)
            mask[tok0: tok0 + n_tok] = 1.0
            m = R.unpatchify(mask, (T, Hp, Wp), C)
            latent.copy_(torch.where(m > 0, rb(rb(latent) + rb(vel * dsigma)), latent))
            if vel_out is not None:
                vel_out.copy_(torch.where(m > 0, vel, vel_out))
            return
        v = hc if hu is None else hu + cfg_scale * (hc - hu)
        full = torch.zeros((T * Hp * Wp, 4 * C), dtype=F32)
        full[tok0: tok0 + n_tok] = v
        vel = R.unpatchify(full, (T, Hp, Wp), C)
        mask = torch.zeros((T * Hp * Wp, 4 * C), dtype=F32)
        mask[tok0: tok0 + n_tok] = 1.0
        m = R.unpatchify(mask, (T, Hp, Wp), C)
        latent.add_(vel * m * dsigma)
        if vel_out is not None:
            vel_out.copy_(torch.where(m > 0, vel, vel_out))

    def cast_bf16(self, src, out):
        out.copy_(src.to(BF16))
