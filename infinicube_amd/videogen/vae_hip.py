"""The Wan-VAE's tile networks on libicvideo's shifted-row convolution kernel (csrc/conv.hip) instead of MIOpen.

Same function as ``vae.WanVAENet.encode / decode`` (same modules, same parameters — this file only EXECUTES them), but every
activation lives as a PADDED NDHWC volume (``Vol``): the causal time padding and the spatial padding are real zero rows, so a
convolution tap is one constant row offset and every nn.Conv3d / nn.Conv2d of the network is ONE call of ``icv_conv3d_ndhwc``
(an MFMA GEMM over (tap, channel) with a shifting A-row base) with its bias and, for the last convolution of a residual block,
the residual add fused into the epilogue; the channel RMS norm + SiLU in front of a convolution is ``icv_rmsnorm_act_volume``,
which also writes the zero padding the convolution relies on.  What this buys over stock PyTorch-ROCm (MIOpen):

  * no kernel search: the first call of a process costs what every call costs (MIOpen's find step was ~20 s of the first
    ``generate()``, profiles/r04/e2e_generate_14b.json), and every process runs the SAME kernels — tiles dealt to the ranks
    of an N-GPU run are bit-identical to the unsharded call on the GPU too;
  * no causal F.pad copies, no NCDHW<->NDHWC transposes, no separate bias / residual passes.

Stride-2 convolutions (encoder) are evaluated at stride 1 and subsampled; the nearest-neighbour upsampling and the temporal
interleave of the decoder are strided copies between volumes.  Numerics: bf16 operands, fp32 accumulation, ONE rounding per
convolution (bias and residual are added in fp32 before it) — within bf16 rounding of the stock module
(tests/test_vae_hip_gpu.py compares both with the fp32 network).
"""

from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import native

BF16 = torch.bfloat16
PT = 2          # leading zero frames of every volume (the causal padding of a 3-tap time axis)


class Vol:
    """Padded NDHWC volume [T + PT, H + 2, W + 2, C] as a row matrix, with enough addressable rows on both sides for every
    tap of a 3x3x3 convolution evaluated on the rows of the real frames."""

    POISON = False      # tests: fill every new volume with NaNs first, so that a read of a halo / margin that should not matter shows

    def __init__(self, T: int, H: int, W: int, C: int, device, zero_pads: bool = True):
        self.T, self.H, self.W, self.C = T, H, W, C
        self.Hp, self.Wp, self.Tp = H + 2, W + 2, T + PT
        self.frame_rows = self.Hp * self.Wp
        self.rows = self.Tp * self.frame_rows
        self.margin = 2 * self.Wp + 4          # the farthest tap: two rows down + two pixels right (the stride-1 form of the strided 3x3)
        self.buf = torch.empty((self.margin + self.rows + self.margin, C), dtype=BF16, device=device)
        if self.POISON:
            self.buf.fill_(float("nan"))
        if zero_pads:       # the padding frames are READ by real outputs; the margins only by halo outputs (kept finite anyway)
            self.buf[: self.margin + PT * self.frame_rows].zero_()
            self.buf[self.margin + self.rows:].zero_()

    @property
    def mat(self) -> torch.Tensor:               # [rows, C]: row 0 = padded position (0, 0, 0)
        return self.buf[self.margin: self.margin + self.rows]

    def vol(self) -> torch.Tensor:               # [Tp, Hp, Wp, C]
        return self.mat.view(self.Tp, self.Hp, self.Wp, self.C)

    def interior(self) -> torch.Tensor:          # [T, H, W, C] strided view of the real positions
        return self.vol()[PT:, 1:-1, 1:-1]

    def zero_halo(self) -> "Vol":
        """Zero the spatial halo of the real frames (needed only where a convolution reads a volume that no norm wrote)."""
        v = self.vol()[PT:]
        v[:, 0].zero_(); v[:, -1].zero_(); v[:, :, 0].zero_(); v[:, :, -1].zero_()
        return self


def _ceil(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class _ConvW:
    """A convolution's parameters in the kernel's layout: bf16 [cout_pad, K] (tap-major, channel-minor, zero-padded channels /
    filters / K) + f32 bias, and the tap geometry (dt, dh, dw offsets relative to the output position)."""

    def __init__(self, mod: nn.Module, device, taps: Tuple[Tuple[int, int, int], ...], cin_real: int):
        w = mod.weight.detach()
        cout = w.shape[0]
        w = w.reshape(cout, cin_real, -1).permute(0, 2, 1)          # [cout, taps, cin]
        assert w.shape[1] == len(taps), (w.shape, len(taps))
        self.cin, self.cout, self.cout_real = _ceil(cin_real, 32), _ceil(cout, 4), cout
        K = _ceil(len(taps) * self.cin, 64)
        wk = torch.zeros((self.cout, len(taps), self.cin), dtype=torch.float32, device=device)
        wk[:cout, :, :cin_real] = w.to(device=device, dtype=torch.float32)
        self.w = torch.zeros((self.cout, K), dtype=BF16, device=device)
        self.w[:, : len(taps) * self.cin] = wk.reshape(self.cout, -1).to(BF16)
        self.bias = torch.zeros((self.cout,), dtype=torch.float32, device=device)
        if mod.bias is not None:
            self.bias[:cout] = mod.bias.detach().to(device=device, dtype=torch.float32)
        self.taps = taps
        self._offs: Dict[Tuple[int, int], "ctypes.Array"] = {}

    def offsets(self, Hp: int, Wp: int):
        key = (Hp, Wp)
        if key not in self._offs:
            self._offs[key] = (ctypes.c_int64 * len(self.taps))(*[(dt * Hp + dh) * Wp + dw for dt, dh, dw in self.taps])
        return self._offs[key]


TAPS_333 = tuple((dt - 2, dh - 1, dw - 1) for dt in range(3) for dh in range(3) for dw in range(3))      # causal 3x3x3
TAPS_311 = tuple((dt - 2, 0, 0) for dt in range(3))                                                        # causal (3,1,1)
TAPS_133 = tuple((0, dh - 1, dw - 1) for dh in range(3) for dw in range(3))                                # per-frame 3x3, pad 1
TAPS_133_DOWN = tuple((0, dh, dw) for dh in range(3) for dw in range(3))      # per-frame 3x3 after ZeroPad2d((0,1,0,1)), evaluated at stride 1
TAPS_111 = ((0, 0, 0),)


class VaeHip:
    """Executes a ``vae.WanVAENet`` on padded volumes.  ``decode_tile(z)`` / ``encode_tile(x)`` take and return what
    ``net.decode`` / ``net.encode`` do ([1, C, T, H, W] tensors)."""

    def __init__(self, net: nn.Module, device):
        self.net, self.device, self.lib = net, torch.device(device), native.lib()
        self._w: Dict[int, _ConvW] = {}
        self._g: Dict[int, torch.Tensor] = {}

    # ---- parameters -------------------------------------------------------------------------------------------------------
    # Re-laid-out parameters are cached per module OBJECT: the key is id(module), guarded by a weak reference (a collected module's
    # id can be handed to a new one) and by the parameter's version counter (a later load_state_dict writes it in place).
    def _conv_w(self, mod: nn.Module, taps) -> _ConvW:
        import weakref
        rec = self._w.get(id(mod))
        if rec is None or rec.owner() is not mod or rec.version != mod.weight._version or rec.taps != taps:
            rec = _ConvW(mod, self.device, taps, mod.weight.shape[1])
            rec.version, rec.owner = mod.weight._version, weakref.ref(mod)
            self._w[id(mod)] = rec
        return rec

    def _gamma(self, norm: nn.Module) -> torch.Tensor:
        import weakref
        g = self._g.get(id(norm))
        if g is None or g[2]() is not norm or g[1] != norm.gamma._version:
            g = (norm.gamma.detach().reshape(-1).to(device=self.device, dtype=torch.float32).contiguous(), norm.gamma._version, weakref.ref(norm))
            self._g[id(norm)] = g
        return g[0]

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---- the two kernels --------------------------------------------------------------------------------------------------
    def conv(self, x: Vol, mod: nn.Module, taps, resid: Optional[Vol] = None, first_frame: int = 0) -> Vol:
        """out = conv(x) (+ resid) on the rows of the real frames from ``first_frame`` on; a new volume with the layer's
        (padded) output channels whose padding frames are zero and whose spatial halo holds garbage."""
        rec = self._conv_w(mod, taps)
        if x.C != rec.cin:
            raise ValueError(f"convolution expects {rec.cin} (padded) input channels, the volume has {x.C}")
        out = Vol(x.T, x.H, x.W, rec.cout, self.device)
        m0, m1 = (PT + first_frame) * x.frame_rows, x.rows
        if resid is not None and (resid.C != rec.cout or resid.rows != x.rows):
            raise ValueError("residual volume does not match the convolution's output")
        native.check(self.lib.icv_conv3d_ndhwc(
            x.mat.data_ptr(), x.C, x.margin, x.margin, rec.w.data_ptr(), rec.bias.data_ptr(), rec.offsets(x.Hp, x.Wp), len(rec.taps), rec.cin,
            m0, m1, rec.cout, out.mat.data_ptr(), out.C, resid.mat.data_ptr() if resid is not None else None,
            resid.C if resid is not None else 0, self._stream()), "icv_conv3d_ndhwc")
        return out

    def norm_act(self, x: Vol, norm: nn.Module, act: int) -> Vol:
        out = Vol(x.T, x.H, x.W, x.C, self.device, zero_pads=False)
        out.buf[: out.margin].zero_(); out.buf[out.margin + out.rows:].zero_()
        native.check(self.lib.icv_rmsnorm_act_volume(x.mat.data_ptr(), out.mat.data_ptr(), self._gamma(norm).data_ptr(), x.Tp, x.Hp, x.Wp, PT, x.C,
                                                     norm.scale, 1e-12, act, self._stream()), "icv_rmsnorm_act_volume")
        return out

    # ---- blocks -----------------------------------------------------------------------------------------------------------
    def res_block(self, blk: nn.Module, x: Vol) -> Vol:
        r = blk.residual          # [RMS_norm, SiLU, conv, RMS_norm, SiLU, Dropout, conv]
        h = self.conv(self.norm_act(x, r[0], 1), r[2], TAPS_333)
        h = self.norm_act(h, r[3], 1)
        sc = x if isinstance(blk.shortcut, nn.Identity) else self.conv(x, blk.shortcut, TAPS_111)
        return self.conv(h, r[6], TAPS_333, resid=sc)

    def attn_block(self, blk: nn.Module, x: Vol) -> Vol:
        T, H, W, C = x.T, x.H, x.W, x.C
        qkv = self.conv(self.norm_act(x, blk.norm, 0), blk.to_qkv, TAPS_111).interior().reshape(T, 1, H * W, 3 * C)
        q, k, v = (t.contiguous() for t in qkv.chunk(3, dim=-1))
        y = Vol(T, H, W, C, self.device)
        y.interior().copy_(F.scaled_dot_product_attention(q, k, v).reshape(T, H, W, C))
        return self.conv(y, blk.proj, TAPS_111, resid=x)          # x + proj(attention)

    def upsample(self, rs: nn.Module, x: Vol) -> Vol:
        T, H, W, C = x.T, x.H, x.W, x.C
        if rs.mode == "upsample3d" and T > 1:
            # frame 0 is never time-convolved; frames 1.. are their own causal sequence: with frame 0 zeroed, the two frames in
            # front of frame 1 are zeros, which is exactly its causal padding
            f0 = x.vol()[PT].clone()
            x.vol()[PT].zero_()
            rest = self.conv(x, rs.time_conv, TAPS_311, first_frame=1)          # [.., 2C]: channel s*C + c -> output frame 2j + s
            T2 = 1 + 2 * (T - 1)
            up = Vol(T2, 2 * H, 2 * W, C, self.device)
            ui, ri = up.interior(), rest.interior()
            for a in (0, 1):
                for b in (0, 1):
                    ui[0, a::2, b::2] = f0[1:-1, 1:-1]
                    for s in (0, 1):
                        ui[1 + s::2, a::2, b::2] = ri[1:, :, :, s * C:(s + 1) * C]
        else:
            up = Vol(T, 2 * H, 2 * W, C, self.device)
            ui, xi = up.interior(), x.interior()
            for a in (0, 1):
                for b in (0, 1):
                    ui[:, a::2, b::2] = xi
        up.zero_halo()
        return self.conv(up, rs.resample[1], TAPS_133)

    def downsample(self, rs: nn.Module, x: Vol) -> Vol:
        T, H, W = x.T, x.H, x.W
        full = self.conv(x.zero_halo(), rs.resample[1], TAPS_133_DOWN)          # stride 1; the wanted outputs sit on the even positions
        y = Vol(T, H // 2, W // 2, full.C, self.device)
        y.interior().copy_(full.interior()[:, 0::2, 0::2])
        if rs.mode == "downsample3d" and T > 1:
            # o_j = conv3(x_2j, x_2j+1, x_2j+2): the causal form evaluated at frames 2, 4, ... ; frame 0 passes through
            tc = self.conv(y, rs.time_conv, TAPS_311, first_frame=2)         # (3,1,1): a halo only ever feeds halo outputs
            n_out = (T - 3) // 2 + 1
            z = Vol(1 + n_out, H // 2, W // 2, tc.C, self.device)
            z.interior()[0] = y.interior()[0]
            z.interior()[1:] = tc.interior()[2:2 * n_out + 1:2]
            return z
        return y

    # ---- the two networks -------------------------------------------------------------------------------------------------
    def _to_vol(self, x: torch.Tensor, cpad: int) -> Vol:
        """[1, C, T, H, W] -> volume with ``cpad`` channels (zeros above C)."""
        _, C, T, H, W = x.shape
        v = Vol(T, H, W, cpad, self.device)
        if cpad != C:
            v.buf.zero_()
        v.interior()[..., :C] = x[0].permute(1, 2, 3, 0)
        return v.zero_halo() if cpad == C else v

    def decode_tile(self, z: torch.Tensor) -> torch.Tensor:
        net, dec = self.net, self.net.decoder
        zz = z / net.inv_std + net.mean                                   # in the network's dtype, like the stock module
        w2 = net.conv2.weight.detach().float().reshape(net.z_dim, net.z_dim)       # conv2 is 1x1x1 on 16 channels: a tiny matmul
        zz = torch.einsum("oc,bcthw->bothw", w2, zz.float()) + net.conv2.bias.detach().float().view(1, -1, 1, 1, 1)
        x = self.conv(self._to_vol(zz.to(BF16), 32), dec.conv1, TAPS_333)
        x = self.res_block(dec.middle[0], x)
        x = self.attn_block(dec.middle[1], x)
        x = self.res_block(dec.middle[2], x)
        for layer in dec.upsamples:
            x = self.res_block(layer, x) if hasattr(layer, "residual") else self.upsample(layer, x)
        y = self.conv(self.norm_act(x, dec.head[0], 1), dec.head[2], TAPS_333)
        return y.interior()[..., :3].permute(3, 0, 1, 2)[None].clamp(-1, 1)

    def encode_tile(self, x: torch.Tensor) -> torch.Tensor:
        net, enc = self.net, self.net.encoder
        h = self.conv(self._to_vol(x.to(BF16), 32), enc.conv1, TAPS_333)
        for layer in enc.downsamples:
            h = self.res_block(layer, h) if hasattr(layer, "residual") else self.downsample(layer, h)
        h = self.res_block(enc.middle[0], h)
        h = self.attn_block(enc.middle[1], h)
        h = self.res_block(enc.middle[2], h)
        h = self.conv(self.norm_act(h, enc.head[0], 1), enc.head[2], TAPS_333)
        zc = net.z_dim * 2
        e = h.interior()[..., :zc].float()                                    # [T, H, W, 2z]
        w1 = net.conv1.weight.detach().float().reshape(zc, zc)
        mu = (torch.einsum("oc,thwc->othw", w1, e) + net.conv1.bias.detach().float().view(-1, 1, 1, 1)).to(BF16)[None, : net.z_dim]
        return (mu - net.mean) * net.inv_std
