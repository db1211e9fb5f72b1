"""Wan2.1 3-D causal VAE on stock PyTorch-ROCm (outside the hot loop; north star: "Wan 3D-VAE decode
run on stock PyTorch-ROCm").

The reference names the checkpoint ``Wan2.1_VAE.pth`` [R infinicube/videogen/inference.py:69,79] and
passes ``tiled=True`` [R infinicube/videogen/inference.py:225]; the module itself lives in the absent
diffsynth fork.  This restates the PUBLIC Wan2.1 VAE ([EXT], unverifiable offline): 4x temporal / 8x
spatial compression, z_dim 16, base dim 96, dim_mult (1,2,4,4), 2 residual blocks per level, causal
3-D convolutions, RMS norms, one single-head attention block in the middle, per-channel latent
mean/std.  Module / parameter names mirror the public checkpoint layout (``encoder.downsamples.N...``,
``decoder.upsamples.N...``, ``conv1``, ``conv2``) so ``load_state_dict(strict=True)`` verifies the
restatement against real weights on first contact.

MI355X-first difference: upstream streams the video through the network one 4-frame chunk at a time
with a feature cache to fit an 80 GB GPU.  A causal convolution over the whole clip with zero left
padding is exactly the same function, so with 288 GB of HBM the whole clip is processed at once (the two
first-frame special cases of the temporal resamplers are kept: the first frame is never time-convolved).
Spatial tiling (``tiled=True``: tile 30x52 latent, stride 15x26, linear-ramp blend) is kept because it
changes the numbers (each tile sees only its own receptive field) and the reference asks for it.

MIOpen note (measured on MI355X, 93x480x832 decode, bf16, tools/aux_bench.py): with its default solver set
MIOpen answers the first call of every new 3-D conv shape with `naive_conv_ab_nonpacked_fwd_ncdhw` — 435 s for
the first decode, 17.7 s for every later one.  With the naive solver masked out it picks the CK implicit-GEMM
kernels straight away: 2.6 s first call, 2.1 s after.  The mask is an environment switch read when MIOpen
first runs a convolution, so it is set at import time here (``setdefault``: a user's own value wins).
The 2 s need MIOpen's search ("find") results; they are obtained once per process (tens of seconds with the
naive solver masked) by running the network under ``torch.backends.cudnn.flags(benchmark=True)`` — without
them MIOpen's heuristic pick stays at 10.8 s per encode / 17.6 s per decode.
"""

from __future__ import annotations

import contextlib
import glob
import os

os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")
from typing import List, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

LATENT_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
               0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
LATENT_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
              3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


class CausalConv3d(nn.Conv3d):
    """Conv3d with symmetric spatial padding and all temporal padding on the left (zeros).

    ``fold_pad`` (set by WanVAE on the GPU, ICV_VAE_PAD=conv|copy): instead of materialising the padded clip with F.pad (a full
    copy of every activation: 8 % of a decode, profiles/r04/vae_decode_kernel_stats.md) the convolution pads symmetrically
    itself - 2p frames on BOTH sides in time - and the 2p trailing output frames, the only ones that saw the right-hand zeros,
    are dropped: output frame t of the symmetric convolution reads input frames t-2p..t, so the first T frames ARE the causal
    result, bit for bit the same sums.  The slice is a prefix of the (time-major) NDHWC buffer: a view."""

    fold_pad = False

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._pad = (self.padding[2], self.padding[2], self.padding[1], self.padding[1], 2 * self.padding[0], 0)
        self._sym = (2 * self.padding[0], self.padding[1], self.padding[2])
        self.padding = (0, 0, 0)

    def forward(self, x):
        if self.fold_pad and self.stride == (1, 1, 1) and any(self._sym):
            y = F.conv3d(x, self.weight, self.bias, self.stride, self._sym, self.dilation, self.groups)
            return y[:, :, : x.shape[2]] if self._sym[0] else y
        return super().forward(F.pad(x, self._pad))


class RMS_norm(nn.Module):
    def __init__(self, dim: int, channel_first: bool = True, images: bool = True):
        super().__init__()
        bdims = (1, 1, 1) if not images else (1, 1)
        self.channel_first = channel_first
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones((dim, *bdims) if channel_first else (dim,)))

    # set by WanVAE on the GPU (ICV_VAE_NORM=hip|stock): ("lib", act) -> the norm (and the SiLU that follows it in the network, which
    # is then an Identity) run as ONE libicvideo kernel over the NDHWC rows (csrc/vae_ops.hip, icv_rmsnorm_act_rows)
    fused = None

    def forward(self, x):
        if (self.fused is not None and self.channel_first and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() in (4, 5)
                and x.is_contiguous(memory_format=torch.channels_last_3d if x.dim() == 5 else torch.channels_last)):
            from .. import native
            lib, act = self.fused
            if getattr(self, "_g32", None) is None or self._g32.device != x.device or self._g32_version != self.gamma._version:
                # fp32 copy of gamma for the kernel; re-made when the parameter is written again (a later load_state_dict)
                self._g32 = self.gamma.detach().reshape(-1).to(device=x.device, dtype=torch.float32).contiguous()
                self._g32_version = self.gamma._version
            out = torch.empty_like(x)          # same NDHWC strides
            c = x.shape[1]
            native.check(lib.icv_rmsnorm_act_rows(x.data_ptr(), out.data_ptr(), self._g32.data_ptr(), x.numel() // c, c, self.scale, 1e-12,
                                                  int(act), torch.cuda.current_stream(x.device).cuda_stream), "icv_rmsnorm_act_rows")
            return out
        y = F.normalize(x, dim=(1 if self.channel_first else -1)) * self.scale * self.gamma
        # WanVAE replaced the nn.SiLU behind this norm with an Identity when it installed the fused kernel (fused[1] == 1): the
        # activation must not depend on the layout predicate above holding (batch > 1, fp32 input, a non-NDHWC view, ...)
        return F.silu(y) if (self.fused is not None and self.fused[1] == 1) else y


class Upsample(nn.Upsample):
    """Nearest-exact 2x upsampling.  Upstream goes through fp32 (older PyTorch had no bf16 kernel); nearest sampling only
    COPIES values, so doing it in the tensor's own dtype is bit-identical and moves a third of the bytes."""

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.bfloat16:
            return super().forward(x)
        return super().forward(x.float()).type_as(x)


def _per_frame(fn, x):
    b, c, t, h, w = x.shape
    y = fn(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w))
    return y.reshape(b, t, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)


class Resample(nn.Module):
    def __init__(self, dim: int, mode: str):
        super().__init__()
        self.dim, self.mode = dim, mode
        if mode in ("upsample2d", "upsample3d"):
            self.resample = nn.Sequential(Upsample(scale_factor=(2.0, 2.0), mode="nearest-exact"),
                                          nn.Conv2d(dim, dim // 2, 3, padding=1))
            if mode == "upsample3d":
                self.time_conv = CausalConv3d(dim, dim * 2, (3, 1, 1), padding=(1, 0, 0))
        elif mode in ("downsample2d", "downsample3d"):
            self.resample = nn.Sequential(nn.ZeroPad2d((0, 1, 0, 1)), nn.Conv2d(dim, dim, 3, stride=(2, 2)))
            if mode == "downsample3d":
                self.time_conv = CausalConv3d(dim, dim, (3, 1, 1), stride=(2, 1, 1), padding=(0, 0, 0))
        else:
            raise ValueError(mode)

    def forward(self, x):
        b, c, t, h, w = x.shape
        if self.mode == "upsample3d" and t > 1:
            # frame 0 is never time-convolved; frames 1.. form their own causal sequence (zero left pad)
            rest = self.time_conv(x[:, :, 1:])                                   # [b, 2c, t-1, h, w]
            rest = rest.reshape(b, 2, c, t - 1, h, w)
            rest = torch.stack((rest[:, 0], rest[:, 1]), dim=3).reshape(b, c, 2 * (t - 1), h, w)
            x = torch.cat([x[:, :, :1], rest], dim=2)
        x = _per_frame(self.resample, x)
        if self.mode == "downsample3d" and x.shape[2] > 1:
            # frame 0 passes through; o_j = conv3(x_2j, x_2j+1, x_2j+2), stride 2, no padding
            x = torch.cat([x[:, :, :1], self.time_conv(x)], dim=2)   # padding (0,0,0): plain strided conv
        return x


class ResidualBlock(nn.Module):
    def __init__(self, in_dim: int, out_dim: int, dropout: float = 0.0):
        super().__init__()
        self.residual = nn.Sequential(
            RMS_norm(in_dim, images=False), nn.SiLU(), CausalConv3d(in_dim, out_dim, 3, padding=1),
            RMS_norm(out_dim, images=False), nn.SiLU(), nn.Dropout(dropout), CausalConv3d(out_dim, out_dim, 3, padding=1))
        self.shortcut = CausalConv3d(in_dim, out_dim, 1) if in_dim != out_dim else nn.Identity()

    def forward(self, x):
        return self.residual(x) + self.shortcut(x)


class AttentionBlock(nn.Module):
    """Single-head self-attention over the h*w positions of each frame."""

    def __init__(self, dim: int):
        super().__init__()
        self.norm = RMS_norm(dim)
        self.to_qkv = nn.Conv2d(dim, dim * 3, 1)
        self.proj = nn.Conv2d(dim, dim, 1)

    def forward(self, x):
        def attn(f):
            n, c, h, w = f.shape
            q, k, v = self.to_qkv(self.norm(f)).reshape(n, 1, c * 3, h * w).permute(0, 1, 3, 2).chunk(3, dim=-1)
            y = F.scaled_dot_product_attention(q, k, v)
            return self.proj(y.squeeze(1).permute(0, 2, 1).reshape(n, c, h, w))
        return x + _per_frame(attn, x)


class Encoder3d(nn.Module):
    def __init__(self, dim=96, z_dim=32, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_downsample=(False, True, True)):
        super().__init__()
        dims = [dim * u for u in (1,) + tuple(dim_mult)]
        self.conv1 = CausalConv3d(3, dims[0], 3, padding=1)
        layers: List[nn.Module] = []
        for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
            for _ in range(num_res_blocks):
                layers.append(ResidualBlock(in_dim, out_dim))
                in_dim = out_dim
            if i != len(dim_mult) - 1:
                layers.append(Resample(out_dim, "downsample3d" if temperal_downsample[i] else "downsample2d"))
        self.downsamples = nn.Sequential(*layers)
        self.middle = nn.Sequential(ResidualBlock(out_dim, out_dim), AttentionBlock(out_dim), ResidualBlock(out_dim, out_dim))
        self.head = nn.Sequential(RMS_norm(out_dim, images=False), nn.SiLU(), CausalConv3d(out_dim, z_dim, 3, padding=1))

    def forward(self, x):
        return self.head(self.middle(self.downsamples(self.conv1(x))))


class Decoder3d(nn.Module):
    def __init__(self, dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_upsample=(True, True, False)):
        super().__init__()
        dims = [dim * u for u in (dim_mult[-1],) + tuple(dim_mult[::-1])]
        self.conv1 = CausalConv3d(z_dim, dims[0], 3, padding=1)
        self.middle = nn.Sequential(ResidualBlock(dims[0], dims[0]), AttentionBlock(dims[0]), ResidualBlock(dims[0], dims[0]))
        layers: List[nn.Module] = []
        for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
            if i in (1, 2, 3):
                in_dim = in_dim // 2
            for _ in range(num_res_blocks + 1):
                layers.append(ResidualBlock(in_dim, out_dim))
                in_dim = out_dim
            if i != len(dim_mult) - 1:
                layers.append(Resample(out_dim, "upsample3d" if temperal_upsample[i] else "upsample2d"))
        self.upsamples = nn.Sequential(*layers)
        self.head = nn.Sequential(RMS_norm(out_dim, images=False), nn.SiLU(), CausalConv3d(out_dim, 3, 3, padding=1))

    def forward(self, z):
        return self.head(self.upsamples(self.middle(self.conv1(z))))


class WanVAENet(nn.Module):
    """Parameter layout of the public ``Wan2.1_VAE.pth``: encoder, conv1, conv2, decoder."""

    def __init__(self, dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_downsample=(False, True, True)):
        super().__init__()
        self.z_dim = z_dim
        self.encoder = Encoder3d(dim, z_dim * 2, dim_mult, num_res_blocks, temperal_downsample)
        self.conv1 = CausalConv3d(z_dim * 2, z_dim * 2, 1)
        self.conv2 = CausalConv3d(z_dim, z_dim, 1)
        self.decoder = Decoder3d(dim, z_dim, dim_mult, num_res_blocks, tuple(temperal_downsample[::-1]))
        self.register_buffer("mean", torch.tensor(LATENT_MEAN[:z_dim]).view(1, z_dim, 1, 1, 1), persistent=False)
        self.register_buffer("inv_std", (1.0 / torch.tensor(LATENT_STD[:z_dim])).view(1, z_dim, 1, 1, 1), persistent=False)

    def encode(self, x):       # [b, 3, 1+4k, H, W] in [-1,1] -> normalised mu [b, z, 1+k, H/8, W/8]
        mu, _ = self.conv1(self.encoder(x)).chunk(2, dim=1)
        return (mu - self.mean) * self.inv_std

    def decode(self, z):       # -> [b, 3, 1+4(T-1), 8h, 8w]
        return self.decoder(self.conv2(z / self.inv_std + self.mean)).clamp(-1, 1)


def _ramp_mask(h: int, w: int, bound: Tuple[bool, bool, bool, bool], border: Tuple[int, int], device, dtype):
    def one(n, left, right, b):
        m = torch.ones(n, device=device, dtype=dtype)
        if b > 0:
            r = (torch.arange(b, device=device, dtype=dtype) + 1) / b
            if not left:
                m[:b] = r
            if not right:
                m[-b:] = r.flip(0)
        return m
    mh, mw = one(h, bound[0], bound[1], border[0]), one(w, bound[2], bound[3], border[1])
    return torch.minimum(mh[:, None].expand(h, w), mw[None, :].expand(h, w))[None, None, None]


def _tile_tasks(H, W, size, stride):
    tasks = []
    for h in range(0, H, stride[0]):
        if h - stride[0] >= 0 and h - stride[0] + size[0] >= H:
            continue
        for w in range(0, W, stride[1]):
            if w - stride[1] >= 0 and w - stride[1] + size[1] >= W:
                continue
            tasks.append((h, h + size[0], w, w + size[1]))
    return tasks


class WanVAE:
    """The two-call interface the pipeline uses: ``encode(video[3,F,H,W]) -> latent[16,T,H/8,W/8]`` and
    ``decode(latent) -> video[3,F,H,W]``, optionally spatially tiled like diffsynth's WanVideoVAE."""

    accepts_uint8 = True      # encode() / encode_many() take [F, H, W, 3] uint8 clips and normalise them on the device
    supports_tile_shard = True   # encode_many() / decode() take shard=TileShard(...): tiles dealt to the ranks of a multi-rank run

    def __init__(self, net: WanVAENet, device, dtype=torch.bfloat16):
        self.net, self.device, self.dtype = net.to(device=device, dtype=dtype).eval(), device, dtype
        # weights and activations in NDHWC on the GPU: MIOpen's bf16 implicit-GEMM kernels are NHWC and otherwise transpose
        # around every convolution (measured: encode 1.28 -> 1.11 s, decode 2.09 -> 1.92 s, search 48 -> 27 s, -3.4 GiB);
        # ICV_VAE_CHANNELS_LAST=0 switches it off
        self.channels_last = os.environ.get("ICV_VAE_CHANNELS_LAST", "1") == "1" and torch.device(device).type == "cuda"
        # the causal pad folded into the convolution (CausalConv3d.fold_pad; profiles/r04/vae_layer_tuning.md: decode 1.91 -> 1.79 s,
        # encode 1.11 -> 1.07 s; ICV_VAE_PAD=copy restores the F.pad form)
        self.fold_pad = os.environ.get("ICV_VAE_PAD", "conv") == "conv"
        # the channel RMS norm + the SiLU behind it as one HIP pass over the NDHWC rows (RMS_norm.fused; profiles/r04/vae_layer_tuning.md);
        # ICV_VAE_NORM=stock keeps the composite torch ops
        self.hip_norm = os.environ.get("ICV_VAE_NORM", "hip") == "hip" and dtype == torch.bfloat16
        if self.channels_last:
            for m in self.net.modules():
                if isinstance(m, CausalConv3d):
                    m.fold_pad = self.fold_pad
            if self.hip_norm:
                from .. import native
                lib = native.lib()
                for seq in self.net.modules():
                    if isinstance(seq, nn.Sequential):
                        kids = list(seq.named_children())
                        for (n0, a), (n1, b) in zip(kids[:-1], kids[1:]):
                            if isinstance(a, RMS_norm) and isinstance(b, nn.SiLU):
                                a.fused = (lib, 1)
                                setattr(seq, n1, nn.Identity())      # no parameters: the state-dict layout is unchanged
                for m in self.net.modules():
                    if isinstance(m, RMS_norm) and m.fused is None and m.channel_first:
                        m.fused = (lib, 0)                            # the attention block's norm (no activation behind it)
            for m in self.net.modules():
                if isinstance(m, nn.Conv3d):
                    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last_3d)
                elif isinstance(m, nn.Conv2d):
                    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        # every convolution of the tile networks on libicvideo's shifted-row MFMA kernel (vae_hip.VaeHip, csrc/conv.hip) instead
        # of MIOpen: no kernel search on the first call of a process, the same kernels in every process.  ICV_VAE_CONV=miopen
        # keeps the stock convolutions (the module path above).
        self.hip = None
        if (torch.device(device).type == "cuda" and dtype == torch.bfloat16 and os.environ.get("ICV_VAE_CONV", "hip") == "hip"
                and all(m.in_channels % 32 == 0 or m.in_channels < 32 for m in self.net.modules() if isinstance(m, (nn.Conv3d, nn.Conv2d)))):
            from .vae_hip import VaeHip
            self.hip = VaeHip(self.net, device)

    def _net_encode(self, x):
        return self.hip.encode_tile(x) if self.hip is not None else self.net.encode(x)

    def _net_decode(self, z):
        return self.hip.decode_tile(z) if self.hip is not None else self.net.decode(z)

    @staticmethod
    @contextlib.contextmanager
    def _searched_kernels():
        """MIOpen find mode for the conv kernels (see the module docstring); a no-op on CPU."""
        prev = torch.backends.cudnn.benchmark
        # ICV_VAE_FIND=0: MIOpen's immediate-mode pick instead of its search - slower kernels, but the SAME kernels in every
        # process (the search's winner can differ from run to run, and with it the rounding): what the cross-process tests use
        torch.backends.cudnn.benchmark = os.environ.get("ICV_VAE_FIND", "1") == "1"
        try:
            yield
        finally:
            torch.backends.cudnn.benchmark = prev

    def encode(self, video, *args, **kwargs):
        with torch.no_grad(), self._searched_kernels():
            return self._encode([video], *args, **kwargs)[0]

    def encode_many(self, videos, *args, **kwargs):
        """Several clips of the same size in ONE pass over their tiles (the two guidance buffers): with ``shard=`` the tiles
        of all clips are dealt to the ranks together (18 tiles over 8 ranks = 3 rounds instead of 2 x 2)."""
        with torch.no_grad(), self._searched_kernels():
            return self._encode(list(videos), *args, **kwargs)

    def decode(self, latent, *args, **kwargs):
        with torch.no_grad(), self._searched_kernels():
            return self._decode(latent, *args, **kwargs)

    # ---- tiles across ranks -------------------------------------------------------------------------------------------
    # Outside the loop the only work that matters once the loop runs on N GPUs is the tiled VAE (measured on one MI355X:
    # 2.2 s for the two buffer encodes + 1.9 s for the decode against a projected 24 s loop at N = 8, and every rank was
    # repeating all of it).  The tiles are independent, so with ``shard=TileShard(...)`` task i of the tile list is computed
    # by rank i % world only and broadcast (in the network's own dtype: lossless) to the others; every rank then blends
    # the tiles in the ORIGINAL task order, so the result is bit-identical to the unsharded call on every rank.
    def _run_tiles(self, tasks, compute, shard):
        """[compute(task) for task in tasks], each computed by one rank only when ``shard`` is set."""
        if shard is None or shard.world == 1:
            return [compute(t) for t in tasks]
        mine = {i: compute(t) for i, t in enumerate(tasks) if i % shard.world == shard.rank}
        outs = []
        for i, t in enumerate(tasks):
            owner = i % shard.world
            buf = mine[i].contiguous() if owner == shard.rank else torch.empty(shard.shape_of(t), device=self.device, dtype=self.dtype)
            shard.dist.broadcast(buf, src=shard.peers[owner], group=shard.group)
            outs.append(buf)
        return outs

    def _encode(self, videos, tiled=True, tile_size=(30, 52), tile_stride=(15, 26), shard=None, **unused):
        xs = []
        for video in videos:
            if video.dtype == torch.uint8:
                # [F, H, W, 3] bytes straight from the caller (111 MB instead of a 445 MB float clip across PCIe): the same
                # fp32 `v * (2 / 255) - 1` as pipeline._video_to_tensor, on the device, then ONE rounding to the network dtype
                x = (video.to(self.device).to(torch.float32) * (2.0 / 255.0) - 1.0).permute(3, 0, 1, 2)[None].to(self.dtype)
            else:
                x = video[None].to(device=self.device, dtype=self.dtype)
            xs.append(x.contiguous(memory_format=torch.channels_last_3d) if self.channels_last else x)
        _, _, F_, H, W = xs[0].shape
        if any(x.shape != xs[0].shape for x in xs):
            raise ValueError("encode_many: the clips must have the same shape")
        T = (F_ - 1) // 4 + 1
        if not tiled:
            outs = self._run_tiles(list(range(len(xs))), lambda j: self._net_encode(xs[j]),
                                   None if shard is None else shard.with_shape(lambda j: (1, self.net.z_dim, T, H // 8, W // 8)))
            return [o[0].float() for o in outs]
        size, stride = (tile_size[0] * 8, tile_size[1] * 8), (tile_stride[0] * 8, tile_stride[1] * 8)
        boxes = _tile_tasks(H, W, size, stride)
        tasks = [(j, b) for j in range(len(xs)) for b in boxes]

        def shape_of(task):
            _, (h0, h1, w0, w1) = task
            return (1, self.net.z_dim, T, (min(h1, H) - h0) // 8, (min(w1, W) - w0) // 8)

        zs = self._run_tiles(tasks, lambda t: self._net_encode(xs[t[0]][:, :, :, t[1][0]:t[1][1], t[1][2]:t[1][3]]),
                             None if shard is None else shard.with_shape(shape_of))
        results = []
        for j in range(len(xs)):
            vals = torch.zeros((1, self.net.z_dim, T, H // 8, W // 8), device=self.device, dtype=torch.float32)
            wts = torch.zeros((1, 1, T, H // 8, W // 8), device=self.device, dtype=torch.float32)
            for (jj, (h0, h1, w0, w1)), z in zip(tasks, zs):
                if jj != j:
                    continue
                z = z.float()
                m = _ramp_mask(z.shape[3], z.shape[4], (h0 == 0, h1 >= H, w0 == 0, w1 >= W),
                               ((size[0] - stride[0]) // 8, (size[1] - stride[1]) // 8), self.device, torch.float32)
                vals[:, :, :, h0 // 8: h0 // 8 + z.shape[3], w0 // 8: w0 // 8 + z.shape[4]] += z * m
                wts[:, :, :, h0 // 8: h0 // 8 + z.shape[3], w0 // 8: w0 // 8 + z.shape[4]] += m
            results.append((vals / wts)[0])
        return results

    def _decode(self, latent, tiled=True, tile_size=(30, 52), tile_stride=(15, 26), shard=None, blend=True, **unused):
        z = latent[None].to(device=self.device, dtype=self.dtype)
        if self.channels_last:
            z = z.contiguous(memory_format=torch.channels_last_3d)
        _, _, T, H, W = z.shape
        if not tiled:
            return self._net_decode(z)[0].float()
        tasks = _tile_tasks(H, W, tile_size, tile_stride)

        def shape_of(task):
            h0, h1, w0, w1 = task
            return (1, 3, T * 4 - 3, (min(h1, H) - h0) * 8, (min(w1, W) - w0) * 8)

        ys = self._run_tiles(tasks, lambda t: self._net_decode(z[:, :, :, t[0]:t[1], t[2]:t[3]]),
                             None if shard is None else shard.with_shape(shape_of))
        if not blend:          # a rank that only contributes its tiles (WorkerPool workers)
            return None
        vals = torch.zeros((1, 3, T * 4 - 3, H * 8, W * 8), device=self.device, dtype=torch.float32)
        wts = torch.zeros((1, 1, T * 4 - 3, H * 8, W * 8), device=self.device, dtype=torch.float32)
        for (h0, h1, w0, w1), y in zip(tasks, ys):
            y = y.float()
            m = _ramp_mask(y.shape[3], y.shape[4], (h0 == 0, h1 >= H, w0 == 0, w1 >= W),
                           ((tile_size[0] - tile_stride[0]) * 8, (tile_size[1] - tile_stride[1]) * 8), self.device, torch.float32)
            vals[:, :, :, h0 * 8: h0 * 8 + y.shape[3], w0 * 8: w0 * 8 + y.shape[4]] += y * m
            wts[:, :, :, h0 * 8: h0 * 8 + y.shape[3], w0 * 8: w0 * 8 + y.shape[4]] += m
        return (vals / wts).clamp(-1, 1)[0]


class TileShard:
    """Which rank computes which VAE tile (WanVAE._run_tiles): ``rank`` of ``world`` in ``group`` (None = the default
    group) of an initialised torch.distributed (backend nccl = RCCL over xGMI on GPU ranks, gloo in the CPU tests)."""

    def __init__(self, dist, rank: int, world: int, group=None, shape_of=None):
        self.dist, self.rank, self.world, self.group, self.shape_of = dist, rank, world, group, shape_of
        self.peers = dist.get_process_group_ranks(group) if group is not None else list(range(world))

    @staticmethod
    def current(group=None):
        """The shard of this process in the live default process group, or None outside a multi-rank run / with
        ICV_VAE_SHARD=0."""
        import torch.distributed as dist
        if os.environ.get("ICV_VAE_SHARD", "1") != "1" or not (dist.is_available() and dist.is_initialized()):
            return None
        world = dist.get_world_size(group)
        return TileShard(dist, dist.get_rank(group), world, group) if world > 1 else None

    def with_shape(self, shape_of):
        return TileShard(self.dist, self.rank, self.world, self.group, shape_of)


def load_wan_vae(pattern, device, dtype=torch.bfloat16) -> WanVAE:
    files = sorted(glob.glob(pattern))
    if not files:
        raise FileNotFoundError(f"Wan VAE checkpoint not found: {pattern!r} (skip_download=True: nothing is fetched)")
    sd = torch.load(files[0], map_location="cpu", weights_only=True)
    if all(k.startswith("model.") for k in sd):
        sd = {k[len("model."):]: v for k, v in sd.items()}
    # architecture read off the tensors (dim from the first conv, z_dim from conv2; the multipliers / block counts of the
    # public Wan2.1 VAE): a mismatch still fails loudly in the strict load below
    dim, z_dim = int(sd["encoder.conv1.weight"].shape[0]), int(sd["conv2.weight"].shape[0])
    net = WanVAENet(dim=dim, z_dim=z_dim)
    net.load_state_dict(sd, strict=True)
    return WanVAE(net, device, dtype)
