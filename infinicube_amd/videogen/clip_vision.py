"""CLIP ViT-H/14 vision tower for the Wan2.1 image-to-video branch — stock PyTorch-ROCm, OUTSIDE the hot loop.

BASELINE.json config #5 ("Wan2.1-14B i2v ... image-cond branch") and the reference's checkpoint list
[R infinicube/videogen/download_checkpoint.py:24-29] name the encoder
``models_clip_open-clip-xlm-roberta-large-vit-huge-14.pth``.  The reference's ``generate()`` never passes an
image [R infinicube/videogen/inference.py:216-226], so this is only reachable through the pipeline's extra
``input_image=`` keyword.  Architecture ([EXT] open-clip ViT-H/14 as used by Wan2.1): 224x224 input, 14x14
patches -> 256 + cls = 257 tokens, width 1280, 32 pre-norm blocks of 16 heads, MLP x4 with erf-GELU; Wan
takes the hidden state after the first 31 blocks, with no final norm and no projection.  One forward on
one image per generation: not worth a kernel.  Key names follow Wan2.1's ``clip.py`` ([EXT], unverified —
no checkpoint exists offline): ``visual.{patch_embedding, cls_embedding, pos_embedding, pre_norm,
transformer.N.{norm1, attn.to_qkv, attn.proj, norm2, mlp.0, mlp.2}}``; the text tower's keys are ignored.
"""

from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class _Block(nn.Module):
    def __init__(self, dim: int, heads: int, mlp_ratio: int = 4):
        super().__init__()
        self.heads = heads
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn = nn.Module()
        self.attn.to_qkv = nn.Linear(dim, 3 * dim)
        self.attn.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.mlp = nn.Sequential(nn.Linear(dim, mlp_ratio * dim), nn.GELU(), nn.Linear(mlp_ratio * dim, dim))

    def forward(self, x):
        b, n, d = x.shape
        q, k, v = self.attn.to_qkv(self.norm1(x)).reshape(b, n, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, n, d)
        x = x + self.attn.proj(a)
        return x + self.mlp(self.norm2(x))


class ClipVisionEncoder(nn.Module):
    def __init__(self, image_size=224, patch=14, dim=1280, heads=16, layers=32, use_blocks=31):
        super().__init__()
        self.image_size, self.use_blocks = image_size, use_blocks
        n = (image_size // patch) ** 2
        self.patch_embedding = nn.Conv2d(3, dim, patch, patch, bias=False)
        self.cls_embedding = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embedding = nn.Parameter(torch.zeros(1, n + 1, dim))
        self.pre_norm = nn.LayerNorm(dim, eps=1e-5)
        self.transformer = nn.ModuleList([_Block(dim, heads) for _ in range(layers)])

    @torch.no_grad()
    def forward(self, pixels: torch.Tensor) -> torch.Tensor:
        """pixels [B, 3, H, W] in [-1, 1] -> tokens [B, 257, dim] (hidden state after ``use_blocks`` blocks)."""
        x = F.interpolate(pixels.float(), size=(self.image_size, self.image_size), mode="bicubic", align_corners=False)
        x = x * 0.5 + 0.5
        mean = torch.tensor(CLIP_MEAN, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(CLIP_STD, device=x.device).view(1, 3, 1, 1)
        x = ((x - mean) / std).to(self.patch_embedding.weight.dtype)
        # the patch embedding is a convolution whose stride equals its kernel: a reshape + one matmul computes exactly the same sums
        # and keeps MIOpen (and its first-call kernel search) out of the i2v path; the parameter stays a Conv2d weight (state-dict layout)
        w = self.patch_embedding.weight
        p, g = w.shape[-1], x.shape[-1] // w.shape[-1]
        x = x.reshape(x.shape[0], 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(x.shape[0], g * g, 3 * p * p) @ w.reshape(w.shape[0], -1).t()
        x = torch.cat([self.cls_embedding.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embedding
        x = self.pre_norm(x)
        for blk in self.transformer[: self.use_blocks]:
            x = blk(x)
        return x

    def encode_image(self, image) -> torch.Tensor:
        """PIL image -> f32 [257, dim] on the CPU (the DiT engine moves it to its device)."""
        import numpy as np
        arr = torch.from_numpy(np.asarray(image.convert("RGB"), dtype=np.float32) / 127.5 - 1.0)
        dev = self.patch_embedding.weight.device
        return self.forward(arr.permute(2, 0, 1)[None].to(dev))[0].float().cpu()


def load_clip_vision(path: str, device, dtype=torch.bfloat16, **arch) -> ClipVisionEncoder:
    """Load the ``visual.*`` half of the Wan2.1 CLIP checkpoint (strict on the blocks that are used)."""
    from .io import load_state_dict
    sd: Dict[str, torch.Tensor] = load_state_dict(path)
    vis = {k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}
    if not vis:
        raise KeyError(f"{path}: no 'visual.*' tensors — not a Wan2.1 CLIP checkpoint")
    enc = ClipVisionEncoder(**arch)
    missing, unexpected = enc.load_state_dict(vis, strict=False)
    if missing:
        raise KeyError(f"{path}: CLIP vision tower is missing {sorted(missing)[:8]} ...")
    return enc.to(device=device, dtype=dtype).eval()   # unexpected = post_norm / head: unused by Wan
