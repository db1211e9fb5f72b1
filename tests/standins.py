"""Deterministic stand-ins for the components OUTSIDE the hot loop (UMT5 text encoder, Wan-VAE, CLIP image tower)
so the full ``generate()`` path can be exercised without the 11 GB / 0.5 GB checkpoints (none exist
offline).  They have the real components' interfaces and tensor geometry (4x temporal / 8x spatial
compression, 16 latent channels, [text_len, text_dim] context) and no learned meaning.  Used by the
tests, ``__graft_entry__.smoke`` and examples — never selected silently: ``from_pretrained`` loads
real files or raises."""

from __future__ import annotations

import hashlib

import torch
import torch.nn.functional as F

from infinicube_amd.videogen.config import WanDiTConfig


class HashTextEncoder:
    """prompt -> seeded pseudo-embedding [text_len, text_dim]; rows past the 'token count' are zero,
    like the zero-padded UMT5 context."""

    def __init__(self, cfg: WanDiTConfig):
        self.text_len, self.text_dim = cfg.text_len, cfg.text_dim

    def encode(self, prompt: str) -> torch.Tensor:
        seed = int.from_bytes(hashlib.sha256(prompt.encode("utf-8")).digest()[:4], "little")
        g = torch.Generator().manual_seed(seed)
        n_tok = max(1, min(self.text_len, len(prompt.split()) + 1))
        out = torch.zeros((self.text_len, self.text_dim), dtype=torch.float32)
        out[:n_tok] = torch.randn((n_tok, self.text_dim), generator=g) * 0.1
        return out


class HashImageEncoder:
    """PIL image -> seeded pseudo CLIP tokens [img_len, img_dim] (i2v branch), a function of the pixels."""

    def __init__(self, cfg: WanDiTConfig):
        self.img_len, self.img_dim = cfg.img_len, cfg.img_dim

    def encode_image(self, image) -> torch.Tensor:
        seed = int.from_bytes(hashlib.sha256(image.convert("RGB").tobytes()).digest()[:4], "little")
        g = torch.Generator().manual_seed(seed)
        return torch.randn((self.img_len, self.img_dim), generator=g)


class PoolVAE:
    """video [3, F, H, W] in [-1,1] <-> latent [16, (F-1)/4+1, H/8, W/8] by average pooling and a fixed
    random 3 <-> 16 channel map (first frame alone, then groups of 4, like the causal Wan-VAE)."""

    z_dim = 16

    def __init__(self, seed: int = 11):
        g = torch.Generator().manual_seed(seed)
        self.enc = torch.randn((16, 3), generator=g)
        self.dec = torch.linalg.pinv(self.enc)

    def encode(self, video: torch.Tensor, tiled: bool = True, **unused) -> torch.Tensor:
        c, f, h, w = video.shape
        assert c == 3 and f % 4 == 1 and h % 8 == 0 and w % 8 == 0
        x = F.avg_pool2d(video.float().cpu().permute(1, 0, 2, 3), 8)               # [F, 3, h8, w8]
        t = torch.cat([x[:1], x[1:].reshape((f - 1) // 4, 4, 3, h // 8, w // 8).mean(1)], 0)
        return torch.einsum("zc,tchw->zthw", self.enc, t).contiguous()

    def decode(self, latent: torch.Tensor, tiled: bool = True, **unused) -> torch.Tensor:
        z = latent.float().cpu()
        t = torch.einsum("cz,zthw->tchw", self.dec, z)                              # [T, 3, h8, w8]
        frames = torch.cat([t[:1], t[1:].repeat_interleave(4, dim=0)], 0)
        up = F.interpolate(frames, scale_factor=8, mode="nearest")
        return up.permute(1, 0, 2, 3).clamp(-1, 1).contiguous()
