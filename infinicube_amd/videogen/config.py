"""Wan2.1 DiT shape configuration and token-grid arithmetic.

Model dimensions are the public Wan2.1 ones (tagged [EXT] there: the
reference delegates the model to an un-vendored ``diffsynth`` fork, pyproject.toml:71).  The
resolution table mirrors [R infinicube/inference/guidance_buffer_generation.py:79-82] and the
93-frame cap mirrors [R infinicube/inference/guidance_buffer_generation.py:744-745].
"""

from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Tuple

HEAD_DIM = 128  # both Wan2.1 sizes; the HIP attention kernel is specialised on it


@dataclass(frozen=True)
class WanDiTConfig:
    name: str
    dim: int
    ffn_dim: int
    num_heads: int
    num_layers: int
    in_dim: int = 16
    out_dim: int = 16
    text_dim: int = 4096
    text_len: int = 512
    freq_dim: int = 256
    eps: float = 1e-6
    patch: Tuple[int, int, int] = (1, 2, 2)
    buffer_channels: int = 16  # per guidance buffer (semantic, coordinate) after VAE encode
    # image-to-video branch (BASELINE.json config #5; [EXT] Wan2.1 i2v): in_dim = 16 noise + 4 mask + 16
    # first-frame latent channels, and a second cross-attention over img_len CLIP tokens of width img_dim
    img_dim: int = 0           # 0 = text-to-video (no image branch); 1280 for Wan2.1-I2V (CLIP ViT-H/14)
    img_len: int = 257

    @property
    def has_image_input(self) -> bool:
        return self.img_dim > 0

    @property
    def cond_channels(self) -> int:
        """Channels of the step-invariant conditioning latent y concatenated under the noise (i2v: 20)."""
        return self.in_dim - self.out_dim

    @property
    def head_dim(self) -> int:
        return self.dim // self.num_heads

    @property
    def patch_elems(self) -> int:
        return self.patch[0] * self.patch[1] * self.patch[2]

    def validate(self) -> "WanDiTConfig":
        if self.dim % self.num_heads or self.head_dim != HEAD_DIM:
            raise ValueError(f"head_dim must be {HEAD_DIM}, got dim={self.dim} heads={self.num_heads}")
        if self.dim % 64 or self.ffn_dim % 64:
            raise ValueError("dim and ffn_dim must be multiples of 64")
        if self.in_dim < self.out_dim:
            raise ValueError(f"in_dim {self.in_dim} < out_dim {self.out_dim}")
        return self


WAN_1_3B = WanDiTConfig("wan2.1-t2v-1.3b", dim=1536, ffn_dim=8960, num_heads=12, num_layers=30)
WAN_14B = WanDiTConfig("wan2.1-t2v-14b", dim=5120, ffn_dim=13824, num_heads=40, num_layers=40)
# Small shapes for parity tests: same head_dim (128) so the same HIP kernels run.
WAN_TINY = WanDiTConfig("wan-tiny", dim=256, ffn_dim=512, num_heads=2, num_layers=2,
                        text_dim=128, text_len=32, freq_dim=64)
WAN_SMALL = WanDiTConfig("wan-small", dim=512, ffn_dim=1408, num_heads=4, num_layers=3,
                         text_dim=256, text_len=64)

WAN_14B_I2V = replace(WAN_14B, name="wan2.1-i2v-14b", in_dim=36, img_dim=1280)
WAN_TINY_I2V = replace(WAN_TINY, name="wan-tiny-i2v", in_dim=36, img_dim=64, img_len=257)

PRESETS = {"1.3b": WAN_1_3B, "14b": WAN_14B, "tiny": WAN_TINY, "small": WAN_SMALL,
           "14b-i2v": WAN_14B_I2V, "tiny-i2v": WAN_TINY_I2V}


def preset(name: str, **overrides) -> WanDiTConfig:
    cfg = PRESETS[name.lower()]
    return replace(cfg, **overrides).validate() if overrides else cfg.validate()


def infer_config_from_state_dict(sd) -> WanDiTConfig:
    """Recover the model size from checkpoint tensor shapes (diffsynth does the same for
    1.3B vs 14B)."""
    dim = int(sd["patch_embedding.weight"].shape[0])
    in_dim = int(sd["patch_embedding.weight"].shape[1])
    ffn = int(sd["blocks.0.ffn.0.weight"].shape[0])
    layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    text_dim = int(sd["text_embedding.0.weight"].shape[1])
    freq_dim = int(sd["time_embedding.0.weight"].shape[1])
    out_dim = int(sd["head.head.weight"].shape[0]) // 4
    img_dim = int(sd["img_emb.proj.1.weight"].shape[1]) if "img_emb.proj.1.weight" in sd else 0
    for c in (WAN_1_3B, WAN_14B):
        if (c.dim, c.ffn_dim, c.num_layers) == (dim, ffn, layers):
            name = c.name.replace("t2v", "i2v") if img_dim else c.name
            return replace(c, name=name, in_dim=in_dim, out_dim=out_dim, text_dim=text_dim, freq_dim=freq_dim,
                           img_dim=img_dim)
    return WanDiTConfig(f"wan-d{dim}-l{layers}", dim=dim, ffn_dim=ffn, num_heads=dim // HEAD_DIM,
                        num_layers=layers, in_dim=in_dim, out_dim=out_dim, text_dim=text_dim,
                        freq_dim=freq_dim, img_dim=img_dim).validate()


@dataclass(frozen=True)
class TokenGrid:
    """Latent/token geometry of one generation: frames -> latent (T,H8,W8) -> tokens (T,Hp,Wp)."""
    num_frames: int
    height: int
    width: int

    def __post_init__(self):
        if self.num_frames % 4 != 1:
            raise ValueError(f"num_frames must be 1 mod 4, got {self.num_frames}")
        if self.height % 16 or self.width % 16:
            raise ValueError(f"height/width must be multiples of 16, got {self.height}x{self.width}")

    @property
    def T(self) -> int:
        return (self.num_frames - 1) // 4 + 1

    @property
    def latent_hw(self) -> Tuple[int, int]:
        return self.height // 8, self.width // 8

    @property
    def Hp(self) -> int:
        return self.height // 16

    @property
    def Wp(self) -> int:
        return self.width // 16

    @property
    def tokens_per_frame(self) -> int:
        return self.Hp * self.Wp

    @property
    def S(self) -> int:
        return self.T * self.Hp * self.Wp

    def latent_shape(self, channels: int = 16) -> Tuple[int, int, int, int]:
        h8, w8 = self.latent_hw
        return (channels, self.T, h8, w8)


# BASELINE.json configs: #1 = 17 f 256x448; #2-#4 = 93 f 480x832; #5 = 93 f 720x1280.
GRID_CFG1 = TokenGrid(17, 256, 448)
GRID_480P = TokenGrid(93, 480, 832)
GRID_720P = TokenGrid(93, 720, 1280)


def dit_forward_flops(cfg: WanDiTConfig, S: int) -> float:
    """Algorithmic FLOPs of one DiT forward (BASELINE.md §2 / SURVEY.md §8d formula)."""
    d, f, L, Tx = cfg.dim, cfg.ffn_dim, cfg.num_layers, cfg.text_len
    per_layer = (8.0 * S * d * d + 4.0 * S * S * d + 4.0 * S * d * d
                 + 4.0 * Tx * d * d + 4.0 * S * Tx * d + 4.0 * S * d * f)
    if cfg.has_image_input:   # i2v: K/V projection of and attention over the CLIP tokens
        per_layer += 4.0 * cfg.img_len * d * d + 4.0 * S * cfg.img_len * d
    return float(L) * per_layer
