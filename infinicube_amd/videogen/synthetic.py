"""Seeded synthetic weights / inputs of the Wan2.1 DiT shape (no checkpoints exist offline).

Recipe from SURVEY.md §8d: Linear weights N(0, 0.02), biases N(0, 0.01),
``modulation ~ N(0,1)/sqrt(d)``, norm affine weights 1 + N(0, 0.02), and a NON-ZERO buffer
embedder (the reference zero-inits it [R infinicube/videogen/inference.py:86-88], which would
make every conditioning test vacuous).  State-dict names follow the DiffSynth layout listed in
SURVEY.md Appendix A.1 so real checkpoints load through the same code path.
"""

from __future__ import annotations

from typing import Dict

import torch

from .config import TokenGrid, WanDiTConfig


def _randn(shape, std, seed, device, dtype):
    if torch.device(device).type == "cpu":
        g = torch.Generator(device="cpu").manual_seed(seed)
        return (torch.randn(shape, generator=g, dtype=torch.float32) * std).to(dtype)
    g = torch.Generator(device=device).manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32, device=device) * std).to(dtype)


def make_dit_state_dict(cfg: WanDiTConfig, seed: int = 0, device="cpu",
                        dtype=torch.float32, weight_std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Random DiT weights.  ``device='cpu'`` is bit-reproducible everywhere (parity tests);
    ``device='cuda'`` generates the 14B tensors directly in HBM (bench)."""
    d, f = cfg.dim, cfg.ffn_dim
    sd: Dict[str, torch.Tensor] = {}
    counter = [seed * 100003]

    def lin(name, n_out, n_in, std=weight_std):
        counter[0] += 1
        sd[f"{name}.weight"] = _randn((n_out, n_in), std, counter[0], device, dtype)
        counter[0] += 1
        sd[f"{name}.bias"] = _randn((n_out,), 0.01, counter[0], device, dtype)

    def affine(name, n, with_bias=False):
        counter[0] += 1
        sd[f"{name}.weight"] = (1.0 + _randn((n,), 0.02, counter[0], device, torch.float32)).to(dtype)
        if with_bias:
            counter[0] += 1
            sd[f"{name}.bias"] = _randn((n,), 0.01, counter[0], device, dtype)

    counter[0] += 1
    sd["patch_embedding.weight"] = _randn((d, cfg.in_dim) + tuple(cfg.patch), 0.02, counter[0], device, dtype)
    counter[0] += 1
    sd["patch_embedding.bias"] = _randn((d,), 0.01, counter[0], device, dtype)
    lin("text_embedding.0", d, cfg.text_dim)
    lin("text_embedding.2", d, d)
    lin("time_embedding.0", d, cfg.freq_dim)
    lin("time_embedding.2", d, d)
    lin("time_projection.1", 6 * d, d)
    if cfg.has_image_input:      # img_emb = LayerNorm, Linear, GELU, Linear, LayerNorm ([EXT] Wan2.1 MLPProj)
        affine("img_emb.proj.0", cfg.img_dim, with_bias=True)
        lin("img_emb.proj.1", cfg.img_dim, cfg.img_dim)
        lin("img_emb.proj.3", d, cfg.img_dim)
        affine("img_emb.proj.4", d, with_bias=True)
    for i in range(cfg.num_layers):
        p = f"blocks.{i}"
        for attn in ("self_attn", "cross_attn"):
            for proj in ("q", "k", "v", "o"):
                lin(f"{p}.{attn}.{proj}", d, d)
            affine(f"{p}.{attn}.norm_q", d)
            affine(f"{p}.{attn}.norm_k", d)
        if cfg.has_image_input:
            lin(f"{p}.cross_attn.k_img", d, d)
            lin(f"{p}.cross_attn.v_img", d, d)
            affine(f"{p}.cross_attn.norm_k_img", d)
        affine(f"{p}.norm3", d, with_bias=True)
        lin(f"{p}.ffn.0", f, d)
        lin(f"{p}.ffn.2", d, f)
        counter[0] += 1
        sd[f"{p}.modulation"] = _randn((1, 6, d), d ** -0.5, counter[0], device, dtype)
    lin("head.head", cfg.out_dim * cfg.patch_elems, d)
    counter[0] += 1
    sd["head.modulation"] = _randn((1, 2, d), d ** -0.5, counter[0], device, dtype)
    return sd


def make_buffer_embedder_state_dict(cfg: WanDiTConfig, seed: int = 7, device="cpu",
                                    dtype=torch.float32, variant: str = "concat",
                                    zero_init: bool = False) -> Dict[str, torch.Tensor]:
    """Buffer-embedder weights.  ``variant='concat'`` = hypothesis H1 of SURVEY.md §8a K1 (one
    Conv3d over the two VAE-encoded buffers concatenated on channels); ``'dual'`` = H2 (one
    Conv3d per buffer, summed)."""
    d, c = cfg.dim, cfg.buffer_channels
    std = 0.0 if zero_init else 0.02
    sd: Dict[str, torch.Tensor] = {}
    if variant == "concat":
        sd["proj.weight"] = _randn((d, 2 * c) + tuple(cfg.patch), std, seed, device, dtype)
        sd["proj.bias"] = _randn((d,), 0.0 if zero_init else 0.01, seed + 1, device, dtype)
    elif variant == "dual":
        for j, nm in enumerate(("semantic_proj", "coordinate_proj")):
            sd[f"{nm}.weight"] = _randn((d, c) + tuple(cfg.patch), std, seed + 2 * j, device, dtype)
            sd[f"{nm}.bias"] = _randn((d,), 0.0 if zero_init else 0.01, seed + 2 * j + 1, device, dtype)
    else:
        raise ValueError(f"unknown buffer embedder variant {variant!r}")
    return sd


def make_latent_noise(grid: TokenGrid, seed: int = 0, channels: int = 16) -> torch.Tensor:
    """``randn((C,T,H/8,W/8), generator=cpu.manual_seed(seed), fp32)`` — the upstream noise recipe
    (rand_device='cpu')."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn((1,) + grid.latent_shape(channels), generator=g, dtype=torch.float32)[0]


def make_text_context(cfg: WanDiTConfig, seed: int) -> torch.Tensor:
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn((cfg.text_len, cfg.text_dim), generator=g, dtype=torch.float32) * 0.1


def make_clip_features(cfg: WanDiTConfig, seed: int = 5) -> torch.Tensor:
    """Stand-in for the CLIP ViT-H/14 penultimate-layer tokens of the conditioning image [img_len, img_dim]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn((cfg.img_len, cfg.img_dim), generator=g, dtype=torch.float32)


def make_cond_latents(cfg: WanDiTConfig, grid: TokenGrid, seed: int = 6) -> torch.Tensor:
    """Stand-in for the i2v conditioning latent y [in_dim - 16, T, H/8, W/8]: 4 mask channels (first latent
    frame = 1, rest 0) over the VAE encoding of [image, zeros...]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    c = cfg.cond_channels
    y = torch.randn(grid.latent_shape(c), generator=g, dtype=torch.float32)
    y[:4] = 0.0
    y[:4, 0] = 1.0
    return y


def make_buffer_latents(cfg: WanDiTConfig, grid: TokenGrid, seed: int = 3) -> torch.Tensor:
    """Stand-in for the two VAE-encoded guidance buffers: (2*buffer_channels, T, H/8, W/8)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(grid.latent_shape(2 * cfg.buffer_channels), generator=g, dtype=torch.float32)


def make_dummy_buffers(grid: TokenGrid):
    """cfg #1 'dummy guidance buffers' (SURVEY.md §8d): constant semantic, per-channel ramps."""
    import numpy as np
    n, h, w = grid.num_frames, grid.height, grid.width
    sem = np.full((n, h, w, 3), 128, dtype=np.uint8)
    co = np.empty((n, h, w, 3), dtype=np.uint8)
    co[..., 0] = np.linspace(0, 255, w, dtype=np.float32).astype(np.uint8)[None, None, :]
    co[..., 1] = np.linspace(0, 255, h, dtype=np.float32).astype(np.uint8)[None, :, None]
    co[..., 2] = np.linspace(0, 255, n, dtype=np.float32).astype(np.uint8)[:, None, None]
    return sem, co


class PinholeStandIn:
    """The one method of the reference's camera model that the coordinate-buffer function calls
    (``get_intrinsics_matrix()`` [R infinicube/utils/buffer_utils.py:212])."""

    def __init__(self, fx, fy, cx, cy, w: int = 0, h: int = 0):
        self.k = torch.tensor([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]], dtype=torch.float32)
        self.w, self.h = w, h

    def get_intrinsics_matrix(self):
        return self.k

    def get_rays(self, height: int = None, width: int = None):
        """[H, W, 3] normalised camera rays (OpenCV axes), K^-1 (u, v, 1) / |.| like the reference's
        PinholeCamera._get_rays_impl [R infinicube/camera/pinhole.py:110-140]."""
        h, w = height or self.h, width or self.w
        v, u = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        uv1 = torch.stack([u, v, torch.ones_like(u)], -1).reshape(-1, 3)
        r = (torch.inverse(self.k) @ uv1.T).T.reshape(h, w, 3)
        return r / r.norm(dim=-1, keepdim=True)


def make_scene_maps(grid: TokenGrid, device="cpu"):
    """A synthetic street scene in the form stage 2 hands to the buffer functions
    [R infinicube/inference/guidance_buffer_generation.py:690-713]: per frame a metric depth map (0 = sky /
    infinitely far), a Waymo class-index map and a uint16 instance map, plus the pinhole camera and the
    camera-to-world poses of a camera driving forward.  Ground plane 1.6 m under the camera, building
    walls 9 m either side up to 12 m, sky above, and two box 'vehicles' that approach and drift.
    Returns (depth f32 [N,H,W], semantic i32 [N,H,W], instance i32 [N,H,W], camera, poses f32 [N,4,4])."""
    n, h, w = grid.num_frames, grid.height, grid.width
    dev = torch.device(device)
    fx = fy = 0.9 * w
    cx, cy = w / 2.0, h * 0.55
    cam = PinholeStandIn(fx, fy, cx, cy)
    v, u = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=dev), torch.arange(w, dtype=torch.float32, device=dev), indexing="ij")
    dx, dy = (u + 0.5 - cx) / fx, (v + 0.5 - cy) / fy                 # ray direction (dx, dy, 1), OpenCV axes (y down)
    inf = torch.full_like(dx, float("inf"))
    z_ground = torch.where(dy > 1e-4, 1.6 / dy.clamp(min=1e-4), inf)
    z_wall = torch.where(dx.abs() > 1e-4, 9.0 / dx.abs().clamp(min=1e-4), inf)
    wall_ok = (-(dy * z_wall)) < 12.0 - 1.6                            # the wall ends 12 m above the ground
    z_wall = torch.where(wall_ok, z_wall, inf)
    base = torch.minimum(z_ground, z_wall)
    base_sem = torch.where(z_ground <= z_wall, torch.full_like(dx, 18.0), torch.full_like(dx, 14.0))   # ROAD / BUILDING
    depth = torch.empty((n, h, w), dtype=torch.float32, device=dev)
    sem = torch.empty((n, h, w), dtype=torch.int32, device=dev)
    inst = torch.zeros((n, h, w), dtype=torch.int32, device=dev)
    poses = torch.eye(4, dtype=torch.float32).repeat(n, 1, 1)
    for i in range(n):
        d = torch.where(base > 150.0, torch.zeros_like(base), base)   # beyond the voxel world: sky (depth 0)
        s = torch.where(d > 0, base_sem, torch.zeros_like(base_sem))
        ins = torch.zeros_like(base)
        for j, (x0, z0, vz, cls) in enumerate(((-2.5, 40.0, -0.25, 1), (3.0, 25.0, 0.05, 2))):   # CAR, TRUCK
            zc = z0 + vz * i
            if zc <= 3.0:
                continue
            ul, ur = cx + fx * (x0 - 1.0) / zc, cx + fx * (x0 + 1.0) / zc
            vt, vb = cy + fy * (1.6 - 1.7) / zc, cy + fy * 1.6 / zc
            box = (u >= ul) & (u < ur) & (v >= vt) & (v < vb) & ((d == 0) | (d > zc))
            d = torch.where(box, torch.full_like(d, zc), d)
            s = torch.where(box, torch.full_like(s, float(cls)), s)
            ins = torch.where(box, torch.full_like(ins, float(j + 1)), ins)
        depth[i], sem[i], inst[i] = d, s.to(torch.int32), ins.to(torch.int32)
        poses[i, 2, 3] = 0.35 * i                                      # 3.5 m/s at 10 fps, straight ahead
        poses[i, 0, 3] = 0.4 * float(torch.sin(torch.tensor(i / 15.0)))
    return depth, sem, inst, cam, poses


def make_voxel_world(grid: TokenGrid, seed: int = 0, density: float = 150.0):
    """A synthetic VOXEL WORLD in the form stage 1 hands to stage 2 (a point cloud with Waymo class indices, z-up,
    x-front [R infinicube/utils/fvdb_utils.py:87]) plus a camera and a forward-driving trajectory: a 120 m street (road,
    sidewalks, building walls 9 m either side, poles) and two vehicles with instance ids.  `density` = points per m^2.
    Returns (points f32 [P,3], semantic i32 [P], instance i32 [P], camera with get_rays()/get_intrinsics_matrix(),
    poses f32 [N,4,4] camera-to-world)."""
    n, h, w = grid.num_frames, grid.height, grid.width
    g = torch.Generator().manual_seed(seed)

    def plane(cnt, lo, hi, fixed_axis, fixed_val):
        pts = torch.rand((cnt, 3), generator=g) * (torch.tensor(hi) - torch.tensor(lo)) + torch.tensor(lo)
        pts[:, fixed_axis] = fixed_val + 0.02 * torch.randn(cnt, generator=g)
        return pts

    L = 120.0
    parts, sems, insts = [], [], []

    def add(pts, sem, inst=0):
        parts.append(pts); sems.append(torch.full((len(pts),), sem, dtype=torch.int32)); insts.append(torch.full((len(pts),), inst, dtype=torch.int32))

    add(plane(int(density * L * 12), [-5.0, -6.0, 0.0], [L, 6.0, 0.0], 2, 0.0), 18)                 # ROAD
    add(plane(int(density * L * 3), [-5.0, 6.0, 0.0], [L, 9.0, 0.0], 2, 0.12), 22)                  # SIDEWALK (left)
    add(plane(int(density * L * 3), [-5.0, -9.0, 0.0], [L, -6.0, 0.0], 2, 0.12), 22)                # SIDEWALK (right)
    add(plane(int(density * L * 12), [-5.0, 9.0, 0.0], [L, 9.0, 12.0], 1, 9.0), 14)                 # BUILDING (left wall)
    add(plane(int(density * L * 12), [-5.0, -9.0, 0.0], [L, -9.0, 12.0], 1, -9.0), 14)              # BUILDING (right wall)
    for x in range(5, int(L), 15):                                                                  # POLE
        pole = torch.rand((400, 3), generator=g) * torch.tensor([0.2, 0.2, 5.0]) + torch.tensor([float(x), 6.3, 0.0])
        add(pole, 10)
    for j, (x0, y0, sem) in enumerate(((35.0, -2.5, 1), (22.0, 3.0, 2))):                           # CAR, TRUCK
        box = torch.rand((6000, 3), generator=g) * torch.tensor([4.2, 1.9, 1.6]) + torch.tensor([x0, y0 - 0.95, 0.0])
        add(box, sem, j + 1)
    pts, sem, inst = torch.cat(parts), torch.cat(sems), torch.cat(insts)
    cam = PinholeStandIn(0.9 * w, 0.9 * w, w / 2.0, h * 0.55, w, h)
    # camera (x right, y down, z front) -> world (x front, y left, z up); 3.5 m/s at 10 fps, a gentle lateral sway
    base = torch.tensor([[0.0, 0.0, 1.0, 0.0], [-1.0, 0.0, 0.0, 0.0], [0.0, -1.0, 0.0, 1.6], [0.0, 0.0, 0.0, 1.0]])
    poses = base.repeat(n, 1, 1)
    poses[:, 0, 3] = 0.35 * torch.arange(n)
    poses[:, 1, 3] = 0.4 * torch.sin(torch.arange(n) / 15.0)
    return pts, sem, inst, cam, poses
