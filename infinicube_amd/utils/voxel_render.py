"""Voxel world -> depth / semantic / instance guidance buffers on MI355X (second half).

What stage 2 does through fVDB for every frame [R infinicube/utils/fvdb_utils.py:572-605]:
    grid, attrs = points_to_fvdb(points, grid_to_world, attrs={"semantics", "instance"}, voxel_sizes, origins = vs / 2)
    depth    = camera_model.get_zdepth_map_from_voxel(pose, grid)               [R infinicube/camera/base.py:520-571]
    semantic = camera_model.get_semantic_map_from_voxel(pose, grid, attrs[...]) [R infinicube/camera/base.py:573-619]
here without fVDB (an absent, un-vendored CUDA wheel): `points_to_voxels` (same rounding and "argmax-category"
attribute reduction), a dense int32 index volume in HBM (`VoxelVolume`) and libicvideo's ray-cast kernel
(csrc/voxels.hip).  MI355X-first differences from the reference's flow: a static scene is voxelised ONCE and all N
poses are cast in one launch (the reference rebuilds the grid per frame because dynamic objects move; pass per-frame
point sets to `render_frames` for that case).  No CPU fallback.  Assembling object point sets from CAD meshes
(`fvdb.gridbatch_from_mesh`) stays with the caller.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import native

EPS_DEPTH, EPS_VOXEL = 1e-1, 1e-2   # the reference's `segments_along_rays(eps=1e-1)` / `voxels_along_rays(eps=1e-2)`
MAX_VOLUME_BYTES = 64 << 30         # refuse absurd bounding boxes (a 64 GiB index volume)


def points_to_voxels(points: torch.Tensor, attrs: Optional[Dict[str, torch.Tensor]] = None,
                     voxel_sizes: Sequence[float] = (0.2, 0.2, 0.2), origins: Optional[Sequence[float]] = None
                     ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """points [P,3] (any device) -> (ijk int32 [M,3] unique occupied voxels, {name: int32 [M]}).
    ijk = round((p - origin) / voxel_size) (round-half-even, like the reference's `.round()`), origin = voxel_size / 2
    unless given; attribute of a voxel = the most frequent category among its points, ties -> the smallest category."""
    dev = points.device
    vs = torch.tensor(list(voxel_sizes), dtype=torch.float32, device=dev)
    org = vs / 2 if origins is None else torch.tensor(list(origins), dtype=torch.float32, device=dev)
    ijk = torch.round((points.to(torch.float32) - org) / vs).to(torch.int64)
    lo = ijk.amin(0)
    ext = ijk.amax(0) - lo + 1
    lin = ((ijk[:, 2] - lo[2]) * ext[1] + (ijk[:, 1] - lo[1])) * ext[0] + (ijk[:, 0] - lo[0])
    vox, inv = torch.unique(lin, return_inverse=True)                      # sorted by (k, j, i)
    out_ijk = torch.stack([vox % ext[0] + lo[0], (vox // ext[0]) % ext[1] + lo[1], vox // (ext[0] * ext[1]) + lo[2]], 1).to(torch.int32)
    out_attrs: Dict[str, torch.Tensor] = {}
    V = vox.numel()
    for name, a in (attrs or {}).items():
        cats, rank = torch.unique(a.to(dev).reshape(-1).to(torch.int64), return_inverse=True)   # sorted categories
        C = cats.numel()
        key, cnt = torch.unique(inv * C + rank, return_counts=True)
        v, r = key // C, key % C
        score = cnt * C + (C - 1 - r)                      # max count first, then the smallest category
        best = torch.full((V,), -1, dtype=torch.int64, device=dev).scatter_reduce(0, v, score, "amax", include_self=True)
        out_attrs[name] = cats[C - 1 - best % C].to(torch.int32)
    return out_ijk.contiguous(), out_attrs


@dataclass
class VoxelVolume:
    """Dense index volume of a voxel world resident in HBM: vol[z][y][x] = voxel index or -1, one occupancy byte per
    8^3 brick; `vol_min` = ijk of cell (0,0,0), dims multiples of 8 with >= 8 empty cells around the occupied box."""
    vol: torch.Tensor
    bricks: torch.Tensor
    vol_min: np.ndarray
    dims: np.ndarray
    voxel_sizes: np.ndarray
    n_voxels: int

    @staticmethod
    def build(ijk: torch.Tensor, voxel_sizes: Sequence[float] = (0.2, 0.2, 0.2), device="cuda:0", pad: int = 8) -> "VoxelVolume":
        lib = native.lib()
        if not torch.cuda.is_available():
            raise native.NativeError("VoxelVolume: no GPU visible to PyTorch-ROCm; there is no CPU fallback")
        dev = torch.device(device)
        ijk = ijk.to(device=dev, dtype=torch.int32).contiguous()
        lo = ijk.amin(0).cpu().numpy().astype(np.int64) - pad
        lo = (np.floor(lo / 8.0) * 8).astype(np.int64)
        hi = ijk.amax(0).cpu().numpy().astype(np.int64)
        dims = ((hi + pad + 1 - lo + 7) // 8 * 8).astype(np.int64)
        nbytes = int(dims.prod()) * 4
        if nbytes > MAX_VOLUME_BYTES:
            raise MemoryError(f"voxel bounding box {tuple(dims)} needs a {nbytes / 2**30:.1f} GiB index volume; crop the scene")
        vol = torch.full((int(dims[2]), int(dims[1]), int(dims[0])), -1, dtype=torch.int32, device=dev)
        bricks = torch.zeros((int(dims[2]) // 8, int(dims[1]) // 8, int(dims[0]) // 8), dtype=torch.uint8, device=dev)
        c3 = lambda a: (ctypes.c_int * 3)(*[int(x) for x in a])   # noqa: E731
        native.check(lib.icv_voxel_scatter(ijk.data_ptr(), ijk.shape[0], c3(lo), c3(dims), vol.data_ptr(), bricks.data_ptr(),
                                           torch.cuda.current_stream(dev).cuda_stream), "icv_voxel_scatter")
        return VoxelVolume(vol, bricks, lo.astype(np.int32), dims.astype(np.int32), np.asarray(voxel_sizes, np.float32), int(ijk.shape[0]))

    def raycast(self, rays_cam: torch.Tensor, poses: torch.Tensor, attr0: Optional[torch.Tensor] = None,
                attr1: Optional[torch.Tensor] = None, background0: int = 0, background1: int = 0,
                eps_depth: float = EPS_DEPTH, eps_voxel: float = EPS_VOXEL, want_index: bool = False):
        """rays_cam [H,W,3] normalised camera rays (`camera_model.get_rays()`), poses [N,4,4] camera-to-world ->
        (zdepth f32 [N,H,W], attr0 map i32 [N,H,W] | None, attr1 map | None[, hit index i32 [N,H,W]])."""
        lib = native.lib()
        dev = self.vol.device
        h, w = rays_cam.shape[:2]
        rays = rays_cam.to(device=dev, dtype=torch.float32).reshape(h * w, 3).contiguous()
        p = poses.to(device=dev, dtype=torch.float32).reshape(-1, 16).contiguous()
        n = p.shape[0]
        depth = torch.empty((n, h, w), dtype=torch.float32, device=dev)
        mk = lambda a: None if a is None else torch.empty((n, h, w), dtype=torch.int32, device=dev)   # noqa: E731
        a0 = None if attr0 is None else attr0.to(device=dev, dtype=torch.int32).contiguous()
        a1 = None if attr1 is None else attr1.to(device=dev, dtype=torch.int32).contiguous()
        for a in (a0, a1):
            if a is not None and a.numel() != self.n_voxels:
                raise ValueError(f"attribute has {a.numel()} entries, the volume {self.n_voxels} voxels")
        o0, o1 = mk(a0), mk(a1)
        oi = torch.empty((n, h, w), dtype=torch.int32, device=dev) if want_index else None
        # low corner of cell (0,0,0): origin - vs/2 = vol_min * vs (float64 product rounded once, like the oracle)
        glo = (self.vol_min.astype(np.float64) * self.voxel_sizes.astype(np.float64)).astype(np.float32)
        c3 = lambda a: (ctypes.c_int * 3)(*[int(x) for x in a])       # noqa: E731
        f3 = lambda a: (ctypes.c_float * 3)(*[float(x) for x in a])   # noqa: E731
        native.check(lib.icv_voxel_raycast(
            self.vol.data_ptr(), self.bricks.data_ptr(), c3(self.dims), f3(glo), f3(self.voxel_sizes), rays.data_ptr(), p.data_ptr(),
            n, h * w, float(eps_depth), float(eps_voxel), native.ptr(a0), native.ptr(a1), int(background0), int(background1),
            depth.data_ptr(), native.ptr(o0), native.ptr(o1), native.ptr(oi), torch.cuda.current_stream(dev).cuda_stream),
            "icv_voxel_raycast")
        return (depth, o0, o1, oi) if want_index else (depth, o0, o1)


def render_voxel_buffers(camera_model, camera_poses_in_world: torch.Tensor, scene_points: torch.Tensor,
                         scene_semantic: torch.Tensor, scene_instance: Optional[torch.Tensor] = None,
                         voxel_sizes: Sequence[float] = (0.2, 0.2, 0.2), device="cuda:0"):
    """(depth [N,H,W] f32 z-depth with 0 = no hit, semantic [N,H,W] i32 with 0 = UNDEFINED/sky, instance [N,H,W] i32)
    of a STATIC point-cloud scene for all poses — the three fVDB-bound calls of the reference's per-frame loop
    [R infinicube/utils/fvdb_utils.py:572-605] with one voxelisation and one ray-cast launch.  `camera_model` needs
    `get_rays()` -> [H,W,3] normalised camera rays, like the reference's PinholeCamera."""
    single = camera_poses_in_world.dim() == 2
    poses = camera_poses_in_world[None] if single else camera_poses_in_world
    dev = torch.device(device)
    if scene_points.numel() == 0:        # an empty world: every ray misses (depth 0, UNDEFINED, no instance)
        h, w = camera_model.get_rays().shape[:2]
        z = torch.zeros((poses.shape[0], h, w), dtype=torch.int32, device=dev)
        out = (z.to(torch.float32), z, z.clone())
        return tuple(o[0] for o in out) if single else out
    attrs = {"semantics": scene_semantic}
    if scene_instance is not None:
        attrs["instance"] = scene_instance
    ijk, vattrs = points_to_voxels(scene_points.to(dev), {k: v.to(dev) for k, v in attrs.items()}, voxel_sizes)
    volume = VoxelVolume.build(ijk, voxel_sizes, dev)
    depth, sem, inst = volume.raycast(camera_model.get_rays(), poses, vattrs["semantics"], vattrs.get("instance"))
    if inst is None:
        inst = torch.zeros_like(sem)
    return (depth[0], sem[0], inst[0]) if single else (depth, sem, inst)


def render_frames(camera_model, camera_poses_in_world: torch.Tensor, per_frame_points: Sequence[torch.Tensor],
                  per_frame_semantic: Sequence[torch.Tensor], per_frame_instance: Sequence[torch.Tensor],
                  voxel_sizes: Sequence[float] = (0.2, 0.2, 0.2), device="cuda:0"):
    """The reference's flow for scenes with moving objects: frame f renders its OWN point set (static scene + that
    frame's object points, concatenated by the caller as the reference does [R fvdb_utils.py:556-570])."""
    outs = [render_voxel_buffers(camera_model, camera_poses_in_world[f], per_frame_points[f], per_frame_semantic[f],
                                 per_frame_instance[f], voxel_sizes, device) for f in range(len(per_frame_points))]
    return tuple(torch.stack([o[i] for o in outs], 0) for i in range(3))
