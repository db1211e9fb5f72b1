"""Semantic / instance colour guidance buffer on MI355X.

Same names, arguments and results as the reference (`semantic_to_color`, `generate_rgb_semantic_buffer`
[R infinicube/utils/semantic_utils.py:88-131]; instance colours as `coloring_instance_map`
[R infinicube/utils/instance_utils.py:96-143]); the per-pixel look-ups run in libicvideo HIP kernels
(csrc/buffers.hip).  Tables: WAYMO_MAPPING is the reference's class -> palette-slot table
[R infinicube/utils/semantic_utils.py:21-59]; WAYMO_PALETTE follows its recipe (Set2 / Set3 / Set1 / Paired
qualitative colormaps) through matplotlib — the reference goes through `pycg.color.get_cmap_array`, ASSUMED to
return matplotlib's listed colours.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import native

WAYMO_CATEGORY_NAMES = ["UNDEFINED", "CAR", "TRUCK", "BUS", "OTHER_VEHICLE", "MOTORCYCLIST", "BICYCLIST", "PEDESTRIAN",
                        "SIGN", "TRAFFIC_LIGHT", "POLE", "CONSTRUCTION_CONE", "BICYCLE", "MOTORCYCLE", "BUILDING",
                        "VEGETATION", "TREE_TRUNK", "CURB", "ROAD", "LANE_MARKER", "OTHER_GROUND", "WALKABLE", "SIDEWALK"]
# class index -> palette slot (0 signs/lights/cones, 1 riders+pedestrians, 2 walkable, 3 vehicles, 4 vegetation,
# 5 curb/lane marker, 6 building, 7 road/ground, 8 undefined, 9 pole)
WAYMO_MAPPING = np.array([8, 3, 3, 3, 3, 1, 1, 1, 0, 0, 9, 0, 1, 1, 6, 4, 4, 5, 7, 5, 7, 2, 2], dtype=np.int32)


def _build_palette() -> np.ndarray:
    import matplotlib
    cm = lambda n: np.array(matplotlib.colormaps[n].colors, np.float32)   # noqa: E731
    pal = np.zeros((10, 3), dtype=np.float32)
    pal[:8] = cm("Set2")
    pal[3], pal[4], pal[8], pal[9] = cm("Set3")[9], cm("Set1")[2], cm("Paired")[1], cm("Set3")[10]
    return pal


WAYMO_PALETTE = _build_palette()


def _dev():
    if not torch.cuda.is_available():
        raise native.NativeError("semantic buffer: no GPU visible to PyTorch-ROCm; there is no CPU fallback")
    return torch.device("cuda:0")


def semantic_to_color(semantics):
    """semantics: int array/tensor of class indices (any shape) -> float32 colours [..., 3] in [0,1] (numpy)."""
    lib, dev = native.lib(), _dev()
    # a CUDA tensor (e.g. the voxel renderer's class map) stays on the device: no host round trip before the kernel
    sem = semantics if isinstance(semantics, torch.Tensor) else torch.as_tensor(np.asarray(semantics))
    shape = tuple(sem.shape)
    sem_d = sem.to(dev, torch.int32).contiguous().reshape(-1)
    lut = torch.from_numpy(WAYMO_PALETTE[WAYMO_MAPPING]).to(dev).contiguous()
    out = torch.empty((sem_d.numel(), 3), dtype=torch.float32, device=dev)
    native.check(lib.icv_semantic_to_color(sem_d.data_ptr(), sem_d.numel(), lut.data_ptr(), lut.shape[0], out.data_ptr(),
                                           None, torch.cuda.current_stream(dev).cuda_stream), "icv_semantic_to_color")
    return out.reshape(shape + (3,)).cpu().numpy()


def create_instance_mapping(unique_instance_ids, color_map_for_vechile="PuRd", color_map_for_pedestrian="YlOrBr"):
    """Random colormap sample per instance id; same RNG call order as the reference (vehicles, then pedestrians)."""
    import matplotlib as mpl
    ids = np.asarray(unique_instance_ids)
    veh, ped = ids[ids < 2 ** 15], ids[ids >= 2 ** 15]
    out = {}
    for group, cmap in ((veh, mpl.colormaps[color_map_for_vechile]), (ped, mpl.colormaps[color_map_for_pedestrian])):
        for i, x in zip(group, np.random.rand(len(group))):
            out[i] = cmap(x)[:3]
    return out


def generate_rgb_semantic_buffer(semantics_rgb: np.ndarray, instance_buffer) -> np.ndarray:
    """semantics_rgb uint8 [N,H,W,3], instance_buffer uint16 [N,H,W] -> uint8 [N,H,W,3]: instance colour where
    instance > 0, semantic colour elsewhere."""
    lib, dev = native.lib(), _dev()
    if isinstance(instance_buffer, torch.Tensor):          # device-resident ids stay there (only the few unique ids cross PCIe)
        inst = instance_buffer.to(dev, torch.int32).contiguous()
    else:
        inst = torch.from_numpy(np.asarray(instance_buffer).astype(np.int32)).to(dev).contiguous()
    ids = torch.unique(inst).cpu().numpy()
    table = create_instance_mapping(ids[ids != 0])
    lut = np.zeros((65536, 3), dtype=np.uint8)
    for iid, c in table.items():
        lut[int(iid) & 0xffff] = (np.array(c) * 255).astype(np.uint8)
    sem = torch.from_numpy(np.ascontiguousarray(semantics_rgb, dtype=np.uint8)).to(dev)
    out = torch.empty_like(sem)
    lut_d = torch.from_numpy(lut).to(dev)      # kept in a local: the launch below is asynchronous
    native.check(lib.icv_instance_overlay_u8(sem.data_ptr(), inst.data_ptr(), inst.numel(), lut_d.data_ptr(),
                                             out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "icv_instance_overlay_u8")
    torch.cuda.synchronize(dev)
    return out.cpu().numpy()
