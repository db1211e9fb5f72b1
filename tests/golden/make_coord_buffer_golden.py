#!/usr/bin/env python3
"""Generates tests/golden/coord_buffer_cases.npz by running the REFERENCE functions
`generate_coordinate_buffer_from_memory_global_norm` [R infinicube/utils/buffer_utils.py:180-265] and
`unproject_depth_torch` [R infinicube/utils/depth_utils.py:402-466], imported from /root/reference with
stub modules for the packages this container lacks; nothing is copied —
only input/output arrays are saved.  The reference draws its quantile sample with an UNSEEDED
torch.randperm; here torch.manual_seed fixes it (case 'big' has > 100000 valid points, so the sample
matters there).  Run in the build container only:  python tests/golden/make_coord_buffer_golden.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "coord_buffer_cases.npz")


class _Auto(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {})


def _ns(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m


def import_reference():
    _ns("infinicube", f"{REF}/infinicube")
    _ns("infinicube.utils", f"{REF}/infinicube/utils")
    _ns("infinicube.camera", f"{REF}/infinicube/camera")
    # pycg is absent: the palette helper is only needed for the module to import (semantic palette, unused
    # by the coordinate buffer); stand-in backed by matplotlib (an assumption)
    import matplotlib
    pycg, color = types.ModuleType("pycg"), types.ModuleType("pycg.color")
    color.get_cmap_array = lambda n: np.array(matplotlib.colormaps[n].colors, np.float32)
    pycg.color = color
    sys.modules.update({"pycg": pycg, "pycg.color": color})
    for _ in range(20):
        try:
            importlib.import_module("infinicube.camera.pinhole")
            importlib.import_module("infinicube.utils.depth_utils")
            return importlib.import_module("infinicube.utils.buffer_utils")
        except ModuleNotFoundError as e:
            sys.modules[e.name] = _Auto(e.name)
    raise RuntimeError("could not import the reference buffer utilities")


def synthetic_scene(n, h, w, seed):
    """Ground plane + two boxes + sky rows (depth 0 = infinitely far), a camera moving forward."""
    g = np.random.default_rng(seed)
    fx = fy = 0.9 * w
    cx, cy = w / 2.0, h / 2.0
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    depth = np.zeros((n, h, w), np.float32)
    for i in range(n):
        ray_y = (ys - cy) / fy
        d = np.where(ray_y > 0.02, 1.6 / np.maximum(ray_y, 1e-3), 0.0)            # ground 1.6 m below the camera
        d = np.minimum(d, 80.0) * (ray_y > 0.02)
        for (bx, by, bw, bh, bd) in ((0.3, 0.45, 0.15, 0.25, 12.0 - 1.5 * i), (0.6, 0.5, 0.1, 0.2, 25.0 - 1.5 * i)):
            m = (xs > bx * w) & (xs < (bx + bw) * w) & (ys > by * h) & (ys < (by + bh) * h)
            d = np.where(m, bd, d)
        depth[i] = d + (d > 0) * g.normal(0, 0.01, d.shape).astype(np.float32)
    poses = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    for i in range(n):
        yaw = 0.01 * i
        poses[i, :3, :3] = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]], np.float32)
        poses[i, :3, 3] = [0.05 * i, 0.0, 1.5 * i]
    poses = poses @ np.diag([1, 1, 1, 1]).astype(np.float32)
    world0 = np.array([[0.99, 0.05, 0.0, 3.0], [-0.05, 0.99, 0.0, -2.0], [0, 0, 1, 0.5], [0, 0, 0, 1]], np.float32)
    return depth, np.einsum("ij,njk->nik", world0, poses).astype(np.float32), (fx, fy, cx, cy)


def main():
    bu = import_reference()
    pin = sys.modules["infinicube.camera.pinhole"]
    out = {}
    for name, (n, h, w, seed) in {"small": (3, 48, 64, 0), "allsky": (2, 16, 16, 1), "big": (4, 192, 320, 2)}.items():
        depth, poses, (fx, fy, cx, cy) = synthetic_scene(n, h, w, seed)
        if name == "allsky":
            depth[:] = 0
        cam = pin.PinholeCamera.__new__(pin.PinholeCamera)
        cam.fx, cam.fy, cam.cx, cam.cy, cam.device, cam.dtype = fx, fy, cx, cy, "cpu", torch.float32
        torch.manual_seed(1234 + seed)
        res = bu.generate_coordinate_buffer_from_memory_global_norm(
            torch.from_numpy(depth), cam, torch.from_numpy(poses), percentile=0.05)
        out[f"{name}_depth"], out[f"{name}_poses"] = depth, poses
        out[f"{name}_intr"] = np.array([fx, fy, cx, cy], np.float32)
        out[f"{name}_seed"] = np.array([1234 + seed])
        out[f"{name}_coord"] = res.numpy().astype(np.float32)
        out[f"{name}_coord_u8"] = (res * 255).cpu().numpy().astype(np.uint8)     # the caller's conversion
        # The handful of numbers the reference derives on the HOST before its per-pixel work (K^-1, pose_0^-1 pose_n,
        # the two quantiles of the sample), by the same torch calls in the same order.  LAPACK / BLAS results differ in
        # the last bit between CPU models, so the per-pixel kernels are pinned bit for bit from THESE values
        # (tests/test_buffers.py::test_hip_kernels_bit_exact_from_reference_host_values), whatever host they run beside.
        tp, td = torch.from_numpy(poses), torch.from_numpy(depth)
        k = torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=torch.float32)
        to0 = torch.einsum("ij,bjk->bik", torch.inverse(tp[0]), tp)
        du = sys.modules["infinicube.utils.depth_utils"]
        pts = du.unproject_depth_torch(td, to0, k[None].repeat(n, 1, 1))
        pts[td == 0] = 1e7
        flat = pts.reshape(-1, 3)
        valid = flat[flat[:, 2] < 1e6]
        mins, ranges = np.zeros(3, np.float32), np.ones(3, np.float32)
        if valid.shape[0] > 0:
            torch.manual_seed(1234 + seed)
            smp = valid[torch.randperm(valid.shape[0])[:100000]]
            lo, hi = torch.quantile(smp, 0.05, dim=0), torch.quantile(smp, 0.95, dim=0)
            mins, ranges = lo.numpy(), torch.clamp(hi - lo, min=1e-7).numpy()
            chk = (torch.clamp((pts - lo) / torch.clamp(hi - lo, min=1e-7) * 2.0 - 1.0, -1.0, 1.0) + 1.0) / 2.0
            chk[td == 0] = 1.0
            assert torch.equal(chk, res), "host-value replay does not reproduce the reference output"
        out[f"{name}_kinv"] = torch.inverse(k).numpy()
        out[f"{name}_to_cam0"] = to0.numpy()
        out[f"{name}_mins"], out[f"{name}_ranges"] = mins.astype(np.float32), ranges.astype(np.float32)
        out[f"{name}_has_valid"] = np.array([int(valid.shape[0] > 0)])
        print(name, res.shape, float(res.min()), float(res.max()), "valid", int((depth != 0).sum()))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
