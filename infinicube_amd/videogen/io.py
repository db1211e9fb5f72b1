"""Checkpoint / video I/O at the boundary.

``load_state_dict(path) -> Dict[str, Tensor]`` and ``save_video(frames, path, fps=, quality=)`` are
the two free functions the reference imports from diffsynth
[R infinicube/videogen/inference.py:25] and calls at [R infinicube/videogen/inference.py:103,231].
"""

from __future__ import annotations

import glob
import json
import os
import struct
from typing import Dict, List

import torch


def load_state_dict(path: str, device: str = "cpu") -> Dict[str, torch.Tensor]:
    """Flat ``{name: tensor}``; safetensors by extension, torch pickle otherwise."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device=device)
    sd = torch.load(path, map_location=device, weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return sd


def load_sharded_state_dict(pattern: str, device: str = "cpu") -> Dict[str, torch.Tensor]:
    files = sorted(glob.glob(pattern))
    if not files:
        raise FileNotFoundError(f"no checkpoint file matches {pattern!r} (skip_download=True: nothing is fetched)")
    out: Dict[str, torch.Tensor] = {}
    for f in files:
        out.update(load_state_dict(f, device))
    return out


def describe_checkpoint(path: str) -> Dict[str, Dict[str, tuple]]:
    """Read ONLY the safetensors header and report ``buffer_embedder.*`` / ``dit.*`` names+shapes:
    selects the embedder variant and the model size without touching tensor data."""
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(n))
    out = {"buffer_embedder": {}, "dit": {}, "other": {}}
    for k, meta in header.items():
        if k == "__metadata__":
            continue
        grp = "buffer_embedder" if k.startswith("buffer_embedder.") else "dit" if k.startswith("dit.") else "other"
        out[grp][k] = tuple(meta["shape"])
    return out


def save_video(frames: List, save_path: str, fps: int = 10, quality: int = 8) -> None:
    """mp4 at ``save_path`` (stage 3 reads it back [R infinicube/inference/scene_gaussian_generation.py:290-293]).
    With imageio installed — it is part of the reference's environment — this is libx264 through imageio-ffmpeg,
    like diffsynth's save_video.  Without it the file is written by the built-in H.264 writer (``h264pcm``: the same
    codec and container, Constrained-Baseline IDR pictures of I_PCM macroblocks — lossless in 4:2:0, so ``quality`` has
    nothing to act on and the file is large) and a line says so; ``ICV_MP4_CODEC=mjpeg`` selects the smaller
    Motion-JPEG file of ``mp4mux`` instead."""
    os.makedirs(os.path.dirname(os.path.abspath(save_path)), exist_ok=True)
    try:
        import imageio
    except ImportError:
        write_video_without_ffmpeg(frames, save_path, fps=fps, quality=quality)
        return
    import numpy as np
    writer = imageio.get_writer(save_path, fps=fps, quality=quality)
    try:
        for fr in frames:
            writer.append_data(np.asarray(fr))
    finally:
        writer.close()


def write_video_without_ffmpeg(frames: List, save_path: str, fps: float = 10, quality: int = 8) -> str:
    """The in-tree mp4 writers: H.264 (I_PCM, default) or Motion-JPEG (ICV_MP4_CODEC=mjpeg).  Returns the codec used."""
    codec = os.environ.get("ICV_MP4_CODEC", "h264").lower()
    if codec not in ("h264", "mjpeg"):
        raise ValueError(f"ICV_MP4_CODEC must be 'h264' or 'mjpeg', got {codec!r}")
    if codec == "mjpeg":
        from .mp4mux import write_mjpeg_mp4
        print(f"  (imageio not installed: writing {save_path} as Motion-JPEG mp4 instead of libx264)")
        write_mjpeg_mp4(frames, save_path, fps=fps, quality=quality)
    else:
        from .h264pcm import write_h264_mp4
        print(f"  (imageio not installed: writing {save_path} with the built-in H.264 writer (intra PCM, lossless 4:2:0) instead of libx264)")
        write_h264_mp4(frames, save_path, fps=fps)
    return codec
