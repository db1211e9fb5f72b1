"""Drop-in ``WanVideoGenerator`` for InfiniCube stage 2 (boundary "B-outer").

Call surface, defaults, validation messages, progress-line order and the mp4 side effect follow
the reference class one-to-one [R infinicube/videogen/inference.py:30-240] so that
``guidance_buffer_generation.py`` [R infinicube/inference/guidance_buffer_generation.py:739-791]
runs unchanged; behind it sits this repo's MI355X pipeline (pipeline.WanVideoPipeline ->
dit.WanDiT -> libicvideo HIP kernels) instead of diffsynth.  The contract is pinned by
tests/golden/boundary_trace.json, captured from the reference wrapper itself.
"""

from __future__ import annotations

import os
import sys
from typing import Dict, List, Optional

import numpy as np
import torch
from PIL import Image

from .io import load_state_dict, save_video
from .pipeline import ModelConfig, WanVideoPipeline

DEFAULT_PROMPT = "The video is about a driving scene captured at daytime. The weather is clear."
DEFAULT_NEGATIVE_PROMPT = (
    "色调艳丽，过曝，静态，细节模糊不清，字幕，风格，作品，画作，画面，静止，整体发灰，最差质量，低质量，"
    "JPEG压缩残留，丑陋的，残缺的，多余的手指，画得不好的手部，画得不好的脸部，畸形的，毁容的，形态畸形的肢体，"
    "手指融合，静止不动的画面，杂乱的背景，三条腿，背景人很多，倒着走")

# (model id, display name) by size; the three files per model are DiT shards, UMT5 encoder, Wan VAE
_BASE_MODELS = {True: ("Wan-AI/Wan2.1-T2V-1.3B", "Wan2.1-T2V-1.3B"), False: ("Wan-AI/Wan2.1-T2V-14B", "Wan2.1-T2V-14B")}
_BASE_FILES = ("diffusion_pytorch_model*.safetensors", "models_t5_umt5-xxl-enc-bf16.pth", "Wan2.1_VAE.pth")


def _strip_prefix_group(state_dict: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    # str.replace (not removeprefix) on purpose: the reference strips the prefix anywhere in the key
    # [R infinicube/videogen/inference.py:108,122]
    return {k.replace(prefix, ""): v for k, v in state_dict.items() if k.startswith(prefix)}


def _is_rank0() -> bool:
    """Under a multi-rank launch every rank holds the full result; only one of them writes the mp4."""
    try:
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
    except Exception:  # pragma: no cover
        return True


class WanVideoGenerator:
    """Buffer-conditioned Wan2.1 video generation (semantic + coordinate guidance buffers).

    Args mirror the reference constructor [R infinicube/videogen/inference.py:42-50].
    ``pipeline_factory`` is an extension point (not in the reference) used by tests/bench to inject a
    pipeline built from in-memory components instead of checkpoint files.
    """

    def __init__(self, checkpoint_path: str, device: str = "cuda:0", torch_dtype: torch.dtype = torch.bfloat16,
                 buffer_channels: int = 16, enable_vram_management: bool = True, use_wan_1pt3b: bool = False,
                 *, pipeline_factory=None):
        self.checkpoint_path = checkpoint_path
        self.device = device
        self.torch_dtype = torch_dtype
        self.buffer_channels = buffer_channels

        # N-GPU mode behind the unchanged caller: ICV_WORLD=N starts N fresh per-GPU worker processes (rank 0 included) that
        # build this same generator (multigpu.py); this process becomes their client and loads nothing itself
        from . import multigpu
        self._pool = None
        world = multigpu.requested_world()
        if world > 1:
            self._pool = multigpu.pool_for(world, dict(
                checkpoint_path=checkpoint_path, device=device, torch_dtype=torch_dtype, buffer_channels=buffer_channels,
                enable_vram_management=enable_vram_management, use_wan_1pt3b=use_wan_1pt3b))
        if self._pool is not None:
            # settings only (sent with every request); layout / K|V transport = the plan that passed the pool's start-up probe
            self.pipe = self._pool.client_pipeline(device, torch_dtype)
            sys.stdout.write(self._pool.construction_log)     # rank 0's constructor lines = what the code below prints
            return

        model_id, shown = _BASE_MODELS[bool(use_wan_1pt3b)]
        print(f"Loading {shown} base model...")
        configs = [ModelConfig(model_id=model_id, origin_file_pattern=pat, skip_download=True) for pat in _BASE_FILES]
        factory = pipeline_factory or WanVideoPipeline.from_pretrained
        self.pipe = factory(torch_dtype=torch_dtype, device=device, model_configs=configs)

        print(f"Initializing buffer embedder (channels={buffer_channels})...")
        self.pipe.initialize_buffer_embedder(buffer_channels=buffer_channels, zero_init=True)

        print(f"Loading checkpoint: {checkpoint_path}")
        self._load_checkpoint()

        if enable_vram_management:
            print("Enabling VRAM management...")
            self.pipe.enable_vram_management()

        print("✓ WanVideoGenerator initialization complete")

    # A2 ---------------------------------------------------------------------------------------
    def _load_checkpoint(self):
        """Overlay the fine-tuned checkpoint: ``buffer_embedder.*`` strictly, ``dit.*`` partially."""
        state_dict = load_state_dict(self.checkpoint_path)
        if self.pipe.buffer_embedder is not None:
            emb = _strip_prefix_group(state_dict, "buffer_embedder.")
            if emb:
                self.pipe.buffer_embedder.load_state_dict(emb)
                print(f"  ✓ Buffer embedder weights loaded, {len(emb)} parameters")
            else:
                print("  ⚠ Warning: buffer_embedder weights not found in checkpoint")
        dit = _strip_prefix_group(state_dict, "dit.")
        if dit:
            self.pipe.dit.load_state_dict(dit, strict=False)
            print(f"  ✓ DiT weights loaded, {len(dit)} parameters")

    # A3 ---------------------------------------------------------------------------------------
    def _ndarray_to_pil_list(self, buffer_array: np.ndarray) -> List[Image.Image]:
        """(N, H, W, 3) uint8 -> N RGB PIL images; the three checks and their messages are the
        reference's [R infinicube/videogen/inference.py:140-153]."""
        if not isinstance(buffer_array, np.ndarray):
            raise TypeError(f"buffer_array must be numpy.ndarray, got {type(buffer_array)}")
        if buffer_array.ndim != 4 or buffer_array.shape[-1] != 3:
            raise ValueError(f"buffer_array shape must be (N, H, W, 3), got {buffer_array.shape}")
        if buffer_array.dtype != np.uint8:
            raise TypeError(f"buffer_array dtype must be uint8, got {buffer_array.dtype}")
        # inputs may be non-contiguous slices (np.stack(...)[:93]); fromarray copies per frame
        return [Image.fromarray(buffer_array[i], mode="RGB") for i in range(buffer_array.shape[0])]

    # A4 ---------------------------------------------------------------------------------------
    def generate(self, semantic_buffer: np.ndarray, coordinate_buffer: np.ndarray, prompt: str = DEFAULT_PROMPT,
                 negative_prompt=DEFAULT_NEGATIVE_PROMPT, seed: int = 0, tiled: bool = True,
                 output_path: Optional[str] = None, fps: int = 10, quality: int = 8) -> List[Image.Image]:
        """Returns the generated frames as PIL images; writes an mp4 when ``output_path`` is given."""
        if semantic_buffer.shape != coordinate_buffer.shape:
            raise ValueError(
                f"semantic_buffer and coordinate_buffer must have the same shape, "
                f"got {semantic_buffer.shape} and {coordinate_buffer.shape}")
        num_frames, height, width, _channels = semantic_buffer.shape

        print("\nStarting video generation...")
        for label, value in (("Prompt", prompt), ("Frames", num_frames), ("Resolution", f"{height}x{width}"),
                             ("Seed", seed), ("Tiled", tiled)):
            print(f"  - {label}: {value}")

        print("Converting buffer data...")
        semantic_frames = self._ndarray_to_pil_list(semantic_buffer)
        coordinate_frames = self._ndarray_to_pil_list(coordinate_buffer)

        print("Executing video generation...")
        if self._pool is not None:     # the N ranks behind this generator run the request; rank 0's frames come back
            if seed is None:           # unseeded call: ONE drawn seed for every rank (each would otherwise draw its own noise)
                seed = int.from_bytes(os.urandom(7), "little")
            frames = self._pool.generate(semantic_buffer, coordinate_buffer,
                                         dict(prompt=prompt, negative_prompt=negative_prompt, seed=seed, tiled=tiled), self.pipe)
            video = [Image.fromarray(f, mode="RGB") for f in frames]
        else:
            video = self.pipe(prompt=prompt, negative_prompt=negative_prompt, semantic_buffer_video=semantic_frames,
                              coordinate_buffer_video=coordinate_frames, height=height, width=width,
                              num_frames=num_frames, seed=seed, tiled=tiled)

        if output_path is not None and _is_rank0():
            print(f"Saving video to: {output_path}")
            save_video(video, output_path, fps=fps, quality=quality)
            print("✓ Video saved")

        print(f"✓ Video generation complete ({len(video)} frames)")
        return video

    # A5 ---------------------------------------------------------------------------------------
    def __call__(self, *args, **kwargs):
        return self.generate(*args, **kwargs)
