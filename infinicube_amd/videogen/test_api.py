"""Harness for the drop-in WanVideoGenerator — counterpart of the reference's
`infinicube/videogen/test_api.py`: load the two guidance-buffer videos, keep the
first 93 frames, build the generator, generate, report the frame count.

Differences forced by the environment: buffers are read from .npy (uint8 [N,H,W,3]) or, when OpenCV is
installed, from .mp4 like the reference; `--synthetic` runs the same call path without any checkpoint
(random-init weights of the chosen size, stand-in text encoder / VAE) so the script works on a bare GPU box.

    python -m infinicube_amd.videogen.test_api --synthetic --model tiny --frames 17 --height 256 --width 448
    python -m infinicube_amd.videogen.test_api --checkpoint ckpt.safetensors --semantic sem.npy --coordinate coord.npy
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

from . import WanVideoGenerator


def load_buffer(path: str) -> np.ndarray:
    """(N, H, W, 3) uint8 RGB from .npy, or from a video file through OpenCV (BGR -> RGB) like the reference."""
    if path.endswith(".npy"):
        arr = np.load(path)
    else:
        try:
            import cv2
        except ImportError as e:
            raise RuntimeError(f"reading {path!r} needs OpenCV (cv2); convert the buffer to .npy instead") from e
        cap = cv2.VideoCapture(path)
        if not cap.isOpened():
            raise ValueError(f"Failed to open video: {path}")
        frames = []
        while True:
            ok, frame = cap.read()
            if not ok:
                break
            frames.append(cv2.cvtColor(frame, cv2.COLOR_BGR2RGB))
        cap.release()
        if not frames:
            raise ValueError(f"No frames loaded from video: {path}")
        arr = np.stack(frames, axis=0)
    print(f"  Loaded {arr.shape[0]} frames, shape: {arr.shape}")
    return arr


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--semantic", default=None)
    ap.add_argument("--coordinate", default=None)
    ap.add_argument("--use-wan-1pt3b", action="store_true")
    ap.add_argument("--output", default=None)
    ap.add_argument("--prompt", default="The video is about a driving scene captured at daytime. The weather is clear.")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--model", default="tiny", choices=["tiny", "small", "1.3b", "14b"])
    ap.add_argument("--frames", type=int, default=93)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--steps", type=int, default=None, help="override the pipeline's 50 steps (synthetic mode)")
    args = ap.parse_args(argv)

    factory = None
    if args.synthetic:
        import tempfile
        from safetensors.torch import save_file
        from . import synthetic as syn
        from .config import TokenGrid, preset
        from .pipeline import DiTHolder, WanVideoPipeline
        # --synthetic is a self-test mode: the stand-in text encoder / VAE are test infrastructure (tests/standins.py), not product code
        tests_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")
        if not os.path.exists(os.path.join(tests_dir, "standins.py")):
            ap.error("--synthetic needs the repository's tests/ directory (tests/standins.py)")
        import importlib.util       # by file path: tests/ never goes on sys.path (its module names must not shadow anything)
        spec = importlib.util.spec_from_file_location("_icv_selftest_standins", os.path.join(tests_dir, "standins.py"))
        standins = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(standins)
        HashTextEncoder, PoolVAE = standins.HashTextEncoder, standins.PoolVAE
        cfg = preset(args.model)
        grid = TokenGrid(args.frames, args.height, args.width)
        sd = syn.make_dit_state_dict(cfg, device="cuda:0", dtype=torch.bfloat16)
        bsd = syn.make_buffer_embedder_state_dict(cfg)
        ck = tempfile.NamedTemporaryFile(suffix=".safetensors", delete=False).name
        save_file({"buffer_embedder." + k: v for k, v in bsd.items()}, ck)
        args.checkpoint = ck
        semantic, coordinate = syn.make_dummy_buffers(grid)

        def factory(torch_dtype, device, model_configs):
            pipe = WanVideoPipeline(device, torch_dtype, DiTHolder(sd, cfg), HashTextEncoder(cfg), PoolVAE())
            if args.steps:
                pipe.num_inference_steps = args.steps
            return pipe
    else:
        if not (args.checkpoint and args.semantic and args.coordinate):
            ap.error("--checkpoint, --semantic and --coordinate are required (or use --synthetic)")
        print("\n[1/4] Loading buffer videos...")
        semantic, coordinate = load_buffer(args.semantic), load_buffer(args.coordinate)
    assert semantic.shape == coordinate.shape, f"Buffer shapes don't match: {semantic.shape} vs {coordinate.shape}"
    semantic, coordinate = semantic[:93], coordinate[:93]           # the caller's 93-frame cap
    print(f"Final buffer shape: {semantic.shape}")

    print("\n[2/4] Initializing WanVideoGenerator...")
    gen = WanVideoGenerator(checkpoint_path=args.checkpoint, device="cuda:0", torch_dtype=torch.bfloat16, buffer_channels=16,
                            enable_vram_management=True, use_wan_1pt3b=args.use_wan_1pt3b, pipeline_factory=factory)
    print("\n[3/4] Generating video...")
    t0 = time.time()
    frames = gen.generate(semantic_buffer=semantic, coordinate_buffer=coordinate, prompt=args.prompt, seed=0, tiled=True,
                          output_path=args.output, fps=10, quality=8)
    torch.cuda.synchronize()
    print(f"\n[4/4] Test complete!\n✓ Generated {len(frames)} frames in {time.time() - t0:.1f}s")
    return frames


if __name__ == "__main__":
    main()
