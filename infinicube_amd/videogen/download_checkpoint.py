"""Checkpoint download + inventory for the video-generation stage (counterpart of the reference's download script).

The reference instantiates a pipeline once so that diffsynth downloads six files into ``models/<model_id>/``
[R infinicube/videogen/download_checkpoint.py:19-31; R README.md:33].  Run as a script this does the same: every
one of the six (model_id, file pattern) pairs that is missing under the models root is fetched into that layout
(``ModelConfig.download``: ModelScope when installed, as diffsynth defaults to, else the Hugging Face hub), then the
inventory is printed — which entries are present and, with ``--inspect``, which architecture the DiT shards'
tensor shapes imply.  ``--no-download`` prints the inventory only.  (The generator itself always passes
``skip_download=True`` [R infinicube/videogen/inference.py:67-69] and never touches the network.)

    python -m infinicube_amd.videogen.download_checkpoint [--models-root models] [--no-download] [--inspect]
"""

from __future__ import annotations

import argparse
import glob
import os
import sys
from typing import List, Tuple

from .pipeline import ModelConfig

# (model_id, origin_file_pattern, what it is) — the list at [R infinicube/videogen/download_checkpoint.py:24-29]
REQUIRED: List[Tuple[str, str, str]] = [
    ("Wan-AI/Wan2.1-T2V-1.3B", "diffusion_pytorch_model*.safetensors", "Wan2.1 t2v 1.3B DiT"),
    ("Wan-AI/Wan2.1-T2V-14B", "diffusion_pytorch_model*.safetensors", "Wan2.1 t2v 14B DiT (sharded)"),
    ("Wan-AI/Wan2.1-T2V-14B", "models_t5_umt5-xxl-enc-bf16.pth", "UMT5-XXL text encoder"),
    ("Wan-AI/Wan2.1-T2V-14B", "Wan2.1_VAE.pth", "Wan 3-D VAE"),
    ("Wan-AI/Wan2.1-I2V-14B-480P", "diffusion_pytorch_model*.safetensors", "Wan2.1 i2v 14B DiT (image branch)"),
    ("Wan-AI/Wan2.1-I2V-14B-480P", "models_clip_open-clip-xlm-roberta-large-vit-huge-14.pth", "CLIP ViT-H/14 image encoder (i2v)"),
]


def inventory(models_root: str = "models"):
    """[(ModelConfig, description, [matching files])] for the six files the stage needs."""
    out = []
    for model_id, pattern, what in REQUIRED:
        mc = ModelConfig(model_id=model_id, origin_file_pattern=pattern, skip_download=True)
        files = sorted(glob.glob(os.path.join(models_root, model_id, pattern)))
        out.append((mc, what, files))
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--models-root", default=os.environ.get("ICV_MODEL_ROOT", "models"))
    ap.add_argument("--inspect", action="store_true", help="read DiT shard headers and print the implied architecture")
    ap.add_argument("--no-download", action="store_true", help="inventory only: do not fetch missing files")
    args = ap.parse_args(argv)
    if not args.no_download:
        for model_id, pattern, what in REQUIRED:
            mc = ModelConfig(model_id=model_id, origin_file_pattern=pattern, local_model_path=args.models_root)
            if not mc.present():
                try:
                    mc.download()
                except Exception as e:   # no network / hub package: report and go on to the inventory
                    print(f"  could not download {model_id}/{pattern} ({what}): {type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}")
    missing = 0
    for mc, what, files in inventory(args.models_root):
        mark = "ok     " if files else "MISSING"
        missing += not files
        size = sum(os.path.getsize(f) for f in files) / 2 ** 30
        print(f"[{mark}] {mc.model_id}/{mc.origin_file_pattern}  ({what}; {len(files)} file(s), {size:.1f} GiB)")
        if files and args.inspect and files[0].endswith(".safetensors"):
            from .config import infer_config_from_state_dict
            from .io import load_sharded_state_dict
            cfg = infer_config_from_state_dict(load_sharded_state_dict(os.path.join(args.models_root, mc.model_id, mc.origin_file_pattern)))
            print(f"          -> {cfg.name}: dim {cfg.dim}, ffn {cfg.ffn_dim}, {cfg.num_layers} layers, in_dim {cfg.in_dim}"
                  + (f", image branch ({cfg.img_dim})" if cfg.has_image_input else ""))
    if missing:
        print(f"{missing} of {len(REQUIRED)} entries missing under {os.path.abspath(args.models_root)!r}"
              + (" (--no-download: nothing was fetched)." if args.no_download else "."))
    return 1 if missing else 0


if __name__ == "__main__":
    sys.exit(main())
