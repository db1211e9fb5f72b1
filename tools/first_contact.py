"""First contact with REAL weights: the one script that settles ORACLE_RISKS.md R1-R7 (and the e4m3 projection set).

No checkpoint exists in the build container or on the GPU boxes (the path's arithmetic lives in the absent, un-pinned
`diffsynth` fork [R pyproject.toml:71]; the reference holds no golden vector [R infinicube/videogen/test_api.py:88-90]), so the
parity of the denoising loop is UNPINNED until somebody with the files runs this.  Given the fine-tuned checkpoint the reference's
generator takes [R infinicube/videogen/inference.py:42-60] and the `models/` folder of its download script
[R infinicube/videogen/download_checkpoint.py:19-31]:

    python tools/first_contact.py --checkpoint checkpoints/wan14b-t2v-buffer-step-1200.safetensors [--models-root models]
           [--1pt3b] [--frames 93 --height 480 --width 832] [--steps 4] [--fp8-study] [--reference-latent ref.pt]

 1. R4  `describe_checkpoint`: names / shapes under `buffer_embedder.*` -> embedder hypothesis H1 (concat conv) / H2 (dual conv) /
        unknown (then nothing below runs: write the variant first); `dit.*` overlay keys vs the base model's key set.
 2. R8/R9 structural checks on the base DiT's key list (full-d RMSNorm weights, affine norm3, [1,6,d] modulation).
 3. R12 strict loads of UMT5 / Wan-VAE (and CLIP for i2v) through `WanVideoPipeline.from_pretrained` - a name mismatch raises.
 4. R1-R3 one short CFG loop on dummy guidance buffers with and without `ICV_REFERENCE_ROUNDING` (bf16 timestep / noise / CFG /
        Euler like an all-bf16 upstream pipeline): latent PSNR between the two arms = how much the rounding points matter with THESE
        weights; with `--reference-latent` (a latent saved from the fork on the same seed / buffers / prompt / steps) each arm is
        compared with it and the closer one is named.
 5. (--fp8-study) the e4m3 projection subsets against this build's bf16 arm with the REAL weights: is the narrow default
        (`WanDiT.FP8_DEFAULT`, chosen on random-init weights at 14B depth) still needed, or do O / FFN tolerate e4m3 too?
Prints one JSON record; exits non-zero if a structural assumption fails.  Needs a GPU (the loop has no CPU path)."""
from __future__ import annotations

import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def psnr(a, b):
    a, b = a.double(), b.double()
    mse = float(((a - b) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * math.log10(float(b.abs().max()) ** 2 / mse)


def embedder_hypothesis(desc: dict):
    """(name, detail) from the `buffer_embedder.*` header entries of the fine-tuned checkpoint (ORACLE_RISKS R4)."""
    be = {k[len("buffer_embedder."):]: tuple(v) for k, v in desc["buffer_embedder"].items()}
    if set(be) == {"proj.weight", "proj.bias"} and len(be["proj.weight"]) == 5 and be["proj.weight"][2:] == (1, 2, 2):
        return "H1", f"one Conv3d({be['proj.weight'][1]} -> {be['proj.weight'][0]}, kernel = stride = (1,2,2)) over the channel-concatenated buffer latents"
    if set(be) == {"semantic_proj.weight", "semantic_proj.bias", "coordinate_proj.weight", "coordinate_proj.bias"}:
        return "H2", f"one Conv3d({be['semantic_proj.weight'][1]} -> {be['semantic_proj.weight'][0]}) per buffer, outputs summed"
    return "unknown", f"unexpected keys / shapes: {be} - neither H1 nor H2 (SURVEY §8a K1): implement this variant in pipeline.BufferEmbedder + dit.load_buffer_embedder first"


def structural_checks(sd_keys_shapes: dict):
    """ORACLE_RISKS R8 / R9 on the base DiT's tensors; returns (ok, findings)."""
    f, ok = {}, True
    d = sd_keys_shapes["blocks.0.self_attn.q.weight"][0]
    nq = sd_keys_shapes.get("blocks.0.self_attn.norm_q.weight")
    f["R8_rmsnorm_over_full_d"] = nq == (d,)
    f["R9_norm3_affine"] = "blocks.0.norm3.weight" in sd_keys_shapes and "blocks.0.norm3.bias" in sd_keys_shapes
    f["R9_norm1_not_affine"] = "blocks.0.norm1.weight" not in sd_keys_shapes
    f["R9_modulation_shape"] = sd_keys_shapes.get("blocks.0.modulation") == (1, 6, d) and sd_keys_shapes.get("head.modulation") == (1, 2, d)
    f["patch_embedding"] = sd_keys_shapes.get("patch_embedding.weight")
    for k, v in f.items():
        if k.startswith("R") and v is not True:
            ok = False
    return ok, f


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--checkpoint", required=True, help="the fine-tuned buffer checkpoint (buffer_embedder.* + dit.* keys)")
    ap.add_argument("--models-root", default="models")
    ap.add_argument("--1pt3b", dest="small", action="store_true")
    ap.add_argument("--frames", type=int, default=93)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--fp8-study", action="store_true")
    ap.add_argument("--reference-latent", default=None, help="torch.save'd latent [16, T, H/8, W/8] from the fork: same seed, buffers, prompt, steps")
    ap.add_argument("--describe-only", action="store_true", help="steps 1-2 only (reads file headers; no GPU needed)")
    args = ap.parse_args(argv)

    from infinicube_amd.videogen.io import describe_checkpoint
    out = {"checkpoint": args.checkpoint}
    desc = describe_checkpoint(args.checkpoint)
    hyp, detail = embedder_hypothesis(desc)
    out["R4_buffer_embedder"] = {"hypothesis": hyp, "detail": detail, "keys": {k: list(v) for k, v in desc["buffer_embedder"].items()}}
    out["dit_overlay_keys"] = len(desc["dit"])
    out["unprefixed_keys_dropped_by_the_loader"] = sorted(desc["other"])[:20]
    if hyp == "unknown":
        print(json.dumps(out, indent=1))
        return 2
    if args.describe_only:
        print(json.dumps(out, indent=1))
        return 0

    import numpy as np
    import torch
    os.environ["ICV_MODEL_ROOT"] = args.models_root
    from infinicube_amd.videogen import synthetic as syn
    from infinicube_amd.videogen.config import TokenGrid
    from infinicube_amd.videogen.inference import WanVideoGenerator

    grid = TokenGrid(args.frames, args.height, args.width)
    gen = WanVideoGenerator(args.checkpoint, device="cuda:0", use_wan_1pt3b=args.small)       # step 3: strict loads happen here
    base = {k: tuple(v.shape) for k, v in gen.pipe.dit.state_dict().items()}
    ok, findings = structural_checks(base)
    out["structure"] = findings
    out["dit_overlay_unknown_keys"] = sorted(k[len("dit."):] for k in desc["dit"] if k[len("dit."):] not in base)[:20]
    gen.pipe.num_inference_steps = args.steps
    sem, co = syn.make_dummy_buffers(grid)
    prompt = "The video is about a driving scene captured at daytime. The weather is clear."
    kw = dict(prompt=prompt, negative_prompt="", semantic_buffer_video=gen._ndarray_to_pil_list(sem), coordinate_buffer_video=gen._ndarray_to_pil_list(co),
              height=grid.height, width=grid.width, num_frames=grid.num_frames, seed=args.seed, tiled=True, return_latents=True)
    lat = {}
    for rr in (False, True):
        gen.pipe.reference_rounding = rr
        lat[rr] = gen.pipe(**kw).float().cpu()
        assert torch.isfinite(lat[rr]).all()
    out["R1_R3_rounding"] = {"steps": args.steps, "latent_psnr_exact_vs_reference_rounding_db": psnr(lat[True], lat[False])}
    if args.reference_latent:
        ref = torch.load(args.reference_latent, map_location="cpu").float().reshape(lat[False].shape)
        p = {("reference_rounding" if rr else "exact"): psnr(lat[rr], ref) for rr in (False, True)}
        out["R1_R3_rounding"]["latent_psnr_vs_fork_db"] = p
        out["R1_R3_rounding"]["closer_arm"] = max(p, key=p.get)
        out["R1_R3_rounding"]["meets_40_db"] = max(p.values()) >= 40.0
    if args.fp8_study:
        from infinicube_amd.videogen.dit import WanDiT
        gen.pipe.reference_rounding = False
        study = {}
        for name, sel in (("default " + ",".join(WanDiT.FP8_DEFAULT), WanDiT.FP8_DEFAULT), ("wqkv,xq_w,xo_w", ("wqkv", "xq_w", "xo_w")),
                          ("wqkv,wo", ("wqkv", "wo")), ("wqkv,f0_w", ("wqkv", "f0_w")), ("wqkv,f2_w", ("wqkv", "f2_w")), ("all six", WanDiT.FP8_WEIGHTS)):
            os.environ["ICV_FP8_WEIGHTS"] = ",".join(sel)
            gen.pipe.gemm_dtype = gen.pipe.attn_dtype = "fp8"
            gen.pipe._engine = None                                   # new engine: the weights are quantised at pack time
            study[name] = psnr(gen.pipe(**kw).float().cpu(), lat[False])
        os.environ.pop("ICV_FP8_WEIGHTS", None)
        out["fp8_projection_subsets_latent_psnr_vs_bf16_db"] = study
        out["fp8_largest_subset_at_40_db"] = max((k for k, v in study.items() if v >= 40.0), key=lambda k: k.count(",") + 6 * (k == "all six"), default=None)
    print(json.dumps(out, indent=1))
    return 0 if ok else 3


if __name__ == "__main__":
    sys.exit(main())
