#!/usr/bin/env python3
"""Pins the two encoders OUTSIDE the hot loop against an independent public implementation.

The reference names their checkpoints (`models_t5_umt5-xxl-enc-bf16.pth`, `models_clip_open-clip-xlm-roberta-large-
vit-huge-14.pth` [R infinicube/videogen/inference.py:68,78; download_checkpoint.py:24-29]); the code that consumes them
is in the absent diffsynth fork.  `transformers` (importable in the build container and on the GPU box) ships the same
PUBLIC architectures: `UMT5EncoderModel` (google/umt5-xxl is the model the Wan checkpoint was exported from) and
`CLIPVisionModel` (ViT-H/14).  This script builds tiny random-weight instances of both, runs them, and stores inputs,
the HF state dicts and the HF outputs in tests/golden/aux_encoders.npz; tests/test_aux_pinned.py maps the keys onto
`infinicube_amd.videogen.text_encoder.UMT5Encoder` / `clip_vision.ClipVisionEncoder` and requires the same outputs
(also live against transformers when it is importable).  Run:  python tests/golden/make_aux_encoder_golden.py
"""
import os

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aux_encoders.npz")
UMT5 = dict(vocab_size=97, d_model=48, d_kv=12, d_ff=80, num_layers=3, num_heads=4)
CLIP = dict(hidden_size=64, intermediate_size=256, num_hidden_layers=4, num_attention_heads=4, image_size=224, patch_size=14)


def hf_umt5():
    from transformers import UMT5Config, UMT5EncoderModel
    cfg = UMT5Config(**UMT5, relative_attention_num_buckets=32, relative_attention_max_distance=128,
                     feed_forward_proj="gated-gelu", dropout_rate=0.0, layer_norm_epsilon=1e-6, is_encoder_decoder=False,
                     use_cache=False)
    torch.manual_seed(0)
    m = UMT5EncoderModel(cfg).float().eval()
    for p in m.parameters():
        torch.nn.init.normal_(p, std=0.3 if p.dim() > 1 else 0.1)
        if p.dim() == 1:
            p.data += 1.0
    return m


def hf_clip():
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg = CLIPVisionConfig(**CLIP, hidden_act="gelu", layer_norm_eps=1e-5, attention_dropout=0.0)
    torch.manual_seed(1)
    m = CLIPVisionModel(cfg).float().eval()
    for p in m.parameters():
        torch.nn.init.normal_(p, std=0.05)
    return m


def inputs():
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, UMT5["vocab_size"], (1, 24), generator=g)
    mask = torch.ones(1, 24, dtype=torch.long)
    mask[:, 17:] = 0
    pix = (torch.rand((1, 3, 224, 224), generator=g) * 2 - 1).half().float()   # already 224 x 224 in [-1, 1]; fp16-exact (stored as fp16)
    return ids, mask, pix


def main():
    ids, mask, pix = inputs()
    t5, clip = hf_umt5(), hf_clip()
    with torch.no_grad():
        t5_out = t5(input_ids=ids, attention_mask=mask).last_hidden_state[0]
        mean = torch.tensor((0.48145466, 0.4578275, 0.40821073)).view(1, 3, 1, 1)
        std = torch.tensor((0.26862954, 0.26130258, 0.27577711)).view(1, 3, 1, 1)
        hs = clip(pixel_values=((pix * 0.5 + 0.5) - mean) / std, output_hidden_states=True).hidden_states
    blob = {"ids": ids.numpy(), "mask": mask.numpy(), "pix": pix.numpy().astype(np.float16),
            "t5_out": t5_out.numpy(), "clip_penultimate": hs[-2][0].numpy()}
    for k, v in t5.state_dict().items():
        blob["t5/" + k] = v.numpy()
    for k, v in clip.state_dict().items():
        blob["clip/" + k] = v.numpy()
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
