"""The encoders outside the hot loop, PINNED: `text_encoder.UMT5Encoder` against `transformers.UMT5EncoderModel`
and `clip_vision.ClipVisionEncoder` against `transformers.CLIPVisionModel` (penultimate hidden state), on tiny
random-weight configurations with the parameters key-mapped one to one.  Checked against the committed outputs of the
HF modules (tests/golden/aux_encoders.npz, generator tests/golden/make_aux_encoder_golden.py) and, when `transformers`
is importable, against the live HF modules as well.  What this pins: layer order, pre-norm placement, un-scaled T5
attention, per-layer bidirectional relative-position buckets, gated tanh-GELU FFN, padding mask; CLIP class/position
embedding, pre-LayerNorm, fused-QKV attention, erf-GELU MLP, 'hidden state after all but the last block'.
What it cannot pin: the Wan checkpoints' own key names ([EXT], ORACLE_RISKS.md)."""
import os
import sys

import numpy as np
import pytest
import torch

from infinicube_amd.videogen.clip_vision import ClipVisionEncoder
from infinicube_amd.videogen.text_encoder import UMT5Encoder

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "aux_encoders.npz"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))


def _sd(prefix):
    return {k[len(prefix):]: torch.from_numpy(G[k]) for k in G.files if k.startswith(prefix)}


def umt5_from_hf(hf_sd, cfg):
    """HF UMT5EncoderModel state dict -> this repo's UMT5Encoder (Wan checkpoint naming)."""
    m = UMT5Encoder(vocab_size=cfg["vocab_size"], dim=cfg["d_model"], dim_attn=cfg["d_kv"] * cfg["num_heads"],
                    dim_ffn=cfg["d_ff"], num_heads=cfg["num_heads"], num_layers=cfg["num_layers"], num_buckets=32)
    sd = {"token_embedding.weight": hf_sd["shared.weight"], "norm.weight": hf_sd["encoder.final_layer_norm.weight"]}
    for i in range(cfg["num_layers"]):
        a, f = f"encoder.block.{i}.layer.0", f"encoder.block.{i}.layer.1"
        for p in "qkvo":
            sd[f"blocks.{i}.attn.{p}.weight"] = hf_sd[f"{a}.SelfAttention.{p}.weight"]
        sd[f"blocks.{i}.pos_embedding.embedding.weight"] = hf_sd[f"{a}.SelfAttention.relative_attention_bias.weight"]
        sd[f"blocks.{i}.norm1.weight"] = hf_sd[f"{a}.layer_norm.weight"]
        sd[f"blocks.{i}.norm2.weight"] = hf_sd[f"{f}.layer_norm.weight"]
        sd[f"blocks.{i}.ffn.gate.0.weight"] = hf_sd[f"{f}.DenseReluDense.wi_0.weight"]
        sd[f"blocks.{i}.ffn.fc1.weight"] = hf_sd[f"{f}.DenseReluDense.wi_1.weight"]
        sd[f"blocks.{i}.ffn.fc2.weight"] = hf_sd[f"{f}.DenseReluDense.wo.weight"]
    m.load_state_dict(sd, strict=True)
    return m.float().eval()


def clip_from_hf(hf_sd, cfg):
    L = cfg["num_hidden_layers"]
    m = ClipVisionEncoder(image_size=cfg["image_size"], patch=cfg["patch_size"], dim=cfg["hidden_size"],
                          heads=cfg["num_attention_heads"], layers=L, use_blocks=L - 1)
    # transformers 5.x drops the "vision_model." prefix of CLIPVisionModel's state dict; accept both spellings
    hf_sd = {"v." + (k[len("vision_model."):] if k.startswith("vision_model.") else k): t for k, t in hf_sd.items()}
    e, v = "v.embeddings", "v"
    sd = {"patch_embedding.weight": hf_sd[f"{e}.patch_embedding.weight"],
          "cls_embedding": hf_sd[f"{e}.class_embedding"].reshape(1, 1, -1),
          "pos_embedding": hf_sd[f"{e}.position_embedding.weight"][None],
          "pre_norm.weight": hf_sd[f"{v}.pre_layrnorm.weight"], "pre_norm.bias": hf_sd[f"{v}.pre_layrnorm.bias"]}
    for i in range(L):
        h = f"{v}.encoder.layers.{i}"
        for wb in ("weight", "bias"):
            sd[f"transformer.{i}.attn.to_qkv.{wb}"] = torch.cat([hf_sd[f"{h}.self_attn.{p}_proj.{wb}"] for p in "qkv"], 0)
            sd[f"transformer.{i}.attn.proj.{wb}"] = hf_sd[f"{h}.self_attn.out_proj.{wb}"]
            sd[f"transformer.{i}.norm1.{wb}"] = hf_sd[f"{h}.layer_norm1.{wb}"]
            sd[f"transformer.{i}.norm2.{wb}"] = hf_sd[f"{h}.layer_norm2.{wb}"]
            sd[f"transformer.{i}.mlp.0.{wb}"] = hf_sd[f"{h}.mlp.fc1.{wb}"]
            sd[f"transformer.{i}.mlp.2.{wb}"] = hf_sd[f"{h}.mlp.fc2.{wb}"]
    m.load_state_dict(sd, strict=True)
    return m.float().eval()


def test_umt5_matches_transformers_golden():
    import make_aux_encoder_golden as mk
    m = umt5_from_hf(_sd("t5/"), mk.UMT5)
    ids, mask = torch.from_numpy(G["ids"]), torch.from_numpy(G["mask"])
    out = m(ids, mask)[0]
    want = torch.from_numpy(G["t5_out"])
    n = int(mask.sum())
    err = float((out[:n] - want[:n]).abs().max())
    assert err <= 2e-4 * float(want[:n].abs().max()), f"UMT5 encoder differs from transformers' UMT5EncoderModel: {err}"


def test_clip_matches_transformers_golden():
    import make_aux_encoder_golden as mk
    m = clip_from_hf(_sd("clip/"), mk.CLIP)
    pix = torch.from_numpy(G["pix"].astype(np.float32))
    out = m(pix)[0]
    want = torch.from_numpy(G["clip_penultimate"])
    assert out.shape == want.shape == (257, mk.CLIP["hidden_size"])
    err = float((out - want).abs().max())
    assert err <= 2e-4 * float(want.abs().max()), f"CLIP vision tower differs from transformers' CLIPVisionModel: {err}"


def test_live_against_transformers():
    """Same comparison with the HF modules instantiated now (skipped only if transformers cannot be imported)."""
    pytest.importorskip("transformers")
    import make_aux_encoder_golden as mk
    ids, mask, pix = mk.inputs()
    t5 = mk.hf_umt5()
    with torch.no_grad():
        want = t5(input_ids=ids, attention_mask=mask).last_hidden_state[0]
    got = umt5_from_hf(t5.state_dict(), mk.UMT5)(ids, mask)[0]
    n = int(mask.sum())
    assert float((got[:n] - want[:n]).abs().max()) <= 2e-4 * float(want[:n].abs().max())
    assert np.allclose(want.numpy(), G["t5_out"], atol=1e-5), "the committed fixture is stale: re-run make_aux_encoder_golden.py"
    clip = mk.hf_clip()
    mean = torch.tensor((0.48145466, 0.4578275, 0.40821073)).view(1, 3, 1, 1)
    std = torch.tensor((0.26862954, 0.26130258, 0.27577711)).view(1, 3, 1, 1)
    with torch.no_grad():
        wantc = clip(pixel_values=((pix * 0.5 + 0.5) - mean) / std, output_hidden_states=True).hidden_states[-2][0]
    gotc = clip_from_hf(clip.state_dict(), mk.CLIP)(pix)[0]
    assert float((gotc - wantc).abs().max()) <= 2e-4 * float(wantc.abs().max())
