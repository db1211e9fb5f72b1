"""Stock-PyTorch components outside the hot loop (Wan-VAE, UMT5 encoder): structural properties that
hold for the published architectures regardless of weights.  [EXT]: no checkpoints exist offline, so
these tests pin shapes, parameter naming, causality and masking — not learned behaviour."""
import numpy as np
import pytest
import torch

from infinicube_amd.videogen.text_encoder import T5RelativeEmbedding, UMT5Encoder, UMT5TextEncoder
from infinicube_amd.videogen.vae import WanVAE, WanVAENet, _tile_tasks


def _tiny_vae():
    torch.manual_seed(0)
    net = WanVAENet(dim=8, z_dim=4, dim_mult=(1, 2, 4, 4), num_res_blocks=1)
    for p in net.parameters():
        torch.nn.init.normal_(p, std=0.2) if p.dim() > 1 else None
    return net.float().eval()


def test_vae_shapes_and_param_names():
    net = WanVAENet()
    sd = net.state_dict()
    want = {
        "encoder.conv1.weight": (96, 3, 3, 3, 3),
        "encoder.downsamples.0.residual.0.gamma": (96, 1, 1, 1),
        "encoder.downsamples.2.resample.1.weight": (96, 96, 3, 3),
        "encoder.downsamples.3.shortcut.weight": (192, 96, 1, 1, 1),
        "encoder.downsamples.5.time_conv.weight": (192, 192, 3, 1, 1),
        "encoder.downsamples.8.time_conv.weight": (384, 384, 3, 1, 1),
        "encoder.middle.1.to_qkv.weight": (1152, 384, 1, 1),
        "encoder.head.2.weight": (32, 384, 3, 3, 3),
        "conv1.weight": (32, 32, 1, 1, 1),
        "conv2.weight": (16, 16, 1, 1, 1),
        "decoder.conv1.weight": (384, 16, 3, 3, 3),
        "decoder.upsamples.3.time_conv.weight": (768, 384, 3, 1, 1),
        "decoder.upsamples.3.resample.1.weight": (192, 384, 3, 3),
        "decoder.upsamples.4.shortcut.weight": (384, 192, 1, 1, 1),
        "decoder.upsamples.11.resample.1.weight": (96, 192, 3, 3),
        "decoder.upsamples.14.residual.6.weight": (96, 96, 3, 3, 3),
        "decoder.head.2.weight": (3, 96, 3, 3, 3),
    }
    for k, shp in want.items():
        assert k in sd and tuple(sd[k].shape) == shp, (k, tuple(sd[k].shape) if k in sd else None)
    assert "decoder.upsamples.11.time_conv.weight" not in sd and "encoder.downsamples.2.time_conv.weight" not in sd
    n_params = sum(v.numel() for v in sd.values())
    assert 120e6 < n_params < 135e6          # ~127 M parameters


def test_vae_geometry_and_causality():
    net = _tiny_vae()
    x = torch.randn(1, 3, 9, 32, 48)
    z = net.encode(x)
    assert z.shape == (1, 4, 3, 4, 6)
    y = net.decode(z)
    assert y.shape == (1, 3, 9, 32, 48) and float(y.abs().max()) <= 1.0
    # causal: latent frame j depends only on input frames <= 4j (this is what makes whole-clip
    # processing identical to upstream's chunk-by-chunk streaming with a feature cache)
    z5 = net.encode(x[:, :, :5])
    assert torch.allclose(z5, z[:, :, :2], atol=1e-5)
    z1 = net.encode(x[:, :, :1])
    assert torch.allclose(z1, z[:, :, :1], atol=1e-5)
    # decoder: output frames <= 4j depend only on latent frames <= j
    y2 = net.decode(z[:, :, :2])
    assert torch.allclose(y2, y[:, :, :5], atol=1e-5)


def test_vae_tiling():
    net = _tiny_vae()
    vae = WanVAE(net, "cpu", torch.float32)
    video = torch.rand(3, 5, 64, 96) * 2 - 1
    z_full = vae.encode(video, tiled=False)
    z_one = vae.encode(video, tiled=True, tile_size=(8, 12), tile_stride=(4, 6))      # one tile covers everything
    assert torch.allclose(z_full, z_one, atol=1e-5)
    z_t = vae.encode(video, tiled=True, tile_size=(4, 6), tile_stride=(2, 3))
    assert z_t.shape == z_full.shape == (4, 2, 8, 12) and torch.isfinite(z_t).all()
    y = vae.decode(z_full, tiled=True, tile_size=(4, 6), tile_stride=(2, 3))
    assert y.shape == (3, 5, 64, 96) and torch.isfinite(y).all()
    assert torch.allclose(vae.decode(z_full, tiled=False), vae.decode(z_full, tiled=True, tile_size=(8, 12), tile_stride=(4, 6)), atol=1e-5)
    # upstream's tile enumeration for 480p latents (60 x 104) with (30,52)/(15,26): 3 x 3 tiles
    assert len(_tile_tasks(60, 104, (30, 52), (15, 26))) == 9


def test_t5_relative_buckets_and_masking():
    emb = T5RelativeEmbedding(32, 4)
    b = emb(6, 6)
    assert b.shape == (1, 4, 6, 6)
    # bucket index: 0 on the diagonal, keys to the right use the upper half of the buckets
    torch.manual_seed(0)
    enc = UMT5Encoder(vocab_size=50, dim=32, dim_attn=32, dim_ffn=64, num_heads=4, num_layers=2, num_buckets=32).eval()
    ids = torch.randint(0, 50, (1, 10))
    mask = torch.ones(1, 10, dtype=torch.long)
    mask[:, 6:] = 0
    out = enc(ids, mask)
    ids2 = ids.clone()
    ids2[:, 6:] = 7                      # change only masked (padding) tokens
    out2 = enc(ids2, mask)
    assert torch.allclose(out[:, :6], out2[:, :6], atol=1e-5), "padding tokens must not influence real tokens"
    names = set(enc.state_dict())
    for k in ("token_embedding.weight", "blocks.0.norm1.weight", "blocks.0.attn.q.weight", "blocks.1.ffn.gate.0.weight",
              "blocks.1.ffn.fc1.weight", "blocks.1.ffn.fc2.weight", "blocks.0.pos_embedding.embedding.weight", "norm.weight"):
        assert k in names, k
    assert not any(k.endswith(".bias") for k in names)


def test_text_encoder_zero_pads_context():
    class Tok:
        def __call__(self, texts, **kw):
            n = kw["max_length"]
            ids = torch.zeros(1, n, dtype=torch.long)
            m = torch.zeros(1, n, dtype=torch.long)
            k = min(n, len(texts[0].split()) + 1)
            ids[0, :k] = torch.arange(1, k + 1)
            m[0, :k] = 1
            return {"input_ids": ids, "attention_mask": m}
    enc = UMT5Encoder(vocab_size=50, dim=32, dim_attn=32, dim_ffn=64, num_heads=4, num_layers=1).eval()
    te = UMT5TextEncoder(enc, Tok(), "cpu", text_len=16)
    e = te.encode("a  driving   scene &amp; more")
    assert e.shape == (16, 32) and float(e[6:].abs().max()) == 0.0 and float(e[:6].abs().max()) > 0.0


def test_clip_vision_tower_geometry(tmp_path):
    """CLIP ViT vision tower (i2v conditioning, stock torch): 257 tokens for 224/14, penultimate-block output,
    strict loading of the visual.* half of a checkpoint."""
    from PIL import Image
    from safetensors.torch import save_file
    from infinicube_amd.videogen.clip_vision import ClipVisionEncoder, load_clip_vision
    arch = dict(image_size=224, patch=14, dim=64, heads=2, layers=3, use_blocks=2)
    torch.manual_seed(0)
    enc = ClipVisionEncoder(**arch)
    for p in enc.parameters():
        torch.nn.init.normal_(p, std=0.05)
    sd = {"visual." + k: v.contiguous() for k, v in enc.state_dict().items()}
    sd["textual.junk"] = torch.zeros(3)
    sd["visual.post_norm.weight"] = torch.ones(64)       # present upstream, unused by Wan
    path = str(tmp_path / "clip.safetensors")
    save_file(sd, path)
    loaded = load_clip_vision(path, "cpu", torch.float32, **arch)
    img = Image.fromarray(np.random.default_rng(1).integers(0, 255, (96, 160, 3), dtype=np.uint8), mode="RGB")
    tok = loaded.encode_image(img)
    assert tok.shape == (257, 64) and torch.isfinite(tok).all()
    assert torch.allclose(tok, enc.eval().encode_image(img), atol=1e-6)
    # the last block is NOT applied (use_blocks = layers - 1)
    full = ClipVisionEncoder(**{**arch, "use_blocks": 3})
    full.load_state_dict(enc.state_dict())
    assert not torch.allclose(full.encode_image(img), tok)
    del sd["visual.transformer.1.mlp.0.weight"]
    save_file(sd, path)
    with pytest.raises(KeyError, match="missing"):
        load_clip_vision(path, "cpu", torch.float32, **arch)
