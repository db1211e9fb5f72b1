"""`WanVideoPipeline.from_pretrained` / `WanVideoGenerator(...)` on REAL FILES (tiny ones made here, in the on-disk
formats the reference's patterns name [R infinicube/videogen/inference.py:67-69]): a DiT sharded over two
`diffusion_pytorch_model-0000N-of-00002.safetensors`, `models_t5_umt5-xxl-enc-bf16.pth` (torch pickle, Wan naming)
with a `google/umt5-xxl` tokenizer directory next to it, `Wan2.1_VAE.pth`, plus the fine-tune checkpoint overlay.  No
real checkpoint exists offline, so this pins the loaders' file handling and architecture inference - not the real
files' key names (ORACLE_RISKS.md R12).  The loop runs on the test-only oracle operator set (CPU)."""
import contextlib
import io
import os

import numpy as np
import torch
from safetensors.torch import save_file

from infinicube_amd.videogen import synthetic as syn
from infinicube_amd.videogen.config import TokenGrid, preset
from infinicube_amd.videogen.text_encoder import UMT5Encoder
from infinicube_amd.videogen.vae import WanVAENet
from oracle_ops import OracleOps

CFG = preset("tiny", text_dim=48)
GRID = TokenGrid(5, 64, 96)


def _make_model_dir(root):
    d = root / "models" / "Wan-AI" / "Wan2.1-T2V-1.3B"
    d.mkdir(parents=True)
    sd = syn.make_dit_state_dict(CFG)
    keys = sorted(sd)
    half = len(keys) // 2
    save_file({k: sd[k].contiguous() for k in keys[:half]}, str(d / "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({k: sd[k].contiguous() for k in keys[half:]}, str(d / "diffusion_pytorch_model-00002-of-00002.safetensors"))
    torch.manual_seed(0)
    t5 = UMT5Encoder(vocab_size=64, dim=48, dim_attn=48, dim_ffn=96, num_heads=4, num_layers=2, num_buckets=32)
    torch.save(t5.state_dict(), str(d / "models_t5_umt5-xxl-enc-bf16.pth"))
    vae = WanVAENet(dim=8, z_dim=16)
    torch.save({"model." + k: v for k, v in vae.state_dict().items()}, str(d / "Wan2.1_VAE.pth"))
    # a minimal fast tokenizer in the directory layout the loader expects (<dir of the T5 file>/google/umt5-xxl)
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {"<pad>": 0, "</s>": 1, "<unk>": 2}
    for i, wd in enumerate("the video is about a driving scene captured at daytime weather clear street".split()):
        vocab.setdefault(wd, 3 + i)
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    tok_dir = d / "google" / "umt5-xxl"
    tok_dir.mkdir(parents=True)
    PreTrainedTokenizerFast(tokenizer_object=tk, pad_token="<pad>", eos_token="</s>", unk_token="<unk>").save_pretrained(str(tok_dir))
    return sd


def test_generator_from_files(tmp_path, monkeypatch):
    import infinicube_amd.videogen.inference as inf
    from infinicube.videogen import WanVideoGenerator
    from infinicube_amd.videogen.pipeline import WanVideoPipeline
    sd = _make_model_dir(tmp_path)
    monkeypatch.setenv("ICV_MODEL_ROOT", str(tmp_path / "models"))
    bsd = syn.make_buffer_embedder_state_dict(CFG)
    ck = str(tmp_path / "wan1.3b-t2v-buffer-step-1.safetensors")
    save_file({**{"buffer_embedder." + k: v for k, v in bsd.items()}, "dit.head.modulation": sd["head.modulation"] * 0.5}, ck)
    monkeypatch.setattr(WanVideoPipeline, "_get_ops", lambda self: OracleOps())      # CPU: the test-only operator set
    with contextlib.redirect_stdout(io.StringIO()) as log:
        g = WanVideoGenerator(ck, device="cpu", torch_dtype=torch.float32, use_wan_1pt3b=True)
        assert g.pipe.dit.cfg.dim == CFG.dim and g.pipe.dit.cfg.text_dim == 48 and g.pipe.dit.cfg.num_layers == CFG.num_layers
        assert torch.equal(g.pipe.dit.state_dict()["head.modulation"], sd["head.modulation"] * 0.5)
        assert torch.equal(g.pipe.dit.state_dict()["blocks.1.ffn.0.weight"], sd["blocks.1.ffn.0.weight"])      # from shard 1 or 2
        assert g.pipe.text_encoder.model.dim == 48 and len(g.pipe.text_encoder.model.blocks) == 2
        e = g.pipe.text_encoder.encode("the video is about a driving scene")
        assert e.shape == (512, 48) and float(e[:6].abs().max()) > 0 and float(e[8:].abs().max()) == 0.0
        g.pipe.num_inference_steps = 2
        sem, co = syn.make_dummy_buffers(GRID)
        frames = g.generate(sem, co, seed=0, output_path=str(tmp_path / "video_480p_front.mp4"))
    assert len(frames) == GRID.num_frames and frames[0].size == (GRID.width, GRID.height)
    assert os.path.getsize(tmp_path / "video_480p_front.mp4") > 0
    assert "Loading Wan2.1-T2V-1.3B base model..." in log.getvalue() and "DiT weights loaded, 1 parameters" in log.getvalue()
    a = np.stack([np.asarray(f) for f in frames])
    assert a.std() > 0
