"""Whole generate() path on CPU: the product's host code (generator -> pipeline -> DiT driver ->
sequence plan) with the TEST-ONLY oracle operator set injected and deterministic stand-ins for the
UMT5 / VAE components.  Checks the drop-in's observable behaviour: frame count/size, seed
determinism, that prompt / seed / buffers each change the result, checkpoint overlay semantics."""
import contextlib
import io
import json
import os
import re

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

import infinicube_amd.videogen.inference as inf
from infinicube.videogen import WanVideoGenerator
from infinicube_amd.videogen import synthetic as syn
from infinicube_amd.videogen.config import TokenGrid, preset
from infinicube_amd.videogen.pipeline import BufferEmbedder, DiTHolder, ModelConfig, WanVideoPipeline
from standins import HashTextEncoder, PoolVAE
from oracle_ops import OracleOps

CFG, GRID = preset("tiny"), TokenGrid(5, 64, 96)


@pytest.fixture(scope="module")
def generator(tmp_path_factory):
    sd = syn.make_dit_state_dict(CFG)
    bsd = syn.make_buffer_embedder_state_dict(CFG)
    ck = {**{"buffer_embedder." + k: v for k, v in bsd.items()},
          "dit.blocks.0.modulation": sd["blocks.0.modulation"] * 1.5, "optimizer.junk": torch.zeros(1)}
    path = str(tmp_path_factory.mktemp("ck") / "step-1.safetensors")
    save_file(ck, path)

    def factory(torch_dtype, device, model_configs):
        assert [m.origin_file_pattern for m in model_configs] == [
            "diffusion_pytorch_model*.safetensors", "models_t5_umt5-xxl-enc-bf16.pth", "Wan2.1_VAE.pth"]
        return WanVideoPipeline(device, torch_dtype, DiTHolder(sd, CFG), HashTextEncoder(CFG), PoolVAE(), ops=OracleOps())

    with contextlib.redirect_stdout(io.StringIO()):
        g = WanVideoGenerator(path, device="cpu", use_wan_1pt3b=True, pipeline_factory=factory)
    g._sd, g._bsd = sd, bsd
    return g


def _gen(g, sem, co, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        lat = g.pipe(prompt=kw.get("prompt", "a street"), negative_prompt="bad", semantic_buffer_video=g._ndarray_to_pil_list(sem),
                     coordinate_buffer_video=g._ndarray_to_pil_list(co), height=GRID.height, width=GRID.width,
                     num_frames=GRID.num_frames, seed=kw.get("seed", 0), tiled=True, num_inference_steps=2, return_latents=True)
    return lat


def test_checkpoint_overlay(generator):
    g = generator
    assert torch.equal(g.pipe.buffer_embedder.state_dict()["proj.weight"], g._bsd["proj.weight"])
    assert torch.equal(g.pipe.dit.state_dict()["blocks.0.modulation"], g._sd["blocks.0.modulation"] * 1.5)
    assert torch.equal(g.pipe.dit.state_dict()["blocks.1.modulation"], g._sd["blocks.1.modulation"])
    assert g.pipe.vram_management_enabled


def test_generate_frames_and_determinism(generator, monkeypatch, tmp_path):
    g = generator
    sem, co = syn.make_dummy_buffers(GRID)
    saved = {}
    monkeypatch.setattr(inf, "save_video", lambda fr, p, fps, quality: saved.update(n=len(fr), p=p, fps=fps, q=quality))
    assert g.pipe.num_inference_steps == 50 and g.pipe.cfg_scale == 5.0 and g.pipe.sigma_shift == 5.0   # fork defaults
    with contextlib.redirect_stdout(io.StringIO()), monkeypatch.context() as mp:
        mp.setattr(g.pipe, "num_inference_steps", 2)    # generate() exposes no step count: shorten for the test only
        frames = g.generate(sem, co, seed=0, output_path=str(tmp_path / "video_480p_front.mp4"))
    assert len(frames) == GRID.num_frames and frames[0].size == (GRID.width, GRID.height) and frames[0].mode == "RGB"
    assert saved == {"n": GRID.num_frames, "p": str(tmp_path / "video_480p_front.mp4"), "fps": 10, "q": 8}
    a, b = _gen(g, sem, co, seed=0), _gen(g, sem, co, seed=0)
    assert torch.equal(a, b), "same seed + buffers + prompt must give identical latents"
    assert not torch.equal(a, _gen(g, sem, co, seed=1))
    assert not torch.equal(a, _gen(g, sem, co, prompt="a different prompt"))
    sem2 = sem.copy()
    sem2[:, :32] = 10
    assert not torch.equal(a, _gen(g, sem2, co)), "guidance buffers must condition the result"


def test_buffer_embedder_strict_and_variants():
    be = BufferEmbedder(CFG.dim, 16, zero_init=True)
    assert all(float(v.abs().max()) == 0 for v in be.parameters())      # zero_init
    be.load_state_dict(syn.make_buffer_embedder_state_dict(CFG, variant="dual"))
    assert be.variant == "dual"
    with pytest.raises(RuntimeError, match="expected keys"):
        BufferEmbedder(CFG.dim).load_state_dict({"conv.weight": torch.zeros(1)})
    with pytest.raises(RuntimeError, match="size mismatch"):
        BufferEmbedder(CFG.dim).load_state_dict({"proj.weight": torch.zeros(3), "proj.bias": torch.zeros(CFG.dim)})


def test_from_pretrained_fails_loudly_without_files(tmp_path, monkeypatch):
    monkeypatch.setenv("ICV_MODEL_ROOT", str(tmp_path))
    with pytest.raises(FileNotFoundError):
        WanVideoPipeline.from_pretrained(device="cpu", model_configs=[
            ModelConfig(model_id="Wan-AI/Wan2.1-T2V-1.3B", origin_file_pattern="diffusion_pytorch_model*.safetensors", skip_download=True)])


def test_product_has_no_cpu_fallback():
    """HipOps must refuse a CPU device / missing GPU instead of silently computing elsewhere."""
    from infinicube_amd import native
    from infinicube_amd.videogen.ops import HipOps
    with pytest.raises(native.NativeError):
        HipOps("cpu")
    if not torch.cuda.is_available():
        with pytest.raises(native.NativeError):
            HipOps("cuda:0")


def test_package_never_imports_oracle():
    import pathlib
    root = pathlib.Path(__file__).resolve().parents[1]
    for f in list((root / "infinicube_amd").rglob("*.py")) + list((root / "infinicube").rglob("*.py")) + list((root / "diffsynth").rglob("*.py")):
        src = f.read_text(encoding="utf-8")
        assert not re.search(r"^\s*(from|import)\s+(oracle|oracle_ops|tests)\b", src, flags=re.M), f


def test_image_to_video_pipeline_branch():
    """BASELINE.json config #5 plumbing: an i2v DiT (in_dim 36 + CLIP branch) through the pipeline's extra
    ``input_image=`` keyword — the image must change the result, and misuse must fail loudly."""
    from PIL import Image
    from standins import HashImageEncoder
    cfg = preset("tiny-i2v")
    sd = syn.make_dit_state_dict(cfg)
    pipe = WanVideoPipeline("cpu", torch.bfloat16, DiTHolder(sd), HashTextEncoder(cfg), PoolVAE(), ops=OracleOps(),
                            image_encoder=HashImageEncoder(cfg))
    assert pipe.dit.cfg.has_image_input and pipe.dit.cfg.in_dim == 36 and pipe.dit.cfg.img_dim == cfg.img_dim
    pipe.initialize_buffer_embedder(16, zero_init=True)
    rng = np.random.default_rng(0)
    img_a = Image.fromarray(rng.integers(0, 255, (GRID.height, GRID.width, 3), dtype=np.uint8), mode="RGB")
    img_b = Image.fromarray(rng.integers(0, 255, (48, 80, 3), dtype=np.uint8), mode="RGB")   # resized inside
    kw = dict(prompt="a street", negative_prompt="bad", height=GRID.height, width=GRID.width,
              num_frames=GRID.num_frames, seed=0, num_inference_steps=2, return_latents=True)
    la, la2, lb = pipe(input_image=img_a, **kw), pipe(input_image=img_a, **kw), pipe(input_image=img_b, **kw)
    assert la.shape == (16,) + GRID.latent_shape()[1:] and torch.isfinite(la).all()
    assert torch.equal(la, la2) and not torch.equal(la, lb)
    y = pipe._image_cond_latents(img_a, GRID, True, (30, 52), (15, 26))
    assert y.shape == (20,) + GRID.latent_shape()[1:]
    assert bool((y[:4, 0] == 1).all()) and bool((y[:4, 1:] == 0).all())
    with pytest.raises(ValueError, match="image-to-video"):
        pipe(**kw)
    t2v = WanVideoPipeline("cpu", torch.bfloat16, DiTHolder(syn.make_dit_state_dict(CFG), CFG), HashTextEncoder(CFG), PoolVAE(), ops=OracleOps())
    with pytest.raises(ValueError, match="text-to-video"):
        t2v(input_image=img_a, **kw)


def test_fp8_torch_dtype_selects_fp8_projections():
    sd = syn.make_dit_state_dict(CFG)
    pipe = WanVideoPipeline("cpu", torch.float8_e4m3fn, DiTHolder(sd, CFG), HashTextEncoder(CFG), PoolVAE(), ops=OracleOps())
    assert pipe.gemm_dtype == "fp8" and pipe.attn_dtype == "fp8"
    kw = dict(prompt="a street", negative_prompt="bad", height=GRID.height, width=GRID.width, num_frames=GRID.num_frames,
              seed=0, num_inference_steps=2, return_latents=True)
    l8 = pipe(**kw)
    assert pipe._engine.fp8 and pipe._engine.fp8_set == ("wqkv",)          # the default e4m3 set: the QKV projection (dit.WanDiT.FP8_DEFAULT)
    assert pipe._engine.layers[0]["wqkv"][0].dtype == torch.float8_e4m3fn and pipe._engine.layers[0]["f0_w"].dtype == torch.bfloat16
    pipe.gemm_dtype = pipe.attn_dtype = "bf16"    # engine is rebuilt when the mode changes
    l16 = pipe(**kw)
    assert not pipe._engine.fp8
    rel = float((l8 - l16).norm() / l16.norm())
    assert 0 < rel < 0.1, f"fp8 vs bf16 latents rel-L2 {rel}"


def test_reference_rounding_switch_matches_oracle():
    """pipe.reference_rounding: bf16-rounded timestep (937.5 -> 936), bf16 noise, bf16-rounded CFG combine / Euler update —
    the host pipeline (with the oracle operator set) follows the oracle's restatement of the same rounding points, and
    the switch changes the result (ORACLE_RISKS.md R1-R3)."""
    from oracle import wan_ref as R
    from infinicube_amd.videogen.scheduler import FlowMatchScheduler, round_through_bf16
    assert round_through_bf16(937.5) == 936.0 and FlowMatchScheduler(4, 5.0, True).timesteps[1] == round_through_bf16(FlowMatchScheduler(4).timesteps[1])
    sd, bsd = syn.make_dit_state_dict(CFG), syn.make_buffer_embedder_state_dict(CFG)
    pipe = WanVideoPipeline("cpu", torch.bfloat16, DiTHolder(sd, CFG), HashTextEncoder(CFG), PoolVAE(), ops=OracleOps())
    pipe.initialize_buffer_embedder(16, zero_init=True).load_state_dict(bsd)
    sem, co = syn.make_dummy_buffers(GRID)
    from PIL import Image
    kw = dict(prompt="a street", negative_prompt="bad", semantic_buffer_video=[Image.fromarray(f) for f in sem],
              coordinate_buffer_video=[Image.fromarray(f) for f in co], height=GRID.height, width=GRID.width,
              num_frames=GRID.num_frames, seed=0, num_inference_steps=3, return_latents=True)
    plain = pipe(**kw)
    pipe.reference_rounding = True
    rounded = pipe(**kw)
    assert not torch.equal(plain, rounded)
    assert torch.equal(rounded, rounded.to(torch.bfloat16).float()), "with reference rounding the latent is bf16-representable"
    # oracle arm with the same rounding points
    from infinicube_amd.videogen.pipeline import _video_to_tensor
    vae, te = PoolVAE(), HashTextEncoder(CFG)
    bl = torch.cat([vae.encode(_video_to_tensor(kw[k], GRID.height, GRID.width)) for k in ("semantic_buffer_video", "coordinate_buffer_video")], 0)
    noise = syn.make_latent_noise(GRID, seed=0)
    sdr, bsdr = R.round_state_dict_to_bf16(sd), R.round_state_dict_to_bf16(bsd)
    ref = R.denoise_loop(sdr, bsdr, CFG, noise, te.encode("a street"), te.encode("bad"), bl, num_steps=3, reference_rounding=True)
    assert R.psnr(rounded, ref) > 40.0, R.psnr(rounded, ref)
    ref_plain = R.denoise_loop(sdr, bsdr, CFG, noise, te.encode("a street"), te.encode("bad"), bl, num_steps=3)
    assert R.psnr(rounded, ref) > R.psnr(rounded, ref_plain), "the rounded pipeline must sit closer to the rounded oracle than to the exact one"


def test_from_pretrained_fp8_keeps_encoders_in_bf16(tmp_path, monkeypatch):
    """torch_dtype=float8_e4m3fn selects the DiT's fp8 mode; the UMT5 / CLIP / VAE loaders must not be handed e4m3."""
    import infinicube_amd.videogen.clip_vision as cv
    import infinicube_amd.videogen.text_encoder as te
    import infinicube_amd.videogen.vae as vae
    seen = {}
    monkeypatch.setattr(te, "load_umt5_encoder", lambda pat, dev, dt, tok=None: seen.setdefault("t5", dt) and HashTextEncoder(CFG))
    monkeypatch.setattr(cv, "load_clip_vision", lambda pat, dev, dt: seen.setdefault("clip", dt))
    monkeypatch.setattr(vae, "load_wan_vae", lambda pat, dev: PoolVAE())
    d = tmp_path / "m"
    d.mkdir()
    save_file({k: v.contiguous() for k, v in syn.make_dit_state_dict(CFG).items()}, str(d / "diffusion_pytorch_model.safetensors"))
    mcs = [ModelConfig(path=str(d / n)) for n in ("diffusion_pytorch_model*.safetensors", "models_t5_umt5-xxl-enc-bf16.pth",
                                                  "Wan2.1_VAE.pth", "models_clip_open-clip-xlm-roberta-large-vit-huge-14.pth")]
    pipe = WanVideoPipeline.from_pretrained(torch_dtype=torch.float8_e4m3fn, device="cpu", model_configs=mcs)
    assert pipe.gemm_dtype == "fp8" and seen == {"t5": torch.bfloat16, "clip": torch.bfloat16}
    seen.clear()
    WanVideoPipeline.from_pretrained(torch_dtype=torch.float16, device="cpu", model_configs=mcs)
    assert seen == {"t5": torch.float16, "clip": torch.float16}


def test_checkpoint_inventory(tmp_path, capsys):
    """Counterpart of the reference's download script: the same six (model_id, pattern) entries, present / missing,
    and the architecture implied by a DiT shard's tensor shapes."""
    from infinicube_amd.videogen import download_checkpoint as dc
    assert [(m, p) for m, p, _ in dc.REQUIRED][4:] == [
        ("Wan-AI/Wan2.1-I2V-14B-480P", "diffusion_pytorch_model*.safetensors"),
        ("Wan-AI/Wan2.1-I2V-14B-480P", "models_clip_open-clip-xlm-roberta-large-vit-huge-14.pth")]
    root = tmp_path / "models"
    d = root / "Wan-AI" / "Wan2.1-I2V-14B-480P"
    d.mkdir(parents=True)
    sd = syn.make_dit_state_dict(preset("tiny-i2v"))
    keys = sorted(sd)
    save_file({k: sd[k].contiguous() for k in keys[: len(keys) // 2]}, str(d / "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({k: sd[k].contiguous() for k in keys[len(keys) // 2:]}, str(d / "diffusion_pytorch_model-00002-of-00002.safetensors"))
    rc = dc.main(["--models-root", str(root), "--inspect", "--no-download"])
    out = capsys.readouterr().out
    assert rc == 1 and out.count("MISSING") == 5 and "2 file(s)" in out
    assert "in_dim 36, image branch (64)" in out


def test_checkpoint_download_behaviour(tmp_path, capsys, monkeypatch):
    """The reference's download script fetches what is missing [R infinicube/videogen/download_checkpoint.py:19-31]: missing
    entries go through the hub's snapshot_download into models/<model_id>/ with the entry's file pattern; present ones are
    left alone; a hub failure (no network) is reported and the inventory still printed; skip_download=True never fetches."""
    import huggingface_hub
    from infinicube_amd.videogen import download_checkpoint as dc
    from infinicube_amd.videogen.pipeline import ModelConfig
    monkeypatch.setenv("ICV_DOWNLOAD_SOURCE", "huggingface")
    root = tmp_path / "models"
    have = root / "Wan-AI" / "Wan2.1-T2V-14B"
    have.mkdir(parents=True)
    (have / "Wan2.1_VAE.pth").write_bytes(b"x")
    calls = []

    def fake(repo_id, allow_patterns, local_dir):
        calls.append((repo_id, tuple(allow_patterns)))
        if "I2V" in repo_id:
            raise ConnectionError("no route to the hub")
        os.makedirs(local_dir, exist_ok=True)
        open(os.path.join(local_dir, allow_patterns[0].replace("*", "-00001-of-00001")), "wb").write(b"x")
        return local_dir

    monkeypatch.setattr(huggingface_hub, "snapshot_download", fake)
    rc = dc.main(["--models-root", str(root)])
    out = capsys.readouterr().out
    assert ("Wan-AI/Wan2.1-T2V-14B", ("Wan2.1_VAE.pth",)) not in calls and len(calls) == 5
    assert ("Wan-AI/Wan2.1-T2V-1.3B", ("diffusion_pytorch_model*.safetensors",)) in calls
    assert out.count("could not download") == 2 and out.count("MISSING") == 2 and rc == 1
    assert os.path.exists(root / "Wan-AI" / "Wan2.1-T2V-1.3B" / "diffusion_pytorch_model-00001-of-00001.safetensors")
    calls.clear()
    mc = ModelConfig(model_id="Wan-AI/Wan2.1-I2V-14B-480P", origin_file_pattern="diffusion_pytorch_model*.safetensors", skip_download=True,
                     local_model_path=str(root))
    with pytest.raises(FileNotFoundError):
        from infinicube_amd.videogen.pipeline import WanVideoPipeline
        WanVideoPipeline.from_pretrained(device="cpu", model_configs=[mc])
    assert calls == []
    # an interrupted download (1 of 2 shards on disk) is NOT "present": by the shard names, and by the index's weight map
    part = root / "Wan-AI" / "partial"
    part.mkdir(parents=True)
    (part / "diffusion_pytorch_model-00001-of-00002.safetensors").write_bytes(b"x")
    mc2 = ModelConfig(model_id="Wan-AI/partial", origin_file_pattern="diffusion_pytorch_model*.safetensors", local_model_path=str(root))
    assert not mc2.present()
    (part / "diffusion_pytorch_model-00002-of-00002.safetensors").write_bytes(b"x")
    assert mc2.present()
    import json
    (part / "diffusion_pytorch_model.safetensors.index.json").write_text(json.dumps({"weight_map": {
        "a": "diffusion_pytorch_model-00001-of-00002.safetensors", "b": "diffusion_pytorch_model-00003-of-00003.safetensors"}}))
    assert not mc2.present()


def test_first_contact_tool_reads_the_checkpoint_header(tmp_path, capsys):
    """tools/first_contact.py (the script that settles ORACLE_RISKS R1-R7 once real weights exist): its header-only steps on
    synthetic checkpoints of both embedder hypotheses and of an unknown layout, and the structural checks on a DiT key list."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("first_contact", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "first_contact.py"))
    fc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fc)
    cfg = preset("tiny")
    for variant, want in (("concat", "H1"), ("dual", "H2")):
        bsd = syn.make_buffer_embedder_state_dict(cfg, variant=variant)
        path = str(tmp_path / f"{variant}.safetensors")
        save_file({**{"buffer_embedder." + k: v.contiguous() for k, v in bsd.items()}, "dit.blocks.0.modulation": torch.zeros(1, 6, cfg.dim), "stray": torch.zeros(1)}, path)
        assert fc.main(["--checkpoint", path, "--describe-only"]) == 0
        rec = json.loads(capsys.readouterr().out)
        assert rec["R4_buffer_embedder"]["hypothesis"] == want and rec["dit_overlay_keys"] == 1 and rec["unprefixed_keys_dropped_by_the_loader"] == ["stray"]
    path = str(tmp_path / "odd.safetensors")
    save_file({"buffer_embedder.stem.0.weight": torch.zeros(8, 3, 3, 3)}, path)
    assert fc.main(["--checkpoint", path, "--describe-only"]) == 2 and "unknown" in capsys.readouterr().out
    ok, f = fc.structural_checks({k: tuple(v.shape) for k, v in syn.make_dit_state_dict(cfg).items()})
    assert ok and f["R8_rmsnorm_over_full_d"] and f["R9_norm3_affine"] and f["R9_modulation_shape"]
