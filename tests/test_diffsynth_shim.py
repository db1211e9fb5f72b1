"""B-inner: the reference's OWN `infinicube/videogen/inference.py`, unmodified, on top of this repo's
`diffsynth` shim — the four names it imports (`load_state_dict`, `save_video`, `ModelConfig`, `WanVideoPipeline`
[R infinicube/videogen/inference.py:25-26]).  Runs only where the reference checkout exists (the build container):
the reference file is loaded by path (never copied), `from_pretrained` is pointed at an in-memory tiny model (no
checkpoints exist offline), and the frames it produces must be the frames this repo's own `WanVideoGenerator`
produces from the same inputs."""
import contextlib
import importlib.util
import io
import os

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

REF_FILE = "/root/reference/infinicube/videogen/inference.py"


@pytest.mark.skipif(not os.path.isfile(REF_FILE), reason="needs the reference checkout (build container only)")
def test_reference_wrapper_runs_on_the_diffsynth_shim(tmp_path, monkeypatch):
    import diffsynth
    import diffsynth.pipelines.wan_video_new as shim
    import mgpu_factory as F
    from infinicube.videogen import WanVideoGenerator as Ours
    from infinicube_amd.videogen import synthetic as syn
    assert {"load_state_dict", "save_video"} <= set(dir(diffsynth)) and {"ModelConfig", "WanVideoPipeline"} <= set(dir(shim))
    seen = {}

    def from_pretrained(torch_dtype=None, device=None, model_configs=None, **kw):
        seen["configs"] = [(m.model_id, m.origin_file_pattern, m.skip_download) for m in model_configs]
        return F.factory(torch_dtype, device, model_configs)

    monkeypatch.setattr(shim.WanVideoPipeline, "from_pretrained", staticmethod(from_pretrained))
    spec = importlib.util.spec_from_file_location("ref_videogen_inference", REF_FILE)
    ref = importlib.util.module_from_spec(spec)
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    spec.loader.exec_module(ref)            # imports `diffsynth` -> the shim
    bsd = syn.make_buffer_embedder_state_dict(F.CFG)
    path = str(tmp_path / "step-1.safetensors")
    save_file({"buffer_embedder." + k: v for k, v in bsd.items()}, path)
    sem, co = syn.make_dummy_buffers(F.GRID)
    out = str(tmp_path / "video_480p_front.mp4")
    with contextlib.redirect_stdout(io.StringIO()) as log_ref:
        g = ref.WanVideoGenerator(path, device="cpu", use_wan_1pt3b=True)
        frames = g.generate(sem, co, seed=0, output_path=out)
    assert seen["configs"][0] == ("Wan-AI/Wan2.1-T2V-1.3B", "diffusion_pytorch_model*.safetensors", True)
    assert os.path.getsize(out) > 0 and len(frames) == F.GRID.num_frames
    with contextlib.redirect_stdout(io.StringIO()) as log_ours:
        g2 = Ours(path, device="cpu", use_wan_1pt3b=True, pipeline_factory=F.factory)
        frames2 = g2.generate(sem, co, seed=0, output_path=str(tmp_path / "ours.mp4"))
    a, b = np.stack([np.asarray(f) for f in frames]), np.stack([np.asarray(f) for f in frames2])
    assert np.array_equal(a, b), "the reference wrapper on the shim and this repo's wrapper must produce the same frames"
    strip = lambda s: s.replace("ours.mp4", "video_480p_front.mp4")   # noqa: E731
    assert log_ref.getvalue() == strip(log_ours.getvalue()), "progress lines differ between the reference wrapper and ours"
