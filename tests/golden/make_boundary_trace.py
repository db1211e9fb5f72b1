#!/usr/bin/env python3
"""Generates tests/golden/boundary_trace.json by running the REFERENCE wrapper class
(/root/reference/infinicube/videogen/inference.py, loaded by path — nothing is copied) against a
recording stub of its third-party `diffsynth` dependency.  The JSON
holds only captured data: call names/kwargs, printed lines, exception types/messages.
Run in the build container only (needs /root/reference):  python tests/golden/make_boundary_trace.py
"""
import contextlib
import importlib.util
import io
import json
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REF = "/root/reference/infinicube/videogen/inference.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "boundary_trace.json")

LOG = []


def summarize(v):
    if isinstance(v, list) and v and isinstance(v[0], Image.Image):
        return {"pil_list": len(v), "size": list(v[0].size), "mode": v[0].mode}
    if isinstance(v, torch.dtype):
        return str(v)
    if isinstance(v, list):
        return [summarize(x) for x in v]
    if hasattr(v, "kw"):
        return {"ModelConfig": v.kw}
    if isinstance(v, dict):
        return {k: (list(x.shape) if hasattr(x, "shape") else summarize(x)) for k, x in v.items()}
    return v


class ModelConfig:
    def __init__(self, **kw):
        self.kw = kw


class _Module:
    def __init__(self, name):
        self.name = name

    def load_state_dict(self, sd, strict=True):
        LOG.append([f"{self.name}.load_state_dict", {"keys": sorted(sd.keys()), "strict": strict}])


class WanVideoPipeline:
    def __init__(self):
        self.dit = _Module("dit")
        self.buffer_embedder = None

    @classmethod
    def from_pretrained(cls, **kw):
        LOG.append(["from_pretrained", summarize(kw)])
        return cls()

    def initialize_buffer_embedder(self, **kw):
        LOG.append(["initialize_buffer_embedder", kw])
        self.buffer_embedder = _Module("buffer_embedder")

    def enable_vram_management(self, *a, **kw):
        LOG.append(["enable_vram_management", {"args": list(a), **kw}])

    def __call__(self, **kw):
        LOG.append(["__call__", summarize(kw)])
        return [Image.new("RGB", (kw["width"], kw["height"])) for _ in range(kw["num_frames"])]


def load_state_dict(path):
    LOG.append(["load_state_dict", {"path": path}])
    return {"buffer_embedder.proj.weight": torch.zeros(2), "buffer_embedder.proj.bias": torch.zeros(2),
            "dit.blocks.0.modulation": torch.zeros(2), "optimizer.step": torch.zeros(1)}


def save_video(frames, path, **kw):
    LOG.append(["save_video", {"n_frames": len(frames), "path": path, **kw}])


def main():
    ds = types.ModuleType("diffsynth")
    ds.load_state_dict, ds.save_video = load_state_dict, save_video
    pk = types.ModuleType("diffsynth.pipelines")
    wv = types.ModuleType("diffsynth.pipelines.wan_video_new")
    wv.ModelConfig, wv.WanVideoPipeline = ModelConfig, WanVideoPipeline
    sys.modules.update({"diffsynth": ds, "diffsynth.pipelines": pk, "diffsynth.pipelines.wan_video_new": wv})
    spec = importlib.util.spec_from_file_location("ref_videogen_inference", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    trace = {"source": "reference infinicube/videogen/inference.py vs recording diffsynth stub", "cases": {}}
    for name, kw in (("1.3b", dict(use_wan_1pt3b=True)), ("14b", dict())):
        LOG.clear()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            g = mod.WanVideoGenerator("ckpt.safetensors", device="cpu", **kw)
        ctor_out, ctor_log = buf.getvalue().splitlines(), json.loads(json.dumps(LOG))
        LOG.clear()
        sem = np.zeros((17, 256, 448, 3), np.uint8)
        co = np.ones((17, 256, 448, 3), np.uint8)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            frames = g(sem, co, seed=0, tiled=True, output_path="o.mp4")
        gen_out, gen_log = buf.getvalue().splitlines(), json.loads(json.dumps(LOG))
        LOG.clear()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            g.generate(sem, co, prompt="p", negative_prompt="n", seed=7, tiled=False)
        trace["cases"][name] = {"ctor_stdout": ctor_out, "ctor_calls": ctor_log, "generate_stdout": gen_out,
                                "generate_calls": gen_log, "returned_frames": len(frames),
                                "generate_nosave_stdout": buf.getvalue().splitlines(),
                                "generate_nosave_calls": json.loads(json.dumps(LOG))}
    errs = {}
    sem = np.zeros((17, 256, 448, 3), np.uint8)
    bad = {
        "float32": (sem.astype(np.float32), sem.astype(np.float32)),
        "last_dim_1": (sem[..., :1], sem[..., :1]),
        "ndim_3": (sem[0], sem[0]),
        "mismatch": (sem, sem[:16]),
        "not_ndarray": ([1, 2], [1, 2]),
    }
    for k, (a, b) in bad.items():
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                g.generate(a, b)
            errs[k] = None
        except Exception as e:  # noqa: BLE001
            errs[k] = {"type": type(e).__name__, "message": str(e)}
    trace["errors"] = errs
    import inspect
    sig = inspect.signature(mod.WanVideoGenerator.generate)
    trace["generate_defaults"] = {k: (v.default if v.default is not inspect._empty else None) for k, v in sig.parameters.items() if k != "self"}
    sig = inspect.signature(mod.WanVideoGenerator.__init__)
    trace["ctor_defaults"] = {k: (str(v.default) if v.default is not inspect._empty else None) for k, v in sig.parameters.items() if k != "self"}
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(trace, f, indent=1, ensure_ascii=False, default=str)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
