"""Which of the six per-layer projections can run in e4m3 at the REAL Wan2.1-1.3B depth (config #1 grid, 10 steps with
CFG) and keep the final latent >= 40 dB?  Reference arm = this build's own bf16 path (60.7 dB from the fp32 oracle in
tests/test_dit_gpu.py, i.e. 20 dB below the quantisation noise measured here); prints PSNR per subset.
STUDY=14b: the same question at the real Wan2.1-14B depth and the metric's size (40 layers, d = 5120, S = 37 440, a 4-step
CFG loop; the bf16 arm is 50.2 dB from the fp32 oracle there, tests/test_fullsize_gpu.py), a shorter subset list."""
import itertools
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infinicube_amd.videogen import synthetic as syn
from infinicube_amd.videogen.config import GRID_480P, GRID_CFG1, preset
from infinicube_amd.videogen.dit import WanDiT
from infinicube_amd.videogen.ops import HipOps
from infinicube_amd.videogen.scheduler import FlowMatchScheduler


def psnr(a, b):
    a, b = a.double(), b.double()
    mse = float(((a - b) ** 2).mean())
    return 10.0 * math.log10(float(b.abs().max()) ** 2 / mse)


ops = HipOps("cuda:0")
BIG = os.environ.get("STUDY", "1.3b") == "14b"
cfg, grid, STEPS = (preset("14b"), GRID_480P, 4) if BIG else (preset("1.3b"), GRID_CFG1, 10)
sd = syn.make_dit_state_dict(cfg, seed=0, dtype=torch.bfloat16, **({"device": "cuda:0"} if BIG else {}))
bsd = syn.make_buffer_embedder_state_dict(cfg, dtype=torch.bfloat16, **({"device": "cuda:0"} if BIG else {}))
noise = syn.make_latent_noise(grid)
c1, c2, bl = syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2), syn.make_buffer_latents(cfg, grid)


def run(**kw):
    m = WanDiT(cfg, sd, ops, bsd, **kw).prepare(grid)
    lat = noise.clone().to("cuda:0")
    m.denoise(lat, m.encode_context(c1), m.encode_context(c2), m.embed_buffers(bl), FlowMatchScheduler(STEPS), 5.0)
    torch.cuda.synchronize()
    del m
    torch.cuda.empty_cache()
    return lat.cpu()


ref = run()
ALL = WanDiT.FP8_WEIGHTS
subsets = [ALL] + [(w,) for w in ALL] + [tuple(x for x in ALL if x != w) for w in ALL]
subsets += [("wqkv", "f0_w"), ("wqkv", "f0_w", "f2_w"), ("wqkv", "f0_w", "xq_w"), ("wqkv", "xq_w", "f0_w", "f2_w"), ("f0_w", "f2_w"),
            ("wo", "xo_w", "f2_w"), ("wqkv", "wo", "f0_w", "f2_w")]
if BIG:
    print(f"bf16 projections, e4m3 self-attention: PSNR vs bf16 path {psnr(run(attn_dtype='fp8'), ref):6.2f} dB", flush=True)
    subsets = [(w,) for w in ALL] + [("wqkv", "wo"), ("wqkv", "wo", "xq_w", "xo_w"), ("xq_w", "xo_w"), ("f0_w", "f2_w"),
                                    ("wqkv", "wo", "xq_w", "xo_w", "f0_w"), ALL]
for sub in subsets:
    for attn in (("bf16",) if BIG and len(sub) == 1 else ("bf16", "fp8")):
        lat = run(gemm_dtype="fp8", attn_dtype=attn, fp8_weights=sub)
        print(f"fp8 {','.join(sub):40s} attn {attn}: PSNR vs bf16 path {psnr(lat, ref):6.2f} dB", flush=True)
