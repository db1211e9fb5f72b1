"""Config #5 at FULL depth and size (run on the GPU box; ~5 GPU-minutes): Wan2.1-14B image-to-video (36 input channels,
CLIP cross-attention branch), 93 frames 720x1280 (S = 86 400), ONE conditional forward of all 40 layers:
    product, bf16                       vs  oracle/wan_ref.py in fp32 by stock PyTorch on the GPU
    product, torch_dtype=float8_e4m3fn  vs  the same UNQUANTISED fp32 oracle (what the e4m3 mode costs in accuracy)
Prints rel-L2 / cosine of the velocity and the PSNR of a one-step Euler update from pure noise.
(The oracle is the checker, as in tests/test_fullsize_gpu.py; this script is test infrastructure.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from oracle import wan_ref as R  # noqa: E402
from infinicube_amd.videogen import synthetic as syn  # noqa: E402
from infinicube_amd.videogen.config import GRID_720P, preset  # noqa: E402
from infinicube_amd.videogen.dit import WanDiT  # noqa: E402
from infinicube_amd.videogen.ops import HipOps  # noqa: E402
from infinicube_amd.videogen.scheduler import FlowMatchScheduler  # noqa: E402

DEV = "cuda:0"
cfg, grid = preset("14b-i2v"), GRID_720P
ops = HipOps(DEV)
sd = syn.make_dit_state_dict(cfg, seed=0, device=DEV, dtype=torch.bfloat16)
bsd = syn.make_buffer_embedder_state_dict(cfg, device=DEV, dtype=torch.bfloat16)
noise, c1, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_buffer_latents(cfg, grid)
clip, y = syn.make_clip_features(cfg), syn.make_cond_latents(cfg, grid)
sched = FlowMatchScheduler(50)
ts = float(sched.timesteps[0])
gshape = (grid.T, grid.Hp, grid.Wp)
got = {}
for mode in ("bf16", "fp8"):
    kw = {} if mode == "bf16" else dict(gemm_dtype="fp8", attn_dtype="fp8")
    m = WanDiT(cfg, sd, ops, bsd, **kw).prepare(grid, graphs=False)
    ck = m.encode_context(c1, clip)
    add = m.embed_cond_latents(y, add_to=m.embed_buffers(bl))
    lat = noise.clone().to(DEV)
    torch.cuda.synchronize()
    t0 = time.time()
    m.forward_tokens(lat, ck, ts, add, m.head_out[0])
    torch.cuda.synchronize()
    got[mode] = (R.unpatchify(m.head_out[0].cpu(), gshape, cfg.out_dim), time.time() - t0)
    del m, ck, add
    torch.cuda.empty_cache()
sdr = {k: v.float() for k, v in sd.items()}
bsdr = {k: v.float() for k, v in bsd.items()}
del sd, bsd
torch.cuda.empty_cache()
t0 = time.time()
buf = R.buffer_embed(bsdr, bl.to(DEV))
v = R.dit_forward(sdr, cfg, noise.to(DEV), c1.to(DEV), ts, buf, clip_fea=clip.to(DEV), y=y.to(DEV)).cpu()
torch.cuda.synchronize()
t_ref = time.time() - t0
print(f"config #5, Wan2.1-14B i2v, 93 f 720x1280, S={grid.S}, one forward of {cfg.num_layers} layers; fp32 torch oracle on the GPU: {t_ref:.1f} s")
for mode, (vh, t) in got.items():
    rel = float((vh - v).norm() / v.norm())
    cos = float(torch.nn.functional.cosine_similarity(vh.flatten().double(), v.flatten().double(), dim=0))
    p = R.psnr(noise + vh * sched.dsigma(0), noise + v * sched.dsigma(0))
    print(f"  product {mode:4s}: {t:6.2f} s   velocity rel-L2 {rel:.4g}  cosine {cos:.6f}   latent PSNR after one Euler step {p:.1f} dB")
