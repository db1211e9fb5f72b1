"""BASELINE.json config #1 in FULL on the CPU path: Wan2.1-1.3B t2v, 17 frames 256x448 (S = 2 240), 10 deterministic steps with CFG 5, dummy
guidance buffers - the reference's own CPU-runnable case, here through oracle/wan_ref.py (the CPU restatement; the reference's DiT lives in an
absent fork).  BASELINE.md §3 promised this figure; bench.py's `cpu_baseline` is a bounded-sample extrapolation for the 14B / 480p metric.
Prints one line: seconds, threads, CPU model.  Usage: python tools/cpu_config1_full.py [threads] [--save-golden]
--save-golden also writes the final latent to tests/golden/config1_cpu_oracle_latent.npz: the fixture tests/test_dit_gpu.py::test_config1_...
compares the HIP loop (and the GPU-executed oracle) with AT THE FULL GRID - the CPU path itself as the checker of config #1."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infinicube_amd.videogen import synthetic as syn
from infinicube_amd.videogen.config import TokenGrid, preset
from oracle import wan_ref as R

args = [a for a in sys.argv[1:] if not a.startswith("--")]
threads = int(args[0]) if args else (os.cpu_count() or 1)
torch.set_num_threads(threads)
cfg, grid, steps = preset("1.3b"), TokenGrid(17, 256, 448), 10
sd = {k: v.float() for k, v in syn.make_dit_state_dict(cfg, seed=0, dtype=torch.bfloat16).items()}
bsd = {k: v.float() for k, v in syn.make_buffer_embedder_state_dict(cfg, dtype=torch.bfloat16).items()}
noise = syn.make_latent_noise(grid)
c1, c2, bl = syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2), syn.make_buffer_latents(cfg, grid)
t0 = time.time()
with torch.no_grad():
    lat = R.denoise_loop(sd, bsd, cfg, noise, c1, c2, bl, num_steps=steps)
dt = time.time() - t0
model = "?"
try:
    model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception:
    pass
flops = 2 * steps * 6.88e12          # F_fwd of config #1
print(f"config #1 in full on the CPU oracle (fp32, torch {torch.__version__}): Wan2.1-1.3B 17f 256x448, S = {grid.S}, {steps} steps x 2 forwards: "
      f"{dt:.1f} s = {dt / steps:.2f} s per denoise step = {steps / dt:.4f} steps/s on {threads} threads of {os.cpu_count()} logical CPUs ({model}); "
      f"{flops / dt / 1e12:.3f} TFLOP/s algorithmic; latent finite: {bool(torch.isfinite(lat).all())}, rms {float(lat.pow(2).mean().sqrt()):.4f}")

if "--save-golden" in sys.argv:
    import numpy as np
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config1_cpu_oracle_latent.npz")
    np.savez_compressed(out, latent=lat.numpy().astype(np.float32), noise_head=noise.flatten()[:16].numpy(), steps=steps, cfg_scale=5.0,
                        note="oracle/wan_ref.denoise_loop on CPU, Wan2.1-1.3B random-init (synthetic.make_dit_state_dict seed 0, bf16-rounded), 17f 256x448, 10 steps")
    print("golden latent written to", out, os.path.getsize(out), "bytes")
