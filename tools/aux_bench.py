"""Wall-clock of the stock-PyTorch components either side of the loop on the GPU box:
Wan-VAE encode / decode of one 93x480x832 clip, random weights, bf16.   env: WHAT=encode,decode  FIND=0|1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.vae import WanVAE, WanVAENet

torch.manual_seed(0)
torch.backends.cudnn.benchmark = os.environ.get("FIND", "0") == "1"
dev = "cuda:0"
vae = WanVAE(WanVAENet(), dev, torch.bfloat16)
F_, H, W = int(os.environ.get("FRAMES", 93)), int(os.environ.get("H", 480)), int(os.environ.get("W", 832))
video = torch.rand((3, F_, H, W)) * 2 - 1
lat = torch.randn((16, (F_ - 1) // 4 + 1, H // 8, W // 8))
if os.environ.get("COMPARE"):     # the tuned layers vs the plain composite ones on the same weights / input
    os.environ.update(ICV_VAE_PAD="copy")
    torch.manual_seed(0)
    ref = WanVAE(WanVAENet(), dev, torch.bfloat16)
    a, b = vae.decode(lat, tiled=True), ref.decode(lat, tiled=True)
    print(f"decode fold_pad={vae.fold_pad} vs plain: max |diff| {float((a - b).abs().max()):.4f} on [-1, 1] frames, "
          f"{float(((a - b).abs() > 1 / 255).float().mean()) * 100:.3f} % of values differ by more than 1/255, PSNR {float(10 * torch.log10(4.0 / ((a - b) ** 2).mean())):.1f} dB", flush=True)
    a, b = vae.encode(video, tiled=True), ref.encode(video, tiled=True)
    print(f"encode vs plain: max |diff| {float((a - b).abs().max()):.4f}, rms latent {float(b.pow(2).mean().sqrt()):.3f}", flush=True)
    del ref
fns = {"encode": lambda: vae.encode(video, tiled=True), "decode": lambda: vae.decode(lat, tiled=True)}
for name in os.environ.get("WHAT", "encode,decode").split(","):
    fn = fns[name]
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); tw = time.perf_counter() - t0
    t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    print(f"VAE {name} tiled fold_pad={vae.fold_pad} find={torch.backends.cudnn.benchmark}: first call {tw:.1f} s, second {time.perf_counter() - t0:.2f} s  out {tuple(out.shape)}  "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
