"""libicvideo's shifted-row convolution (csrc/conv.hip) on the Wan-VAE's main layer shapes at the sizes a 480p tile really has, next to
MIOpen's best kernel (stock F.conv3d, bf16, NDHWC, find mode) on the same operands.  Run on the GPU box.
TF/s = algorithmic flops of the REAL positions (2 * T*H*W * taps * Cin * Cout) / time: the halo rows the shifted-row form also
computes (1.3 % at 240 x 416, 10 % at 30 x 52) count as overhead, not as work."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from infinicube_amd.videogen import vae as V, vae_hip as VH

dev = torch.device("cuda", 0)
hip = VH.VaeHip(nn.Identity(), dev)
torch.backends.cudnn.benchmark = True
SHAPES = [  # (what, module factory, taps, (T, H, W))
    ("3x3x3  96 ->  96  (decoder stage 3, 6 per tile)", lambda: V.CausalConv3d(96, 96, 3, padding=1), VH.TAPS_333, (93, 240, 416)),
    ("3x3x3 192 -> 192  (decoder stage 2, 6 per tile)", lambda: V.CausalConv3d(192, 192, 3, padding=1), VH.TAPS_333, (93, 120, 208)),
    ("3x3x3 384 -> 384  (decoder stage 1, 5 per tile)", lambda: V.CausalConv3d(384, 384, 3, padding=1), VH.TAPS_333, (47, 60, 104)),
    ("3x3x3 384 -> 384  (decoder stage 0 / middle)", lambda: V.CausalConv3d(384, 384, 3, padding=1), VH.TAPS_333, (24, 30, 52)),
    ("1x3x3 192 ->  96  (after the last upsample)", lambda: nn.Conv2d(192, 96, 3, padding=1), VH.TAPS_133, (93, 240, 416)),
    ("(3,1,1) 384 -> 768 (temporal upsample)", lambda: V.CausalConv3d(384, 768, (3, 1, 1), padding=(1, 0, 0)), VH.TAPS_311, (46, 60, 104)),
    ("3x3x3  96 ->   3  (decoder head)", lambda: V.CausalConv3d(96, 3, 3, padding=1), VH.TAPS_333, (93, 240, 416)),
]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


print("| layer | positions | libicvideo ms | TF/s | MIOpen (find) ms | TF/s | ratio |")
print("|---|---|---|---|---|---|---|")
for what, make, taps, (T, H, W) in SHAPES:
    torch.manual_seed(0)
    mod = make().to(dev, torch.bfloat16)
    cin, cout = mod.weight.shape[1], mod.weight.shape[0]
    x = (torch.randn((1, cin, T, H, W), device=dev) * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    xv = hip._to_vol(x, cin)
    flops = 2.0 * T * H * W * len(taps) * cin * cout
    t_hip = timeit(lambda: hip.conv(xv, mod, taps))
    if isinstance(mod, V.CausalConv3d):
        mod.fold_pad = True
        mod.weight.data = mod.weight.data.contiguous(memory_format=torch.channels_last_3d)
        ref = lambda: mod(x)
    else:
        mod.weight.data = mod.weight.data.contiguous(memory_format=torch.channels_last)
        xf = x[0].permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)
        ref = lambda: mod(xf)
    with torch.no_grad():
        t_ref = timeit(ref)
    print(f"| {what} | {T}x{H}x{W} | {t_hip:.3f} | {flops / t_hip / 1e9:.0f} | {t_ref:.3f} | {flops / t_ref / 1e9:.0f} | x{t_ref / t_hip:.2f} |", flush=True)
    del x, xv
    torch.cuda.empty_cache()
