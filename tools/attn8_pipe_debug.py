"""Which key tile does a variant of attn8 weigh differently from variant 0?  V[key, c] = 1 for keys of tile c (c < 128): output column c
= the softmax mass of tile c."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps
ops = HipOps("cuda:0")
torch.manual_seed(0)
for Sq, Skv in ((300, 3000), (300, 3008), (300, 2048), (64, 5000)):
    H, d = 1, 128
    fold = 128 ** -0.5 * math.log2(math.e)
    q = torch.randn((Sq, d), device="cuda").to(torch.bfloat16)
    k = (torch.randn((Skv, d), device="cuda") * fold).to(torch.bfloat16)
    v = torch.zeros((Skv, d), device="cuda")
    tile = torch.arange(Skv, device="cuda") // 64
    v[torch.arange(Skv, device="cuda"), tile.clamp(max=127)] = 1.0
    v = v.to(torch.bfloat16)
    outs = {}
    for var in (0, 8, 32, 676):
        ops.lib.icv_set_option(b"attn8_variant", var)
        o = torch.empty_like(q)
        ws = ops.attention_fp8_buffers(Sq, Skv, d, H)
        ops.attention_fp8(q, k, v, o, H, ws)
        outs[var] = o.float().cpu()
    nt = (Skv + 63) // 64
    for var in (8, 32, 676):
        diff = (outs[var] - outs[0]).abs().mean(0)[:nt]
        print(f"Sq={Sq} Skv={Skv} nt={nt} variant {var}: mean |mass diff| per tile: max {float(diff.max()):.4f} at tile {int(diff.argmax())}; first 4 {[round(float(x), 4) for x in diff[:4]]} last 4 {[round(float(x), 4) for x in diff[-4:]]}; row sums {float(outs[var][:, :nt].sum(1).mean()):.4f} vs {float(outs[0][:, :nt].sum(1).mean()):.4f}")
ops.lib.icv_set_option(b"attn8_variant", -1)
