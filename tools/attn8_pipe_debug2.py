import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps
ops = HipOps("cuda:0")
torch.manual_seed(0)
for Sq, Skv, base in ((64, 3000, 64), (64, 3000, 2944 - 64), (64, 3064, 64), (64, 3000, 0)):
    H, d = 1, 128
    fold = 128 ** -0.5 * math.log2(math.e)
    q = torch.randn((Sq, d), device="cuda").to(torch.bfloat16)
    k = (torch.randn((Skv, d), device="cuda") * fold * 0.01).to(torch.bfloat16)      # near-uniform attention
    v = torch.zeros((Skv, d), device="cuda")
    idx = torch.arange(base, min(base + 128, Skv), device="cuda")
    v[idx, idx - base] = 1.0
    v = v.to(torch.bfloat16)
    outs = {}
    for var in (0, 32):
        ops.lib.icv_set_option(b"attn8_variant", var)
        o = torch.empty_like(q)
        ws = ops.attention_fp8_buffers(Sq, Skv, d, H)
        ops.attention_fp8(q, k, v, o, H, ws)
        outs[var] = o.float().cpu().mean(0) * Skv
    print(f"Skv={Skv} keys {base}..{base+127}: per-key mass x Skv (variant 0 | 32)")
    for r in range(0, 128, 16):
        print("  ", " ".join(f"{float(x):.2f}" for x in outs[0][r:r + 16]), "|", " ".join(f"{float(x):.2f}" for x in outs[32][r:r + 16]))
ops.lib.icv_set_option(b"attn8_variant", -1)
