"""Is the pieces kernel (csrc/attn7p.hip) with ONE piece faster than the plain attn7 launch at the single-GPU shape?  Same tiles, same
arithmetic (bit-identical); interleaved timing.  Run on the GPU box."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps
ops = HipOps("cuda:0")
H, d = 40, 5120
SCALE = math.log(2.0)
torch.manual_seed(0)
for S, n in ((37440, 37440), (37440, 9360), (37440, 4680)):
    qkv = torch.randn((3, S, d), device="cuda").to(torch.bfloat16)
    qkv[1] = (qkv[1].float() * (128 ** -0.5 * math.log2(math.e))).to(torch.bfloat16)
    q, k, v = qkv[0][:n], qkv[1], qkv[2]
    o1, o2 = torch.empty_like(q), torch.empty_like(q)
    fa = lambda: ops.attention(q, k, v, o1, H, SCALE)
    fb = lambda: ops.attention_pieces(q, [(k, v, -1, 0)], o2, H, SCALE)
    fa(); fb(); torch.cuda.synchronize()
    assert torch.equal(o1, o2)
    res = {"attn7": [], "attn7p": []}
    for rep in range(6):
        for name, f in (("attn7", fa), ("attn7p", fb)) if rep % 2 == 0 else (("attn7p", fb), ("attn7", fa)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                f()
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 4)
    fl = 4.0 * n * S * d
    a, b = sorted(res["attn7"])[len(res["attn7"]) // 2], sorted(res["attn7p"])[len(res["attn7p"]) // 2]
    print(f"n = {n:6d} x {S} keys: attn7 {a:8.3f} ms ({fl / a / 1e9:7.1f} TF/s)   attn7p, one piece {b:8.3f} ms ({fl / b / 1e9:7.1f} TF/s)   attn7p / attn7 time {b / a:.4f}   runs attn7 {['%.3f' % x for x in res['attn7']]} attn7p {['%.3f' % x for x in res['attn7p']]}")
