"""csrc/attn7q.hip (software-pipelined bf16 self-attention) against the default launch (one piece of attn7p.hip): output difference vs the
fp32 reference on small shapes (ragged key counts, carried-state chunks), then interleaved timings at the 14B shapes."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps
ops = HipOps("cuda:0")
LN2 = math.log(2.0)
fold = 128 ** -0.5 * math.log2(math.e)
torch.manual_seed(0)


def run(var, q, k, v, H, chunks=None):
    ops.lib.icv_set_option(b"attn7q", var)
    o = torch.zeros_like(q)
    if chunks is None:
        ops.attention(q, k, v, o, H, LN2)
    else:
        acc = torch.zeros((q.shape[0], q.shape[1]), dtype=torch.float32, device=q.device)
        ml = torch.zeros((q.shape[0], H, 2), dtype=torch.float32, device=q.device)
        for i, (a, b) in enumerate(zip(chunks[:-1], chunks[1:])):
            ops.attention_chunk(q, k[a:b], v[a:b], o, acc, ml, H, LN2, first=i == 0, last=i == len(chunks) - 2)
    ops.lib.icv_set_option(b"attn7q", 0)
    return o


for Sq, Skv, H, chunks in () if os.environ.get('ATTN7Q_TIMING_ONLY') else ((300, 1100, 2, None), (513, 2048, 3, None), (64, 3000, 1, None), (700, 64 * 37 + 11, 3, None), (257, 1025, 1, None),
                           (300, 5000, 2, [0, 1100, 2500, 5000]), (1, 1030, 1, None), (520, 4097, 2, [0, 2049, 4097])):
    d = H * 128
    q = torch.randn((Sq, d), device="cuda").to(torch.bfloat16)
    kf = torch.randn((Skv, d), device="cuda")
    kf[Skv - 1] = q[min(3, Sq - 1)].float() * 3.0
    kf[5] = q[0].float() * 2.0
    k = (kf * fold).to(torch.bfloat16)
    v = torch.randn((Skv, d), device="cuda").to(torch.bfloat16)
    s = torch.einsum("qhd,khd->hqk", q.float().view(Sq, H, 128), k.float().view(Skv, H, 128)) * LN2
    ref = torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), v.float().view(Skv, H, 128)).reshape(Sq, d)
    o0 = run(0, q, k, v, H, chunks).float()
    line = f"Sq={Sq} Skv={Skv} H={H} chunks={chunks}: default rms err {float((o0 - ref).pow(2).mean().sqrt()):.3e}"
    for var in (1, 2):
        o1 = run(var, q, k, v, H, chunks).float()
        line += f" | attn7q({var}) rms err {float((o1 - ref).pow(2).mean().sqrt()):.3e}, max |diff| vs default {float((o1 - o0).abs().max()):.3g}, finite {bool(torch.isfinite(o1).all())}"
    print(line)

def timeit(fn, n=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for Sq, Skv, H in ((37440, 37440, 40),) if os.environ.get('ATTN7Q_TIMING_ONLY') else ((37440, 37440, 40), (9360, 37440, 40), (4680, 37440, 40), (37440, 37440, 12)):
    d = H * 128
    q = torch.randn((Sq, d), device="cuda").to(torch.bfloat16)
    k = (torch.randn((Skv, d), device="cuda") * fold).to(torch.bfloat16)
    v = torch.randn((Skv, d), device="cuda").to(torch.bfloat16)
    o = torch.empty_like(q)
    fl = 4.0 * Sq * Skv * d / 1e9
    res = {}
    for rep in range(3):
        for var in (0, 1, 2):
            ops.lib.icv_set_option(b"attn7q", var)
            res.setdefault(var, []).append(timeit(lambda: ops.attention(q, k, v, o, H, LN2)))
    ops.lib.icv_set_option(b"attn7q", 0)
    print(f"n = {Sq} x {Skv} keys, {H} heads: " + " | ".join(f"{'attn7p' if var == 0 else f'attn7q({var})'} {min(t):.3f} ms = {fl / min(t):.0f} TF/s" for var, t in res.items()))
