import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps
from oracle import wan_ref as R
ops = HipOps("cuda:0")
torch.manual_seed(0)
for Sq, H, chunks in ((36, 2, [24, 24, 24]), (36, 2, [64, 64]), (300, 2, [24, 24, 24]), (36, 2, [24]), (36, 2, [24, 100])):
    d = H * 128
    Skv = sum(chunks)
    q = torch.randn((Sq, d)).to(torch.bfloat16).cuda(); k = torch.randn((Skv, d)).to(torch.bfloat16).cuda(); v = torch.randn((Skv, d)).to(torch.bfloat16).cuda()
    ref = R.attention(q.float().cpu(), k.float().cpu(), v.float().cpu(), H)
    acc = torch.zeros((Sq, d), device="cuda"); ml = torch.zeros((Sq, H, 2), device="cuda"); o = torch.zeros((Sq, d), dtype=torch.bfloat16, device="cuda")
    lo = 0
    for j, c in enumerate(chunks):
        ops.attention_chunk(q, k[lo:lo + c], v[lo:lo + c], o, acc, ml, H, 128 ** -0.5, first=(j == 0), last=(j == len(chunks) - 1))
        lo += c
    torch.cuda.synchronize()
    e = (o.float().cpu() - ref)
    print(Sq, H, chunks, "max err", float(e.abs().max()), "rel", float(e.norm() / ref.norm()))
