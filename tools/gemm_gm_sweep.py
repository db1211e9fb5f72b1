"""Sweep of gemm256's grouped tile order (option gemm256_gm = M-tiles per group: an XCD's 32 concurrent tiles form a gm x 32/gm block)
on the 14B shapes, interleaved rounds, median TF/s (run on the GPU box).  GMS=2,3,4 ROUNDS=5 python tools/gemm_gm_sweep.py"""
import math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps, EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32
ops = HipOps("cuda:0")
GMS = [int(x) for x in os.environ.get("GMS", "1,2,3,4,6,8,16").split(",")]
ROUNDS = int(os.environ.get("ROUNDS", "3"))
S = 37440 * int(os.environ.get("M_MULT", "1"))      # M_MULT=2: the CFG-batched pair's 2S rows (the single-rank default)
for name, M, N, K, epi in (("14b qkv", S, 15360, 5120, EPI_BF16), ("14b o", S, 5120, 5120, EPI_RESID_F32), ("14b xq", S, 5120, 5120, EPI_BF16),
                           ("14b ffn1", S, 13824, 5120, EPI_GELU_BF16), ("14b ffn2", S, 5120, 13824, EPI_RESID_F32),
                           ("sp4 ffn1", 9360, 13824, 5120, EPI_GELU_BF16), ("sp4 o", 9360, 5120, 5120, EPI_RESID_F32),
                           ("1.3b ffn1", S, 8960, 1536, EPI_GELU_BF16), ("1.3b qkv", S, 4608, 1536, EPI_BF16)):
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn((N,), device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.float32 if epi == EPI_RESID_F32 else torch.bfloat16)
    kw = dict(resid=out, gate=bias) if epi == EPI_RESID_F32 else {}
    res = {g: [] for g in GMS}
    for rnd in range(ROUNDS):
        for gm in GMS:
            ops.lib.icv_set_option(b"gemm256_gm", gm)
            for _ in range(2):
                ops.gemm(a, w, bias, out, epi, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.gemm(a, w, bias, out, epi, **kw)
            e1.record(); torch.cuda.synchronize()
            res[gm].append(2.0 * M * N * K / (e0.elapsed_time(e1) / 8) / 1e9)
    print(f"{name:10s}", " | ".join(f"gm {g}: {statistics.median(v):7.1f}" for g, v in res.items()), flush=True)
ops.lib.icv_set_option(b"gemm256_gm", 4)
