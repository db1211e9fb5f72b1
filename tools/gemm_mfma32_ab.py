"""A/B: gemm256 with v_mfma_f32_32x32x16_bf16 fragments on the two-phase schedule vs the 16x16x32 default."""
import math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps, EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32
ops = HipOps("cuda:0")
S = 37440
ops.lib.icv_set_option(b"gemm256", 1)
for name, M, N, K, epi in [("14b qkv", S, 15360, 5120, EPI_BF16), ("14b o", S, 5120, 5120, EPI_RESID_F32), ("14b ffn1", S, 13824, 5120, EPI_GELU_BF16),
                           ("14b ffn2", S, 5120, 13824, EPI_RESID_F32), ("1.3b ffn1", S, 8960, 1536, EPI_GELU_BF16)]:
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn((N,), device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.float32 if epi == EPI_RESID_F32 else torch.bfloat16)
    kw = dict(resid=out, gate=bias) if epi == EPI_RESID_F32 else {}
    res = {}
    cfgs = [("mfma16 sched3", 16, 3), ("mfma32 sched0", 32, 0), ("mfma32 sched1", 32, 1)]
    times = {c[0]: [] for c in cfgs}
    for _ in range(4):
        for nm, mf, sch in cfgs:
            ops.lib.icv_set_option(b"gemm256_mfma", mf); ops.lib.icv_set_option(b"gemm256_sched", sch)
            ops.gemm(a, w, bias, out, epi, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm(a, w, bias, out, epi, **kw)
            e1.record(); torch.cuda.synchronize()
            times[nm].append(e0.elapsed_time(e1) / 5)
    print(f"{name:10s}: " + " | ".join(f"{nm}: {2.0 * M * N * K / statistics.median(t) / 1e9:7.1f} TF" for nm, t in times.items()), flush=True)
