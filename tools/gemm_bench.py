"""Micro-benchmark of icv_gemm_bf16 variants on the DiT shapes (run on the GPU box)."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps, EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32

ops = HipOps("cuda:0")
S = 37440
shapes = [("1.3b qkv", S, 4608, 1536, EPI_BF16), ("1.3b o", S, 1536, 1536, EPI_RESID_F32), ("1.3b ffn1", S, 8960, 1536, EPI_GELU_BF16),
          ("1.3b ffn2", S, 1536, 8960, EPI_RESID_F32), ("14b qkv", S, 15360, 5120, EPI_BF16), ("14b o", S, 5120, 5120, EPI_RESID_F32),
          ("14b ffn1", S, 13824, 5120, EPI_GELU_BF16), ("14b ffn2", S, 5120, 13824, EPI_RESID_F32), ("sp8 14b ffn1", 4680, 13824, 5120, EPI_GELU_BF16),
          ("sp8 14b o", 4680, 5120, 5120, EPI_RESID_F32), ("sp8 14b kv", 4680, 10240, 5120, EPI_BF16), ("sp8 14b ffn2", 4680, 5120, 13824, EPI_RESID_F32),
          ("sp4 14b o", 9360, 5120, 5120, EPI_RESID_F32), ("sp8 1.3b o", 4680, 1536, 1536, EPI_RESID_F32), ("sp8 1.3b ffn1", 4680, 8960, 1536, EPI_GELU_BF16)]
for name, M, N, K, epi in shapes:
    a = (torch.randn((M, K), device="cuda") ).to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn((N,), device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.float32 if epi == EPI_RESID_F32 else torch.bfloat16)
    res = {}
    for variant in (0, 1, 2, 3) + ((4, 5) if ops.lib.icv_set_option(b"require_experiments", 1) == 0 else ()):
        ops.lib.icv_set_option(b"gemm256", 4 if variant == 5 else 3 if variant == 4 else (min(variant, 1) if variant != 2 else 2))
        ops.lib.icv_set_option(b"gemm256_mfma", 32 if variant == 3 else 16)
        kw = dict(resid=out, gate=bias) if epi == EPI_RESID_F32 else {}
        for _ in range(2):
            ops.gemm(a, w, bias, out, epi, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 10
        for _ in range(n):
            ops.gemm(a, w, bias, out, epi, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        res[variant] = 2.0 * M * N * K / ms / 1e9
    # ceiling MEASUREMENT only (never linked into libicvideo.so): the vendor library PyTorch-ROCm dispatches bf16
    # F.linear to (hipBLASLt / rocBLAS), same operands, no epilogue
    if "--ceiling" in sys.argv:
        for _ in range(2):
            torch.nn.functional.linear(a, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.nn.functional.linear(a, w)
        e1.record(); torch.cuda.synchronize()
        res["lib"] = 2.0 * M * N * K / (e0.elapsed_time(e1) / 10) / 1e9
        print(f"{name:14s} vendor-library bf16 GEMM (no epilogue): {res['lib']:7.1f} TF")
    print(f"{name:14s} M={M} N={N} K={K}: 128-tile {res[0]:7.1f} TF | 256-tile {res[1]:7.1f} TF | heuristic {res[2]:7.1f} TF | 256-tile/mfma32 {res[3]:7.1f} TF " + (f" | 4-wave 128x128 {res[4]:7.1f} TF | 4-wave 1-barrier {res[5]:7.1f} TF" if 4 in res else ""))
ops.lib.icv_set_option(b"gemm256", 2); ops.lib.icv_set_option(b"gemm256_mfma", 16)
