"""Micro-benchmark: icv_gemm_fp8 vs icv_gemm_bf16 on the DiT shapes, plus the quantise passes (GPU box)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps, EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32, FP8

ops = HipOps("cuda:0")
S = int(os.environ.get("S", 37440))


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [("1.3b qkv", S, 4608, 1536, EPI_BF16), ("1.3b ffn1", S, 8960, 1536, EPI_GELU_BF16), ("1.3b ffn2", S, 1536, 8960, EPI_RESID_F32),
          ("14b qkv", S, 15360, 5120, EPI_BF16), ("14b o", S, 5120, 5120, EPI_RESID_F32),
          ("14b ffn1", S, 13824, 5120, EPI_GELU_BF16), ("14b ffn2", S, 5120, 13824, EPI_RESID_F32),
          ("sp8 14b ffn1", 4680, 13824, 5120, EPI_GELU_BF16), ("sp8 14b ffn2", 4680, 5120, 13824, EPI_RESID_F32)]
for name, M, N, K, epi in shapes:
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn((N,), device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.float32 if epi == EPI_RESID_F32 else torch.bfloat16)
    kw = dict(resid=out, gate=bias) if epi == EPI_RESID_F32 else {}
    a8, asc = torch.empty((M, K), dtype=FP8, device="cuda"), torch.empty((M,), device="cuda")
    w8, wsc = torch.empty((N, K), dtype=FP8, device="cuda"), torch.empty((N,), device="cuda")
    ops.quantize_rows(a, a8, asc); ops.quantize_rows(w, w8, wsc)
    t16 = timeit(lambda: ops.gemm(a, w, bias, out, epi, **kw))
    ops.lib.icv_set_option(b"gemm_fp8_sched", 0)
    t8_r1 = timeit(lambda: ops.gemm_fp8(a8, asc, w8, wsc, bias, out, epi, **kw))
    ops.lib.icv_set_option(b"gemm_fp8_sched", 3)
    t8 = timeit(lambda: ops.gemm_fp8(a8, asc, w8, wsc, bias, out, epi, **kw))
    tq = timeit(lambda: ops.quantize_rows(a, a8, asc))
    fl = 2.0 * M * N * K / 1e9
    print(f"{name:13s} M={M} N={N} K={K}: bf16 {fl / t16:7.1f} TF ({t16:.3f} ms) | fp8 {fl / t8:7.1f} TF ({t8:.3f} ms; round-1 schedule {fl / t8_r1:7.1f} TF) | "
          f"quantise A {tq:.3f} ms = {M * K * 3 / tq / 1e6:.0f} GB/s | fp8+quant speed-up {t16 / (t8 + tq):.2f}x")
x = torch.randn((S, 5120), device="cuda")
h, h8, hs = torch.empty((S, 5120), dtype=torch.bfloat16, device="cuda"), torch.empty((S, 5120), dtype=FP8, device="cuda"), torch.empty((S,), device="cuda")
sh, sc = torch.randn(5120, device="cuda"), torch.randn(5120, device="cuda")
t0 = timeit(lambda: ops.ln_modulate(x, h, shift=sh, scale=sc))
t1 = timeit(lambda: ops.ln_modulate_fp8(x, h8, hs, shift=sh, scale=sc))
print(f"ln_modulate d=5120: bf16 out {t0:.3f} ms ({S * 5120 * 6 / t0 / 1e6:.0f} GB/s) | fp8 out {t1:.3f} ms ({S * 5120 * 5 / t1 / 1e6:.0f} GB/s)")
