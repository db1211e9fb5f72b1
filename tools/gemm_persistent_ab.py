"""A/B of the persistent GEMM kernel (csrc/gemm256p.hip; option gemm256 = 5 static stride / 6 per-XCD work counter, forced for EVERY
epilogue) against the one-tile-per-block gemm256 launch (gemm256 = 1) on the DiT's GEMM shapes at the rows the CFG-batched pair runs
(2S): interleaved rounds in ONE process, median of the rounds, and a bit-for-bit check (same MFMA order per accumulator).  The shipped
default (gemm256 = 2) takes the work-counter kernel for the bf16 / GELU epilogues only.
    python tools/gemm_persistent_ab.py"""
import math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps, EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32

ops = HipOps("cuda:0")
S2 = 2 * 37440
SHAPES = [("14b qkv", S2, 15360, 5120, EPI_BF16), ("14b o", S2, 5120, 5120, EPI_RESID_F32), ("14b xq", S2, 5120, 5120, EPI_BF16),
          ("14b ffn1", S2, 13824, 5120, EPI_GELU_BF16), ("14b ffn2", S2, 5120, 13824, EPI_RESID_F32),
          ("1.3b qkv", S2, 4608, 1536, EPI_BF16), ("1.3b ffn1", S2, 8960, 1536, EPI_GELU_BF16), ("sp4 14b ffn1", 9360, 13824, 5120, EPI_GELU_BF16)]
ROUNDS, REPS = 5, 6
for name, M, N, K, epi in SHAPES:
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    bias, gate = torch.randn((N,), device="cuda"), torch.randn((N,), device="cuda")
    resid = torch.randn((M, N), device="cuda") if epi == EPI_RESID_F32 else None
    out = torch.empty((M, N), device="cuda", dtype=torch.float32 if epi == EPI_RESID_F32 else torch.bfloat16)
    kw = dict(resid=resid, gate=gate) if epi == EPI_RESID_F32 else {}
    res = {}
    for mode in (1, 5, 6):
        ops.lib.icv_set_option(b"gemm256", mode)
        out.zero_(); ops.gemm(a, w, bias, out, epi, **kw); torch.cuda.synchronize()
        res[mode] = out.clone()
    same = torch.equal(res[1], res[5]) and torch.equal(res[1], res[6])
    times = {1: [], 5: [], 6: []}
    for _ in range(ROUNDS):
        for mode in (1, 5, 6):
            ops.lib.icv_set_option(b"gemm256", mode)
            ops.gemm(a, w, bias, out, epi, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                ops.gemm(a, w, bias, out, epi, **kw)
            e1.record(); torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) / REPS)
    fl = 2.0 * M * N * K
    t1, t5, t6 = statistics.median(times[1]), statistics.median(times[5]), statistics.median(times[6])
    print(f"{name:13s} M={M} N={N} K={K} epi={epi}: one tile per block {fl / t1 / 1e9:7.1f} TF | persistent, static stride {fl / t5 / 1e9:7.1f} TF ({100 * (t1 / t5 - 1):+.1f} %)"
          f" | persistent, per-XCD work counter {fl / t6 / 1e9:7.1f} TF ({100 * (t1 / t6 - 1):+.1f} %)"
          f"{'' if same else '  RESULT MISMATCH'}", flush=True)
ops.lib.icv_set_option(b"gemm256", 2)
