"""A/B of the gemm256 schedule variants (icv_set_option "gemm256_sched": bit 0 = two 32-MFMA phases per K-tile,
bit 1 = batched residual loads in the RESID epilogue) on the DiT's GEMM shapes; interleaved rounds in ONE process,
median of the rounds, and a bit-for-bit check of every variant against variant 0 (same MFMA order per accumulator)."""
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infinicube_amd.videogen.ops import HipOps, EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32

ops = HipOps("cuda:0")
S = 37440
SHAPES = [("14b qkv", S, 15360, 5120, EPI_BF16), ("14b o", S, 5120, 5120, EPI_RESID_F32), ("14b xq", S, 5120, 5120, EPI_BF16),
          ("14b ffn1", S, 13824, 5120, EPI_GELU_BF16), ("14b ffn2", S, 5120, 13824, EPI_RESID_F32),
          ("1.3b qkv", S, 4608, 1536, EPI_BF16), ("1.3b o", S, 1536, 1536, EPI_RESID_F32), ("1.3b ffn1", S, 8960, 1536, EPI_GELU_BF16),
          ("1.3b ffn2", S, 1536, 8960, EPI_RESID_F32), ("sp4 14b ffn1", 9360, 13824, 5120, EPI_GELU_BF16), ("sp4 14b o", 9360, 5120, 5120, EPI_RESID_F32)]
VARIANTS = [int(x) for x in os.environ.get("SCHEDS", "0,1,2,3").split(",")]
ROUNDS, REPS = 5, 6
ops.lib.icv_set_option(b"gemm256", 1)
for name, M, N, K, epi in SHAPES:
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    bias, gate = torch.randn((N,), device="cuda"), torch.randn((N,), device="cuda")
    resid = torch.randn((M, N), device="cuda") if epi == EPI_RESID_F32 else None
    out = torch.empty((M, N), device="cuda", dtype=torch.float32 if epi == EPI_RESID_F32 else torch.bfloat16)
    kw = dict(resid=resid, gate=gate) if epi == EPI_RESID_F32 else {}
    ref = None
    ok = {}
    for v in VARIANTS:
        ops.lib.icv_set_option(b"gemm256_sched", v)
        same = True
        for _ in range(2):      # twice: a race shows as a run-to-run difference as well
            out.zero_()
            ops.gemm(a, w, bias, out, epi, **kw)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            same = same and torch.equal(out, ref)
        ok[v] = same
    times = {v: [] for v in VARIANTS}
    for _ in range(ROUNDS):
        for v in VARIANTS:
            ops.lib.icv_set_option(b"gemm256_sched", v)
            ops.gemm(a, w, bias, out, epi, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                ops.gemm(a, w, bias, out, epi, **kw)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / REPS)
    fl = 2.0 * M * N * K
    print(f"{name:13s} M={M} N={N} K={K} epi={epi}: " + " | ".join(
        f"sched {v}: {fl / statistics.median(times[v]) / 1e9:7.1f} TF (best {fl / min(times[v]) / 1e9:7.1f}){'' if ok[v] else ' MISMATCH'}" for v in VARIANTS), flush=True)
ops.lib.icv_set_option(b"gemm256_sched", 3); ops.lib.icv_set_option(b"gemm256", 2)
