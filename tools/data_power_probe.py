"""Power-limit evidence: throughput of the two MFMA kernels (and the vendor GEMM) as a function of the operand DATA, same
process, interleaved rounds.  The instruction stream is identical in every column; only the bit toggling in the operand /
accumulator paths changes.  (run on the GPU box; writes nothing — redirect stdout)

    random    N(0,1) bf16 — what bench.py and the tests use, and what real activations / weights look like to the datapath
    smallint  values in {-1, 0, 1}: exponent and sign toggle, mantissa bits are all zero
    zero      all-zero operands
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from infinicube_amd.videogen.ops import EPI_BF16, EPI_GELU_BF16, HipOps  # noqa: E402

ops = HipOps("cuda:0")
S = 37440
ROUNDS = int(os.environ.get("ROUNDS", "3"))


def make(shape, kind, scale=1.0):
    if kind == "zero":
        return torch.zeros(shape, device="cuda", dtype=torch.bfloat16)
    if kind == "smallint":
        return torch.randint(-1, 2, shape, device="cuda").to(torch.bfloat16)
    return (torch.randn(shape, device="cuda") * scale).to(torch.bfloat16)


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


kinds = ("random", "smallint", "zero")
# ---- GEMM: 14B FFN1 shape, this library's kernel (GELU epilogue) and the vendor library's (no epilogue)
M, N, K = S, 13824, 5120
data = {k: (make((M, K), k), make((N, K), k, 1.0 / math.sqrt(K))) for k in kinds}
bias = torch.zeros((N,), device="cuda")
out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
best = {(k, w): 0.0 for k in kinds for w in ("icv", "lib")}
for _ in range(ROUNDS):
    for k in kinds:
        a, w = data[k]
        best[(k, "icv")] = max(best[(k, "icv")], 2.0 * M * N * K / timed(lambda: ops.gemm(a, w, bias, out, EPI_GELU_BF16), 10) / 1e9)
        best[(k, "lib")] = max(best[(k, "lib")], 2.0 * M * N * K / timed(lambda: torch.nn.functional.linear(a, w), 10) / 1e9)
print(f"GEMM 14B FFN1 [{M},{K}] x [{N},{K}]^T, TF/s (best of {ROUNDS} interleaved rounds)")
for k in kinds:
    print(f"  {k:9s} gemm256 + GELU epilogue {best[(k, 'icv')]:7.1f}   vendor library (no epilogue) {best[(k, 'lib')]:7.1f}")
del data, out

# ---- self-attention: 14B shape, the DiT's unit-scale call
H = 40
d = H * 128
SC = math.log(2.0)
res = {k: 0.0 for k in kinds}
qkv = {k: tuple(make((S, d), k, (128 ** -0.5 * math.log2(math.e)) if i == 1 else 1.0) for i in range(3)) for k in kinds}
o = torch.empty((S, d), device="cuda", dtype=torch.bfloat16)
for _ in range(ROUNDS):
    for k in kinds:
        q, kk, v = qkv[k]
        res[k] = max(res[k], 4.0 * S * S * d / timed(lambda: ops.attention(q, kk, v, o, H, SC), 3) / 1e9)
print(f"self-attention S={S}, {H} heads (attn7 default), TF/s")
for k in kinds:
    print(f"  {k:9s} {res[k]:7.1f}")
