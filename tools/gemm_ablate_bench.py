"""Timing ablation of the default GEMM (run on the GPU box): how much of gemm256's time is its K-tile staging?
gemm256_ablate = 1 drops the DMA instructions of the main loop, = 2 keeps them but points every one at K-tile 0 (an
L2-resident source: same instruction stream and LDS writes, no HBM traffic); results are wrong, the MFMA / LDS-read stream is unchanged."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps, EPI_GELU_BF16
ops = HipOps("cuda:0")
M, N, K = 37440, 13824, 5120
a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
w = (torch.randn((N, K), device="cuda") / math.sqrt(K)).to(torch.bfloat16)
bias = torch.randn((N,), device="cuda")
out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
def run(label):
    for _ in range(2): ops.gemm(a, w, bias, out, EPI_GELU_BF16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm(a, w, bias, out, EPI_GELU_BF16)
    e1.record(); torch.cuda.synchronize()
    print(f"{label}: {2.0*M*N*K/(e0.elapsed_time(e1)/10)/1e9:.1f} TF/s")
ops.lib.icv_set_option(b"gemm256", 1)
for rnd in range(3):
    for ab in (0, 1, 2):
        ops.lib.icv_set_option(b"gemm256_ablate", ab); run(f"gemm256 FFN1 14B, ablate={ab}")
ops.lib.icv_set_option(b"gemm256_ablate", 0); ops.lib.icv_set_option(b"gemm256", 2)
