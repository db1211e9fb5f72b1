"""Micro-benchmark of the HBM-bound kernels (run on the GPU box): GB/s against the ~6.3 TB/s achievable."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps, RopeTable

ops = HipOps("cuda:0")
S = 37440

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for d in (1536, 5120):
    # rotate over several buffers so the 256 MB infinity cache cannot serve the reads
    nb = 4
    xs = [torch.randn((S, d), device="cuda") for _ in range(nb)]
    outs = [torch.empty((S, d), dtype=torch.bfloat16, device="cuda") for _ in range(nb)]
    sh, sc = torch.randn(d, device="cuda"), torch.randn(d, device="cuda")
    i = [0]
    def ln():
        j = i[0] % nb; i[0] += 1
        ops.ln_modulate(xs[j], outs[j], shift=sh, scale=sc)
    def cast():
        j = i[0] % nb; i[0] += 1
        ops.cast_bf16(xs[j], outs[j])
    bytes_ln = S * d * 6
    for wpr in (1, 2, 4, 0):
        ops.lib.icv_set_option(b"ln_waves_per_row", wpr)
        try:
            ms = timeit(ln)
            print(f"d={d:5d} ln_modulate W={wpr or 'auto':4} {ms * 1e3:8.1f} us  {bytes_ln / ms / 1e9:7.2f} TB/s")
        except Exception as e:
            print(f"d={d} W={wpr}: {str(e)[:80]}")
    ms = timeit(cast)
    print(f"d={d:5d} cast f32->bf16      {ms * 1e3:8.1f} us  {bytes_ln / ms / 1e9:7.2f} TB/s")
    qk = [torch.randn((3, S, d), device="cuda").to(torch.bfloat16) for _ in range(nb)]
    w = torch.ones(d, device="cuda")
    rope = RopeTable.build(24, 60, 104, "cuda:0")
    def rms():
        j = i[0] % nb; i[0] += 1
        ops.rmsnorm_rope(qk[j][0], w, qk[j][1], w, rope=rope)
    ms = timeit(rms)
    print(f"d={d:5d} {'rmsnorm_rope q,k':16s} {ms * 1e3:8.1f} us  {S * d * 8 / ms / 1e9:7.2f} TB/s")
