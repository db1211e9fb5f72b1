"""Micro-benchmark: fp8 attention (prepare + forward) vs the bf16 default on the DiT shapes (GPU box)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps

ops = HipOps("cuda:0")


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


fold = 128 ** -0.5 * math.log2(math.e)
for name, Sq, Skv, H in (("1.3b self", 37440, 37440, 12), ("14b self", 37440, 37440, 40), ("sp4 14b", 9360, 37440, 40), ("14b cross", 37440, 512, 40),
                         ("14b 720p self", 86400, 86400, 40)):
    d = H * 128
    q = torch.randn((Sq, d), device="cuda").to(torch.bfloat16)
    k = (torch.randn((Skv, d), device="cuda") * fold).to(torch.bfloat16)
    v = torch.randn((Skv, d), device="cuda").to(torch.bfloat16)
    o = torch.empty_like(q)
    ws = ops.attention_fp8_buffers(Sq, Skv, d, H)
    t16 = timeit(lambda: ops.attention(q, k, v, o, H, math.log(2.0)))
    t8 = timeit(lambda: ops.attention_fp8(q, k, v, o, H, ws))
    qq, kq, vt, amax = ws
    tf = timeit(lambda: ops.lib.icv_attention_fp8_fwd(qq.data_ptr(), qq.stride(0), kq.data_ptr(), kq.stride(0), vt.data_ptr(), amax.data_ptr(),
                                                    o.data_ptr(), o.stride(0), Sq, Skv, H, ops._stream()))
    fl = 4.0 * Sq * Skv * d / 1e9
    alt = ""
    ref_o = None
    for var in [int(x) for x in os.environ.get("ATTN8_VARIANTS", "").split(",") if x]:
        ops.lib.icv_set_option(b"attn8_variant", var)
        tv = timeit(lambda: ops.lib.icv_attention_fp8_fwd(qq.data_ptr(), qq.stride(0), kq.data_ptr(), kq.stride(0), vt.data_ptr(), amax.data_ptr(),
                                                        o.data_ptr(), o.stride(0), Sq, Skv, H, ops._stream()))
        if ref_o is None:
            ops.lib.icv_set_option(b"attn8_variant", 0)
            ops.lib.icv_attention_fp8_fwd(qq.data_ptr(), qq.stride(0), kq.data_ptr(), kq.stride(0), vt.data_ptr(), amax.data_ptr(),
                                          o.data_ptr(), o.stride(0), Sq, Skv, H, ops._stream())
            ref_o = o.clone()
            ops.lib.icv_set_option(b"attn8_variant", var)
            ops.lib.icv_attention_fp8_fwd(qq.data_ptr(), qq.stride(0), kq.data_ptr(), kq.stride(0), vt.data_ptr(), amax.data_ptr(),
                                          o.data_ptr(), o.stride(0), Sq, Skv, H, ops._stream())
        dmax = float((o.float() - ref_o.float()).abs().max())
        alt += f" | variant {var}: {fl / tv:7.1f} TF (max |diff| vs 0: {dmax:.3g})"
    ops.lib.icv_set_option(b"attn8_variant", -1)
    print(f"{name:14s} Sq={Sq} Skv={Skv} H={H}: bf16 {fl / t16:7.1f} TF ({t16:.3f} ms) | fp8 fwd {fl / tf:7.1f} TF ({tf:.3f} ms) | "
          f"prepare {t8 - tf:.3f} ms | fp8 total speed-up {t16 / t8:.2f}x" + alt)
