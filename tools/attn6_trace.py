"""Phase timeline of the ping-pong attention kernel (attn6 variant 13): s_memtime stamps of block 0, waves 0 / 4."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from infinicube_amd.videogen.ops import HipOps

ops = HipOps("cuda:0")
Sq, Skv, H = 37440, 37440, 12
d = H * 128
q, k, v = (torch.randn((n, d), device="cuda").to(torch.bfloat16) for n in (Sq, Skv, Skv))
o = torch.empty_like(q)
VAR = int(os.environ.get("A6VAR", 13))
print("variant", VAR, "(13 = full; +16 no softmax, +32 no staging, +64 no LDS fragment reads)")
ops.lib.icv_set_option(b"attn_kernel", 6); ops.lib.icv_set_option(b"attn6_variant", VAR)
for _ in range(2):
    ops.attention(q, k, v, o, H, 128 ** -0.5)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 2048)()
fn = ops.lib.icv_debug_attn6_trace
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
assert fn(ctypes.cast(buf, ctypes.c_void_p)) == 0
t = np.array(buf, dtype=np.int64).reshape(2, 1024)
for g in range(2):
    x = t[g][: 7 * 120].reshape(-1, 7)[20:110]          # 7 stamps per tile; skip warm-up tiles
    names = ["BURST issue", "barrier", "wait vmcnt(0) (prefetched K/V)", "4 ds_write + lgkmcnt(0)", "4 global_load issue (+addr VALU)", "softmax", "barrier (to next burst)"]
    d = [x[:, i + 1] - x[:, i] for i in range(6)] + [x[1:, 0] - x[:-1, 6]]
    per = x[1:, 0] - x[:-1, 0]
    print(f"group {g}: tile period {per.mean():.0f} ticks | " + " | ".join(f"{n} {v.mean():.0f}" for n, v in zip(names, d)))

