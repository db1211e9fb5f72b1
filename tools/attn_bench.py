"""Micro-benchmark of icv_attention_fwd on the DiT shapes (run on the GPU box)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps

ops = HipOps("cuda:0")
cases = [("1.3b self", 37440, 37440, 12), ("14b self", 37440, 37440, 40), ("sp8 14b", 4680, 37440, 40), ("sp4 14b", 9360, 37440, 40),
         ("14b cross", 37440, 512, 40), ("14b ximg", 37440, 257, 40), ("sp4 cross", 9360, 512, 40)]
if len(sys.argv) > 1:
    cases = [c for c in cases if any(c[0].startswith(a) for a in sys.argv[1:])]
n = int(os.environ.get("ATTN_ITERS", "3"))
if os.environ.get("ATTN_ACC"):
    # accuracy of the kernel vs fp64 attention on a moderately long sequence, per defer-max threshold
    torch.manual_seed(0)
    Sq, Skv, H = 1024, 4096, 2
    q = torch.randn((Sq, H * 128), device="cuda").to(torch.bfloat16)
    k = torch.randn((Skv, H * 128), device="cuda").to(torch.bfloat16)
    v = torch.randn((Skv, H * 128), device="cuda").to(torch.bfloat16)
    qh, kh, vh = (t.double().reshape(-1, H, 128).transpose(0, 1) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(128), -1) @ vh).transpose(0, 1).reshape(Sq, -1)
    for thr in (0, 2, 4, 8):
        ops.lib.icv_set_option(b"attn_defer_max_log2", thr)
        o = torch.empty_like(q)
        ops.attention(q, k, v, o, H, SCALE)
        e = (o.double() - ref)
        print(f"thr={thr}: max|err| {e.abs().max():.3e}  rms err {e.pow(2).mean().sqrt():.3e}  (rms ref {ref.pow(2).mean().sqrt():.3e})")
    ops.lib.icv_set_option(b"attn_defer_max_log2", int(os.environ.get("ATTN_THR", "8")))
variants = [int(x) for x in os.environ.get("ATTN_VARIANTS", "5").split(",")]
if os.environ.get("ATTN_ABLATE"):
    ops.lib.icv_set_option(b"attn7_ablate", int(os.environ["ATTN_ABLATE"]))
rounds = int(os.environ.get("ATTN_ROUNDS", "3"))
SCALE = math.log(2.0) if os.environ.get("ATTN_UNIT") else 128 ** -0.5   # ATTN_UNIT=1: the DiT's unit-scale call (K carries the scale)
for name, Sq, Skv, H in cases:
    d = H * 128
    q = torch.randn((Sq, d), device="cuda").to(torch.bfloat16)
    k = torch.randn((Skv, d), device="cuda").to(torch.bfloat16)
    v = torch.randn((Skv, d), device="cuda").to(torch.bfloat16)
    if os.environ.get("ATTN_UNIT"):
        k = (k.float() * (128 ** -0.5 * math.log2(math.e))).to(torch.bfloat16)   # K carries the softmax scale, like the DiT's
    if os.environ.get("ATTN_ZERO"):
        q.zero_(); k.zero_(); v.zero_()
    o = torch.empty_like(q)
    best = {vv: [] for vv in variants}
    for rd in range(rounds):           # interleaved rounds: within-process A/B
        for vv in variants:
            # variant codes: <100 -> attn.hip variant; 1000+x -> attn2.hip variant x
            ops.lib.icv_set_option(b"attn7_short", 0)
            if vv >= 9000:
                ops.lib.icv_set_option(b"attn_kernel", 9); ops.lib.icv_set_option(b"attn9_variant", vv - 9000)
            elif vv >= 8000:   # attn7 variant x on the short-key launch shape (4-wave blocks, two per CU)
                ops.lib.icv_set_option(b"attn_kernel", 7); ops.lib.icv_set_option(b"attn7_variant", vv - 8000); ops.lib.icv_set_option(b"attn7_short", 1 << 30)
            elif vv >= 7000:
                ops.lib.icv_set_option(b"attn_kernel", 7); ops.lib.icv_set_option(b"attn7_variant", vv - 7000)
            elif vv >= 6000:
                ops.lib.icv_set_option(b"attn_kernel", 6); ops.lib.icv_set_option(b"attn6_variant", vv - 6000)
            elif vv >= 5000:
                ops.lib.icv_set_option(b"attn_kernel", 5)
            elif vv >= 4000:
                ops.lib.icv_set_option(b"attn_kernel", 4); ops.lib.icv_set_option(b"attn4_variant", vv - 4000)
            elif vv >= 3000:
                ops.lib.icv_set_option(b"attn_kernel", 3); ops.lib.icv_set_option(b"attn3_variant", vv - 3000)
            elif vv >= 1000:
                ops.lib.icv_set_option(b"attn_kernel", 7); ops.lib.icv_set_option(b"attn2_variant", vv - 1000)
            else:
                ops.lib.icv_set_option(b"attn_kernel", 1); ops.lib.icv_set_option(b"attn_variant", vv)
            ops.attention(q, k, v, o, H, SCALE)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.attention(q, k, v, o, H, SCALE)
            e1.record(); torch.cuda.synchronize()
            best[vv].append(e0.elapsed_time(e1) / n)
    fl = 4.0 * Sq * Skv * d
    print(f"{name:10s} Sq={Sq} Skv={Skv} H={H}: " + " | ".join(
        f"v{vv}: {fl / sorted(best[vv])[len(best[vv]) // 2] / 1e9:6.1f} TF (min {min(best[vv]):.3f} ms)" for vv in variants))
ops.lib.icv_set_option(b"attn_variant", 5); ops.lib.icv_set_option(b"attn_kernel", 7); ops.lib.icv_set_option(b"attn2_variant", 12)
ops.lib.icv_set_option(b"attn7_short", -1); ops.lib.icv_set_option(b"attn7_variant", -1)

# cost of splitting one self-attention over C key chunks with carried state (sequence-parallel path)
if os.environ.get("ATTN_CHUNKS"):
    for world in (8, 2):
        Sq, Skv, H = 37440 // world, 37440, 40
        d = H * 128
        q = torch.randn((Sq, d), device="cuda").to(torch.bfloat16)
        k = torch.randn((Skv, d), device="cuda").to(torch.bfloat16)
        v = torch.randn((Skv, d), device="cuda").to(torch.bfloat16)
        o = torch.empty_like(q); acc = torch.empty((Sq, d), device="cuda"); ml = torch.empty((Sq, H, 2), device="cuda")
        for C in (1, 2, 4, 8):
            b = [(Skv * c) // C for c in range(C + 1)]
            def run():
                for c in range(C):
                    ops.attention_chunk(q, k[b[c]:b[c + 1]], v[b[c]:b[c + 1]], o, acc, ml, H, 128 ** -0.5, first=(c == 0), last=(c == C - 1))
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print(f"world={world} Sq={Sq}: {C} chunk(s): {ms:7.3f} ms  {4.0 * Sq * Skv * d / ms / 1e9:7.1f} TF")
