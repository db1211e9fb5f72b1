"""Randomised differential test of the two MFMA kernels' C entry points (run on the GPU box).

Shapes are drawn at random inside each entry point's documented contract (ragged M / N / Sq / Skv, strided operands,
split-plane outputs, every epilogue, both tile families, carried-state key chunks) and each result is compared with a
stock PyTorch fp32 computation of the same op on the GPU (test infrastructure; independent of libicvideo).  Every launch
is repeated and must be bit-identical (race screen for the counted-vmcnt rings).

    python tools/fuzz_kernels.py [seconds] [seed] [cases]   -> prints one line per failure and a summary; exit code 1 on any failure
                                                              (cases > 0: stop after exactly that many cases instead of after
                                                              `seconds` - the sequence is then a pure function of the seed)
"""
import math
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from infinicube_amd.videogen.ops import EPI_BF16, EPI_F32, EPI_GELU_BF16, EPI_RESID_F32, FP8, HipOps  # noqa: E402

DEV = "cuda:0"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = random.Random(seed)
ops = HipOps(DEV)
fails = []


def close_bf16(got, ref, what, abs_floor=2.0 ** -8, rms_bound=None):
    """the test suite's per-op bar (tests/test_kernels_gpu.py assert_bf16_close): |d| <= 2^-7 |ref| + abs_floor * rms(ref),
    and optionally rms(d) <= rms_bound * rms(ref)"""
    got, ref = got.float(), ref.float()
    rms = float(ref.pow(2).mean().sqrt())
    err = (got - ref).abs()
    bad = err > 2.0 ** -7 * ref.abs() + abs_floor * rms
    rms_err = float((got - ref).pow(2).mean().sqrt())
    if not torch.isfinite(got).all() or bad.any() or (rms_bound is not None and rms_err > rms_bound * rms):
        fails.append(f"{what}: {int(bad.sum())} of {bad.numel()} outside tolerance, max err {float(err.max()):.4g}, rms err {rms_err:.3g} vs rms {rms:.3g}")
        return False
    return True


def close_f32(got, ref, what, rtol=3e-3):
    err = float((got.float() - ref).abs().max())
    scale = float(ref.abs().max()) + 1e-6
    if not torch.isfinite(got).all() or err > rtol * scale:
        fails.append(f"{what}: max err {err:.4g} vs scale {scale:.4g}")
        return False
    return True


def fuzz_gemm():
    M = rng.choice([1, 7, 63, 64, 65, 127, 129, 255, 256, 257, 300, 511, 513, 777, 1000, 1500, 2049]) if rng.random() < 0.7 else rng.randint(1, 3000)
    N = 4 * rng.choice([1, 3, 16, 31, 32, 33, 64, 65, 96, 128, 192, 256, 320, 384, 513]) if rng.random() < 0.7 else 4 * rng.randint(1, 700)
    K = 64 * rng.choice([1, 2, 3, 4, 8, 16, 24, 32, 41])
    epi = rng.choice([EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32, EPI_F32])
    tile = rng.choice([0, 1, 2])         # 128-tile kernel, 256-tile kernel, heuristic
    mfma = rng.choice([16, 16, 32])
    sched = rng.choice([0, 1, 2, 3, 3, 7, 11, 19, 35, 51, 67, 67])   # 19 / 35 / 51: non-temporal DMA of the A / W / both streams (A/B only); 67: k-split units (round 4)
    pad = rng.choice([0, 0, 64])         # strided A
    ops.lib.icv_set_option(b"gemm256", tile); ops.lib.icv_set_option(b"gemm256_mfma", mfma); ops.lib.icv_set_option(b"gemm256_sched", sched)
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 2 ** 31))
    a_full = torch.randn((M, K + pad), device=DEV, generator=g).to(torch.bfloat16)
    a = a_full[:, :K]
    w = (torch.randn((N, K), device=DEV, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn((N,), device=DEV, generator=g) * 0.1 if rng.random() < 0.8 else None
    acc = a.float() @ w.float().t() + (bias if bias is not None else 0.0)
    what = f"gemm M={M} N={N} K={K} epi={epi} tile={tile} mfma={mfma} sched={sched} pad={pad} bias={bias is not None}"
    if epi in (EPI_BF16, EPI_GELU_BF16):
        ref = F.gelu(acc, approximate="tanh") if epi == EPI_GELU_BF16 else acc
        nsplit = None
        if epi == EPI_BF16 and N % 12 == 0 and rng.random() < 0.4:
            nsplit = N // 3
        outs = []
        for _ in range(2):
            out = torch.full((M, N) if nsplit is None else (3, M, nsplit), 7.0, dtype=torch.bfloat16, device=DEV)
            ops.gemm(a, w, bias, out, epi, **({} if nsplit is None else {"nsplit": nsplit}))
            outs.append(out)
        if nsplit is not None:
            ref = ref.reshape(M, 3, nsplit).permute(1, 0, 2)
            what += f" nsplit={nsplit}"
        ok = close_bf16(outs[0], ref, what)
    elif epi == EPI_RESID_F32:
        resid = torch.randn((M, N), device=DEV, generator=g)
        gate = torch.randn((N,), device=DEV, generator=g) if rng.random() < 0.7 else None
        outs = []
        for _ in range(2):
            x = resid.clone()
            ops.gemm(a, w, bias, x, epi, resid=x, gate=gate)
            outs.append(x)
        ok = close_f32(outs[0], resid + (gate if gate is not None else 1.0) * acc, what + f" gate={gate is not None}")
    else:
        outs = []
        for _ in range(2):
            out = torch.empty((M, N), device=DEV)
            ops.gemm(a, w, bias, out, epi)
            outs.append(out)
        ok = close_f32(outs[0], acc, what)
    if ok and not torch.equal(outs[0], outs[1]):
        fails.append(what + ": two identical launches differ (race?)")


def fuzz_gemm_fp8():
    """e4m3 GEMM: operands quantised by the library's own row quantiser, reference = the dequantised product in fp32"""
    M = rng.choice([1, 63, 64, 257, 300, 777, 1030, 2049]) if rng.random() < 0.7 else rng.randint(1, 2500)
    N = 4 * rng.choice([1, 16, 33, 64, 65, 128, 192, 260, 384]) if rng.random() < 0.7 else 4 * rng.randint(1, 500)
    K = 128 * rng.choice([1, 2, 3, 4, 12, 20])
    epi = rng.choice([EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32, EPI_F32])
    ops.lib.icv_set_option(b"gemm_fp8_sched", rng.choice([3, 3, 0]))
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 2 ** 31))
    a = torch.randn((M, K), device=DEV, generator=g) * 1.5
    if rng.random() < 0.5:
        a[:, 3 % K] *= 25.0                # an outlier channel: the per-row scale is set by it
    w = torch.randn((N, K), device=DEV, generator=g) / math.sqrt(K)
    a8, asc = torch.empty((M, K), dtype=FP8, device=DEV), torch.empty((M,), device=DEV)
    w8, wsc = torch.empty((N, K), dtype=FP8, device=DEV), torch.empty((N,), device=DEV)
    ops.quantize_rows(a, a8, asc); ops.quantize_rows(w, w8, wsc)
    bias = torch.randn((N,), device=DEV, generator=g) * 0.1 if rng.random() < 0.8 else None
    acc = (a8.float() * asc[:, None]).double() @ (w8.float() * wsc[:, None]).double().t()
    acc = acc.float() + (bias if bias is not None else 0.0)
    what = f"gemm_fp8 M={M} N={N} K={K} epi={epi} bias={bias is not None}"
    outs = []
    if epi in (EPI_BF16, EPI_GELU_BF16):
        ref = F.gelu(acc, approximate="tanh") if epi == EPI_GELU_BF16 else acc
        for _ in range(2):
            out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
            ops.gemm_fp8(a8, asc, w8, wsc, bias, out, epi)
            outs.append(out)
        ok = close_bf16(outs[0], ref, what)
    elif epi == EPI_RESID_F32:
        resid = torch.randn((M, N), device=DEV, generator=g)
        gate = torch.randn((N,), device=DEV, generator=g) if rng.random() < 0.7 else None
        for _ in range(2):
            x = resid.clone()
            ops.gemm_fp8(a8, asc, w8, wsc, bias, x, epi, resid=x, gate=gate)
            outs.append(x)
        ok = close_f32(outs[0], resid + (gate if gate is not None else 1.0) * acc, what + f" gate={gate is not None}", rtol=1e-3)
    else:
        for _ in range(2):
            out = torch.empty((M, N), device=DEV)
            ops.gemm_fp8(a8, asc, w8, wsc, bias, out, epi)
            outs.append(out)
        ok = close_f32(outs[0], acc, what, rtol=1e-3)
    if ok and not torch.equal(outs[0], outs[1]):
        fails.append(what + ": two identical launches differ (race?)")


def fp8_noise_bar(n_rows):
    """rms error bar of an e4m3 attention output against exact attention, as a fraction of the output's rms: 6 % for many rows; the
    estimate from few (row, head) pairs is itself noisy (seed 5 of round 5: 3 of 128 k cases read 6.05-6.12 % at 66-257 rows), so
    the bar widens like 1 / sqrt(rows) below ~ 1000 of them."""
    return 0.15 if n_rows < 64 else 0.06 * (1.0 + 4.0 / n_rows ** 0.5)


def fuzz_attention_fp8():
    """e4m3 attention against exact softmax attention on the SAME bf16 inputs: the bar is the e4m3 noise floor (rms error
    <= 6 % of the output rms: the GPU test suite holds it to 3 % against an oracle with the same quantisation)"""
    H = rng.choice([1, 2, 3])
    Sq = rng.choice([1, 33, 256, 257, 300, 513]) if rng.random() < 0.7 else rng.randint(1, 1200)
    Skv = rng.choice([64, 65, 128, 257, 640, 1100, 3000]) if rng.random() < 0.7 else rng.randint(1, 3000)
    d = H * 128
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 2 ** 31))
    q = torch.randn((Sq, d), device=DEV, generator=g).to(torch.bfloat16)
    k = (torch.randn((Skv, d), device=DEV, generator=g) * (128 ** -0.5 * math.log2(math.e))).to(torch.bfloat16)
    v = torch.randn((Skv, d), device=DEV, generator=g).to(torch.bfloat16)
    ref = ref_attention(q, k, v, H, math.log(2.0))
    ws = ops.attention_fp8_buffers(Sq, Skv, d, H)
    outs = []
    for _ in range(2):
        o = torch.zeros((Sq, d), dtype=torch.bfloat16, device=DEV)
        ops.attention_fp8(q, k, v, o, H, ws)
        outs.append(o)
    what = f"attention_fp8 Sq={Sq} Skv={Skv} H={H}"
    rms = float(ref.pow(2).mean().sqrt())
    err = float((outs[0].float() - ref).pow(2).mean().sqrt())
    # few rows = a noisy estimate of the relative error (one row of uniform attention over many keys has a tiny output)
    bar = fp8_noise_bar(Sq * H)
    if not torch.isfinite(outs[0].float()).all() or err > bar * rms:
        fails.append(f"{what}: rms err {err:.3g} vs rms {rms:.3g}")
    elif not torch.equal(outs[0], outs[1]):
        fails.append(what + ": two identical launches differ (race?)")


def ref_attention(q, k, v, H, scale):
    Sq, Skv = q.shape[0], k.shape[0]
    qh, kh, vh = (t.float().reshape(-1, H, 128).transpose(0, 1) for t in (q, k, v))
    return (torch.softmax(qh @ kh.transpose(1, 2) * scale, -1) @ vh).transpose(0, 1).reshape(Sq, H * 128)


def fuzz_attention():
    H = rng.choice([1, 2, 3, 5])
    Sq = rng.choice([1, 31, 32, 33, 255, 256, 257, 300, 511, 513, 1000]) if rng.random() < 0.7 else rng.randint(1, 1500)
    Skv = rng.choice([1, 63, 64, 65, 127, 128, 129, 257, 512, 640, 1100, 3000]) if rng.random() < 0.7 else rng.randint(1, 4000)
    d = H * 128
    unit = rng.random() < 0.5            # the DiT's call: the softmax scale folded into K, scale argument = ln 2
    var = rng.choice([-1, -1, 132, 0, 1, 4, 8, 128])   # -1 = the shipped default (variant 132 as one piece of attn7p.hip), 132 = the same schedule in attn7.hip
    ops.lib.icv_set_option(b"attn_kernel", 7); ops.lib.icv_set_option(b"attn7_variant", var)
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 2 ** 31))
    amp = rng.choice([1.0, 1.0, 3.0])    # larger scores stress the lazy-max rescale path
    q = (torch.randn((Sq, d), device=DEV, generator=g) * amp).to(torch.bfloat16)
    k = torch.randn((Skv, d), device=DEV, generator=g).to(torch.bfloat16)
    v = torch.randn((Skv, d), device=DEV, generator=g).to(torch.bfloat16)
    if rng.random() < 0.5:               # a few keys aligned with a query: a late, large maximum
        k[Skv - 1] = q[min(3, Sq - 1)]
        k[Skv // 2] = q[min(40, Sq - 1)]
    scale = 128 ** -0.5
    if unit:
        k = (k.float() * (scale * math.log2(math.e))).to(torch.bfloat16)
        ref = ref_attention(q, k, v, H, math.log(2.0))
        call_scale = math.log(2.0)
    else:
        ref = ref_attention(q, k, v, H, scale)
        call_scale = scale
    what = f"attention Sq={Sq} Skv={Skv} H={H} unit={unit} variant={var} amp={amp}"
    chunks = rng.choice([1, 1, 2, 3]) if Skv >= 8 else 1
    outs = []
    for _ in range(2):
        o = torch.zeros((Sq, d), dtype=torch.bfloat16, device=DEV)
        if chunks == 1:
            ops.attention(q, k, v, o, H, call_scale)
        else:                              # the sequence-parallel form: key chunks with the carried fp32 softmax state
            acc = torch.empty((Sq, d), device=DEV)
            ml = torch.empty((Sq, H, 2), device=DEV)
            cuts = sorted(rng.sample(range(1, Skv), chunks - 1)) if _ == 0 else cuts
            b = [0] + cuts + [Skv]
            for c in range(chunks):
                ops.attention_chunk(q, k[b[c]:b[c + 1]], v[b[c]:b[c + 1]], o, acc, ml, H, call_scale, first=(c == 0), last=(c == chunks - 1))
        outs.append(o)
    ok = close_bf16(outs[0], ref, what + f" chunks={chunks}", abs_floor=2.0 ** -5, rms_bound=2.0 ** -7)
    if ok and not torch.equal(outs[0], outs[1]):
        fails.append(what + ": two identical launches differ (race?)")


def fuzz_conv():
    """icv_conv3d_ndhwc through the padded-volume executor: random geometry / channels / tap set / fused residual against an explicit
    fp32 tap sum on the GPU (slices of the padded volume times the tap's weight matrix: no MIOpen anywhere)."""
    import torch.nn as nn
    from infinicube_amd.videogen import vae as V, vae_hip as VH
    global _hip
    if "_hip" not in globals():
        _hip = VH.VaeHip(nn.Identity(), DEV)
    kind, taps = rng.choice([("333", VH.TAPS_333), ("311", VH.TAPS_311), ("133", VH.TAPS_133), ("111", VH.TAPS_111), ("133d", VH.TAPS_133_DOWN)])
    cin = 32 * rng.choice([1, 2, 3, 4, 6, 12])
    cout = rng.choice([3, 4, 12, 16, 32, 33, 64, 96, 100, 128, 192, 200, 384, 768])
    T, H, W = rng.randint(1, 7), rng.randint(1, 40), rng.randint(1, 50)
    shape = {"333": (3, 3, 3), "311": (3, 1, 1), "133": (1, 3, 3), "133d": (1, 3, 3), "111": (1, 1, 1)}[kind]
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 2 ** 31))
    conv = nn.Conv3d(cin, cout, shape).to(DEV)
    with torch.no_grad():
        conv.weight.copy_((torch.randn(conv.weight.shape, device=DEV, generator=g) * (2.0 / (cin * len(taps)) ** 0.5)).to(torch.bfloat16).float())
        conv.bias.copy_(torch.randn((cout,), device=DEV, generator=g) * 0.1)
    x = (torch.randn((1, cin, T, H, W), device=DEV, generator=g) * 0.7).to(torch.bfloat16)
    xv = _hip._to_vol(x, cin)
    resid = None
    if rng.random() < 0.4:
        r = torch.randn((1, (cout + 3) // 4 * 4, T, H, W), device=DEV, generator=g).to(torch.bfloat16)
        resid = _hip._to_vol(r, (cout + 3) // 4 * 4)
    outs = [_hip.conv(xv, conv, taps, resid=resid).interior()[..., :cout].clone() for _ in range(2)]
    xp = F.pad(x[0].permute(1, 2, 3, 0).float(), (0, 0, 2, 2, 2, 2, 2, 2))              # [T+4, H+4, W+4, C]: room for every tap
    w = conv.weight.detach().float().reshape(cout, cin, -1)
    ref = conv.bias.detach().float().expand(T, H, W, cout).clone()
    for i, (dt, dh, dw) in enumerate(taps):
        ref += xp[2 + dt: 2 + dt + T, 2 + dh: 2 + dh + H, 2 + dw: 2 + dw + W] @ w[:, :, i].t()
    if resid is not None:
        ref += r[0, :cout].permute(1, 2, 3, 0).float()
    what = f"conv {kind} {cin}->{cout} on {T}x{H}x{W} resid={resid is not None}"
    if close_bf16(outs[0], ref, what) and not torch.equal(outs[0], outs[1]):
        fails.append(what + ": two identical launches differ (race?)")


def fuzz_attention_fp8_pieces():
    """the e4m3 wire format: W pieces of m keys quantised piece by piece with the global scales must reproduce the unsharded
    e4m3 launch BIT FOR BIT when m is a multiple of 64, and stay inside the e4m3 noise bar against exact attention otherwise."""
    H, W = rng.choice([1, 2, 3]), rng.choice([1, 2, 3, 4, 8])
    m = 64 * rng.randint(1, 12) if rng.random() < 0.5 else rng.randint(1, 700)
    Sq = rng.choice([1, 33, 256, 257, 300]) if rng.random() < 0.7 else rng.randint(1, 900)
    d, Skv = H * 128, m * W
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 2 ** 31))
    q = torch.randn((Sq, d), device=DEV, generator=g).to(torch.bfloat16)
    k = (torch.randn((Skv, d), device=DEV, generator=g) * (128 ** -0.5 * math.log2(math.e))).to(torch.bfloat16)
    v = torch.randn((Skv, d), device=DEV, generator=g).to(torch.bfloat16)
    ws = ops.attention_fp8_buffers(Sq, Skv, d, H)
    o_ref = torch.zeros((Sq, d), dtype=torch.bfloat16, device=DEV)
    ops.attention_fp8(q, k, v, o_ref, H, ws)
    amax = torch.zeros((3, H), device=DEV)
    ops.attention_fp8_kv_amax(k, v, H, amax)
    bb = ops.attention_fp8_blob_bytes(m, H)
    blobs = torch.empty((W * bb,), dtype=torch.uint8, device=DEV)
    for i in range(W):
        ops.attention_fp8_quantize_kv(k[i * m:(i + 1) * m], v[i * m:(i + 1) * m], H, amax, blobs[i * bb:(i + 1) * bb])
    ws2 = ops.attention_fp8_with_amax(ws, amax)
    ops.attention_fp8_prepare(ws2, H, q=q)
    o = torch.zeros((Sq, d), dtype=torch.bfloat16, device=DEV)
    ops.attention_fp8_pieces(ws2, amax, blobs, m, W, Sq, o, None, None, H, first=True, last=True)
    what = f"attention_fp8_pieces Sq={Sq} m={m} W={W} H={H}"
    if m % 64 == 0:
        if not torch.equal(o, o_ref):
            fails.append(what + ": tile-aligned pieces differ from the unsharded launch")
        return
    ref = ref_attention(q, k, v, H, math.log(2.0))
    rms = float(ref.pow(2).mean().sqrt())
    err = float((o.float() - ref).pow(2).mean().sqrt())
    if not torch.isfinite(o.float()).all() or err > fp8_noise_bar(Sq * H) * rms:
        fails.append(f"{what}: rms err {err:.3g} vs rms {rms:.3g}")


def fuzz_attention_fp8_pieces_gated():
    """round 6, the arrival gate of the e4m3 chunk launch (icv_attention_fp8_fwd_pieces_gated): a random walk order with the 'own' blob read from
    another tensor (its slot holds NaN bytes), a random subset of the other blobs delivered by a side stream AFTER the launch (copy, then
    flag): must equal the all-present launch in the same order bit for bit, stay inside the e4m3 noise bar, and nobody may time out."""
    H, W = rng.choice([1, 2, 3]), rng.choice([1, 2, 3, 4, 8])
    m = 64 * rng.randint(1, 12) if rng.random() < 0.5 else rng.randint(1, 700)
    Sq = rng.choice([1, 33, 256, 257, 300]) if rng.random() < 0.7 else rng.randint(1, 900)
    d, Skv = H * 128, m * W
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 2 ** 31))
    q = torch.randn((Sq, d), device=DEV, generator=g).to(torch.bfloat16)
    k = (torch.randn((Skv, d), device=DEV, generator=g) * (128 ** -0.5 * math.log2(math.e))).to(torch.bfloat16)
    v = torch.randn((Skv, d), device=DEV, generator=g).to(torch.bfloat16)
    ws = ops.attention_fp8_buffers(Sq, Skv, d, H)
    amax = torch.zeros((3, H), device=DEV)
    ops.attention_fp8_kv_amax(k, v, H, amax)
    bb = ops.attention_fp8_blob_bytes(m, H)
    blobs = torch.empty((W * bb,), dtype=torch.uint8, device=DEV)
    for i in range(W):
        ops.attention_fp8_quantize_kv(k[i * m:(i + 1) * m], v[i * m:(i + 1) * m], H, amax, blobs[i * bb:(i + 1) * bb])
    ws2 = ops.attention_fp8_with_amax(ws, amax)
    ops.attention_fp8_prepare(ws2, H, q=q)
    own = rng.randrange(W)
    order = [own] + rng.sample([i for i in range(W) if i != own], W - 1)
    own_blob = blobs[own * bb:(own + 1) * bb].clone()
    staged = blobs.clone()
    blobs[own * bb:(own + 1) * bb] = 0x7F
    want = torch.zeros((Sq, d), dtype=torch.bfloat16, device=DEV)
    ops.attention_fp8_pieces(ws2, amax, blobs, m, W, Sq, want, None, None, H, first=True, last=True,
                             gate=dict(seq=[(i, -1, 0) for i in order], own=(own_blob, own)))
    late = [i for i in order[1:] if rng.random() < 0.5]
    flags = torch.zeros((W + 1,), dtype=torch.int32, device=DEV)
    err = torch.zeros((1,), dtype=torch.int32, device=DEV)
    for i in late:
        blobs[i * bb:(i + 1) * bb] = 0x7F
    torch.cuda.synchronize()
    o = torch.zeros_like(want)
    ops.attention_fp8_pieces(ws2, amax, blobs, m, W, Sq, o, None, None, H, first=True, last=True,
                             gate=dict(seq=[(i, i if i in late else -1, 5 if i in late else 0) for i in order], flags=flags, own=(own_blob, own), err=err,
                                       timeout_us=5_000_000))
    with torch.cuda.stream(SIDE):
        for i in rng.sample(late, len(late)):           # any delivery order
            blobs[i * bb:(i + 1) * bb].copy_(staged[i * bb:(i + 1) * bb])
            ops.flag_write(flags, i, 5, delay_us=rng.choice([0, 0, 200]))
    torch.cuda.synchronize()
    what = f"attention_fp8_pieces_gated Sq={Sq} m={m} W={W} H={H} order={order} late={late}"
    if int(err.item()) != 0:
        fails.append(what + f": a wave gave up waiting (err {int(err.item()) & 0xffffffff:#x})")
    if not torch.equal(o, want):
        fails.append(what + ": late blobs change the result")
    ref = ref_attention(q, k, v, H, math.log(2.0))
    rms = float(ref.pow(2).mean().sqrt())
    e = float((o.float() - ref).pow(2).mean().sqrt())
    if not torch.isfinite(o.float()).all() or e > fp8_noise_bar(Sq * H) * rms:
        fails.append(f"{what}: rms err {e:.3g} vs rms {rms:.3g}")


def fuzz_attention_pieces():
    """round 6's arrival-driven launch (icv_attention_fwd_pieces): random piece lists (ragged, one-row, empty, out of memory order) over K|V
    rows held as ONE [S, 2d] matrix; some pieces gated on flags that a side stream raises AFTER the launch (rows copied in late).  Tile-aligned
    pieces in memory order must equal the plain launch bit for bit; everything else must meet the attention bar against exact attention
    and be deterministic; nobody may time out."""
    H = rng.choice([1, 2, 3, 5])
    Sq = rng.choice([1, 33, 256, 257, 300, 513]) if rng.random() < 0.7 else rng.randint(1, 1200)
    P = rng.randint(1, 9)
    aligned = rng.random() < 0.4
    rows = [64 * rng.randint(1, 9) if aligned else rng.choice([0, 1, 63, 64, 65, 130, 300, 777]) if rng.random() < 0.6 else rng.randint(1, 900) for _ in range(P)]
    if sum(rows) == 0:
        rows[0] = 65
    d, Skv = H * 128, sum(rows)
    unit = rng.random() < 0.5
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 2 ** 31))
    q = (torch.randn((Sq, d), device=DEV, generator=g) * rng.choice([1.0, 1.0, 3.0])).to(torch.bfloat16)
    kv = torch.randn((Skv, 2 * d), device=DEV, generator=g).to(torch.bfloat16)
    scale = 128 ** -0.5
    if unit:
        kv[:, :d] = (kv[:, :d].float() * (scale * math.log2(math.e))).to(torch.bfloat16)
    call_scale = math.log(2.0) if unit else scale
    b = [0]
    for r in rows:
        b.append(b[-1] + r)
    order = list(range(P))
    if not aligned:
        rng.shuffle(order)
    ref = ref_attention(q, kv[:, :d], kv[:, d:], H, call_scale)
    late = [i for i in order[1:] if rows[i] and rng.random() < 0.4]
    staged = kv.clone() if late else None
    flags = torch.zeros((P,), dtype=torch.int32, device=DEV)
    err = torch.zeros((1,), dtype=torch.int32, device=DEV)
    what = f"attention_pieces Sq={Sq} rows={rows} order={order} late={late} H={H} unit={unit}"
    outs = []
    for rep in range(2):
        if late:
            for i in late:
                kv[b[i]:b[i + 1]] = float("nan")
            flags.zero_()
            torch.cuda.synchronize()
        pieces = [(kv[b[i]:b[i + 1], :d], kv[b[i]:b[i + 1], d:], (i if i in late else -1), 1) for i in order]
        o = torch.zeros((Sq, d), dtype=torch.bfloat16, device=DEV)
        ops.attention_pieces(q, pieces, o, H, call_scale, flags=flags, err=err, timeout_us=5_000_000)
        if late:
            with torch.cuda.stream(SIDE):
                for i in late:
                    kv[b[i]:b[i + 1]].copy_(staged[b[i]:b[i + 1]])
                    ops.flag_write(flags, i, 1, delay_us=rng.choice([0, 0, 200]))
            torch.cuda.synchronize()
        outs.append(o)
    if int(err.item()) != 0:
        fails.append(what + f": a wait gave up ({int(err.item()) & 0xffffffff:#x})")
        return
    ok = close_bf16(outs[0], ref, what, abs_floor=2.0 ** -5, rms_bound=2.0 ** -7)
    if ok and not torch.equal(outs[0], outs[1]):
        fails.append(what + ": two identical launches differ (race?)")
    if ok and aligned:
        ops.lib.icv_set_option(b"attn_kernel", 7); ops.lib.icv_set_option(b"attn7_variant", -1)
        o2 = torch.zeros((Sq, d), dtype=torch.bfloat16, device=DEV)
        ops.attention(q, kv[:, :d], kv[:, d:], o2, H, call_scale)
        if not torch.equal(outs[0], o2):
            fails.append(what + ": tile-aligned pieces in memory order differ from the plain launch")


SIDE = torch.cuda.Stream(device=DEV, priority=-1)      # see profiles/r06/stream_queue_share_probe.txt
t0, n = time.time(), {"gemm": 0, "attention": 0, "gemm_fp8": 0, "attention_fp8": 0, "conv": 0, "fp8_pieces": 0, "pieces": 0, "fp8_gated": 0}
R6_TOO = os.environ.get("FUZZ_R6", "0") == "1"       # FUZZ_R6=1 adds round 6's entry point: the arrival-driven attention over pieces
R5_TOO = os.environ.get("FUZZ_R5", "0") == "1"       # FUZZ_R5=1 adds round 5's entry points: the convolution and the e4m3 pieces
FP8_TOO = os.environ.get("FUZZ_FP8", "0") == "1"     # FUZZ_FP8=1 adds the e4m3 entry points (a different case sequence)
try:
    while (sum(n.values()) < max_cases) if max_cases > 0 else (time.time() - t0 < budget):
        if R6_TOO and rng.random() < 0.5:
            if FP8_TOO and rng.random() < 0.3:
                fuzz_attention_fp8_pieces_gated(); n["fp8_gated"] += 1
            else:
                fuzz_attention_pieces(); n["pieces"] += 1
        elif R5_TOO and rng.random() < 0.5:
            if rng.random() < 0.6:
                fuzz_conv(); n["conv"] += 1
            else:
                fuzz_attention_fp8_pieces(); n["fp8_pieces"] += 1
        elif FP8_TOO and rng.random() < 0.3:
            if rng.random() < 0.6:
                fuzz_gemm_fp8(); n["gemm_fp8"] += 1
            else:
                fuzz_attention_fp8(); n["attention_fp8"] += 1
        elif rng.random() < 0.6:
            fuzz_gemm(); n["gemm"] += 1
        else:
            fuzz_attention(); n["attention"] += 1
finally:
    ops.lib.icv_set_option(b"gemm256", 2); ops.lib.icv_set_option(b"gemm256_mfma", 16); ops.lib.icv_set_option(b"gemm256_sched", 3)
    ops.lib.icv_set_option(b"attn_kernel", 7); ops.lib.icv_set_option(b"attn7_variant", -1)
    ops.lib.icv_set_option(b"gemm_fp8_sched", 3)
torch.cuda.synchronize()
for f in fails[:40]:
    print("FAIL", f)
print(f"fuzz seed {seed}: {n['gemm']} GEMM cases, {n['attention']} attention cases" +
      (f", {n['gemm_fp8']} e4m3 GEMM cases, {n['attention_fp8']} e4m3 attention cases" if FP8_TOO else "") +
      (f", {n['conv']} convolution cases, {n['fp8_pieces']} e4m3-pieces cases" if R5_TOO else "") +
      (f", {n['pieces']} arrival-driven attention (pieces) cases" if os.environ.get("FUZZ_R6", "0") == "1" else "") +
      (f", {n['fp8_gated']} arrival-gated e4m3 pieces cases" if (R6_TOO and FP8_TOO) else "") +
      f" in {time.time() - t0:.0f} s, {len(fails)} failures")
sys.exit(1 if fails else 0)
