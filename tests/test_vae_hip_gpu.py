"""libicvideo's shifted-row convolution (csrc/conv.hip, icv_conv3d_ndhwc), the padded-volume norm (icv_rmsnorm_act_volume) and the
Wan-VAE tile networks executed on them (infinicube_amd/videogen/vae_hip.py) against plain PyTorch fp32 references of the same
ops (F.conv3d / F.conv2d on the CPU, the stock vae.WanVAENet in fp32) — SURVEY §8f row 4, VERDICT r4 item 2.  Tolerances: one
convolution = bf16 operands, fp32 accumulation, one bf16 rounding: |d| <= 2^-7 |ref| + 2^-8 rms(ref) (SURVEY §8d's per-op bar);
a whole tile network: rel-L2 <= 2e-2 / cosine >= 0.999 vs the fp32 network on the same bf16-rounded parameters."""
import ctypes

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from infinicube_amd.videogen import vae as V  # noqa: E402
from infinicube_amd.videogen import vae_hip as VH  # noqa: E402


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _per_op_ok(got, ref, what):
    got, ref = got.float().cpu(), ref.float().cpu()
    tol = ref.abs() * 2 ** -7 + ref.pow(2).mean().sqrt() * 2 ** -8
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} outside the per-op bar, max |d| {float((got - ref).abs().max()):.4g}"


CASES = [
    # name, module factory, taps, reference
    ("3x3x3 causal 96->96", lambda: V.CausalConv3d(96, 96, 3, padding=1), VH.TAPS_333),
    ("3x3x3 causal 192->384 (two N tiles)", lambda: V.CausalConv3d(192, 384, 3, padding=1), VH.TAPS_333),
    ("3x3x3 causal 32->12 (narrow N, padded filters)", lambda: V.CausalConv3d(32, 12, 3, padding=1), VH.TAPS_333),
    ("(3,1,1) causal 64->128", lambda: V.CausalConv3d(64, 128, (3, 1, 1), padding=(1, 0, 0)), VH.TAPS_311),
    ("1x1x1 96->192", lambda: V.CausalConv3d(96, 192, 1), VH.TAPS_111),
    ("per-frame 3x3 192->96", lambda: nn.Conv2d(192, 96, 3, padding=1), VH.TAPS_133),
]


@pytest.mark.parametrize("name,make,taps", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("with_resid", [False, True])
def test_conv_shift_matches_torch_conv(name, make, taps, with_resid):
    """One convolution of each geometry the VAE uses, odd sizes (partial row tiles, K-tiles that straddle taps at 96 channels),
    with and without the fused residual; reference = the stock module in fp32 on the CPU on the same bf16-rounded operands."""
    dev = _dev()
    torch.manual_seed(0)
    mod = make()
    with torch.no_grad():
        mod.weight.copy_(mod.weight.to(torch.bfloat16).float())
        mod.bias.copy_((torch.randn_like(mod.bias) * 0.1))
    cin = mod.weight.shape[1]
    T, H, W = 5, 7, 13
    x = (torch.randn(1, cin, T, H, W) * 0.5).to(torch.bfloat16)
    hip = VH.VaeHip(nn.Identity(), dev)
    xv = hip._to_vol(x.to(dev), cin)
    rv = None
    if with_resid:
        r = (torch.randn(1, mod.weight.shape[0], T, H, W)).to(torch.bfloat16)
        rv = hip._to_vol(r.to(dev), VH._ceil(mod.weight.shape[0], 4))
    out = hip.conv(xv, mod, taps, resid=rv)
    torch.cuda.synchronize()
    got = out.interior()[..., : mod.weight.shape[0]].permute(3, 0, 1, 2)[None]
    with torch.no_grad():
        ref = mod(x.float()) if isinstance(mod, V.CausalConv3d) else V._per_frame(mod, x.float())
        if with_resid:
            ref = ref + r.float()
    _per_op_ok(got, ref, name)
    # the padding frames of the output are zero (a following (3,1,1) convolution reads them)
    assert float(out.vol()[: VH.PT].abs().max()) == 0.0


def test_conv_shift_fuzz_random_geometries():
    """Random volumes and channel counts (every block width of the kernel, K-tiles that straddle taps, single-tile and
    many-tile row ranges, tails of every kind), each against the stock convolution in fp32 on the CPU."""
    import random
    dev = _dev()
    rng = random.Random(1234)
    hip = VH.VaeHip(nn.Identity(), dev)
    kinds = [("333", VH.TAPS_333), ("311", VH.TAPS_311), ("133", VH.TAPS_133), ("111", VH.TAPS_111)]
    for case in range(14):
        kind, taps = kinds[case % 4]
        cin = rng.choice([32, 64, 96, 160, 192])
        cout = rng.choice([4, 16, 32, 48, 96, 100, 192, 224, 384])
        T, H, W = rng.randint(1, 6), rng.randint(1, 19), rng.randint(1, 23)
        torch.manual_seed(100 + case)
        mod = {"333": lambda: V.CausalConv3d(cin, cout, 3, padding=1), "311": lambda: V.CausalConv3d(cin, cout, (3, 1, 1), padding=(1, 0, 0)),
               "133": lambda: nn.Conv2d(cin, cout, 3, padding=1), "111": lambda: V.CausalConv3d(cin, cout, 1)}[kind]()
        with torch.no_grad():
            mod.weight.copy_(mod.weight.to(torch.bfloat16).float())
        x = (torch.randn(1, cin, T, H, W) * 0.5).to(torch.bfloat16)
        out = hip.conv(hip._to_vol(x.to(dev), cin), mod, taps)
        torch.cuda.synchronize()
        got = out.interior()[..., :cout].permute(3, 0, 1, 2)[None]
        with torch.no_grad():
            ref = mod(x.float()) if isinstance(mod, V.CausalConv3d) else V._per_frame(mod, x.float())
        _per_op_ok(got, ref, f"fuzz case {case}: {kind} {cin}->{cout} on {T}x{H}x{W}")


def test_conv_full_tile_size_sampled_rows_and_linearity():
    """The decoder's largest layer at the size a 480p tile really has: a causal 3x3x3 convolution 96 -> 96 over a padded volume of
    [93 + 2, 240 + 2, 416 + 2] positions (9.6 M rows, 37 k row tiles, 41 K-tiles of which every third straddles two taps).
    (a) 768 sampled output positions (corners, edges, frame 0, random interior) against a direct fp32 evaluation of the 27 taps;
    (b) size-independent property: the kernel is exactly linear under scaling by a power of two - conv(2x) == 2 conv(x) BIT FOR BIT
    on every one of the 9.3 M real positions (no bias): any row mixed up between tiles, taps or K-halves would break it."""
    dev = _dev()
    torch.manual_seed(7)
    T, H, W, C = 93, 240, 416, 96
    mod = V.CausalConv3d(C, C, 3, padding=1, bias=False)
    with torch.no_grad():
        mod.weight.copy_((mod.weight * 4).to(torch.bfloat16).float())
    hip = VH.VaeHip(nn.Identity(), dev)
    xv = VH.Vol(T, H, W, C, dev)
    xv.interior().copy_((torch.randn((T, H, W, C), device=dev) * 0.5).to(torch.bfloat16))
    xv.zero_halo()
    out = hip.conv(xv, mod, VH.TAPS_333)
    torch.cuda.synchronize()
    g = torch.Generator().manual_seed(3)
    pos = [(0, 0, 0), (0, H - 1, W - 1), (T - 1, 0, W - 1), (T - 1, H - 1, 0), (1, 0, 5), (2, 7, 0), (T - 1, H - 1, W - 1)]
    pos += [(int(torch.randint(0, T, (1,), generator=g)), int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))) for _ in range(761)]
    t_i, h_i, w_i = (torch.tensor(v, device=dev) for v in zip(*pos))
    xp = xv.vol().float()                                                    # padded: frame t -> t + 2, pixel -> + 1
    w = mod.weight.detach().to(dev).float()                                  # [co, ci, 3, 3, 3]
    ref = torch.zeros((len(pos), C), device=dev)
    for dt in range(3):
        for dh in range(3):
            for dw in range(3):
                ref += xp[t_i + dt, h_i + dh, w_i + dw] @ w[:, :, dt, dh, dw].t()       # causal: frames t-2..t = padded t..t+2
    got = out.interior()[t_i, h_i, w_i].float()
    _per_op_ok(got, ref, "3x3x3 96->96 at the full 480p tile size, sampled positions")
    xv.mat.mul_(2)                                                           # exact in bf16
    out2 = hip.conv(xv, mod, VH.TAPS_333)
    torch.cuda.synchronize()
    a, b = out.interior(), out2.interior()
    assert torch.equal((a.float() * 2).to(torch.bfloat16), b), "conv(2x) != 2 conv(x): rows mixed up between tiles / taps / K-halves"
    assert float(a.float().abs().mean()) > 0.1


def test_conv_volume_beyond_4_gib_is_cut_into_row_ranges():
    """a lane's A-row position is a 32-bit byte offset from the operand base, so one launch covers < 4 GiB of rows - a
    192-channel (384 with the time-upsample's doubled channels) full-resolution decoder volume of a 240 x 416 tile reaches that at
    109 frames, an untiled 480p volume exceeds it outright.  icv_conv3d_ndhwc now cuts the rows into ranges with shifted bases.
    Here: 60 frames of [242, 418] positions x 384 channels = 4.66 GB of input rows, causal (3,1,1) taps (the time-upsample's
    geometry; its taps reach two FRAMES back across every range boundary) 384 -> 32.  (a) the rows of the last 3 frames - all of
    them beyond the 4 GiB mark - must be BIT-IDENTICAL to the same frames computed as their own short volume (same taps, same K
    order); (b) sampled positions across the whole volume, also next to the cut, against a direct fp32 evaluation."""
    dev = _dev()
    torch.manual_seed(9)
    T, H, W, C, CO = 60, 240, 416, 384, 32
    mod = V.CausalConv3d(C, CO, (3, 1, 1), padding=(1, 0, 0))
    with torch.no_grad():
        mod.weight.copy_((mod.weight * 4).to(torch.bfloat16).float())
    hip = VH.VaeHip(nn.Identity(), dev)
    xv = VH.Vol(T, H, W, C, dev)
    assert xv.rows * C * 2 > (1 << 32)
    for t0 in range(0, T, 10):                                               # filled in slabs: randn of the whole volume would need 9 GB of fp32
        xv.interior()[t0:t0 + 10].copy_((torch.randn((min(10, T - t0), H, W, C), device=dev) * 0.5).to(torch.bfloat16))
    xv.zero_halo()
    out = hip.conv(xv, mod, VH.TAPS_311)
    torch.cuda.synchronize()
    # (a) the last 3 frames as their own volume whose two causal padding frames hold frames T-5, T-4
    sv = VH.Vol(3, H, W, C, dev)
    sv.vol().copy_(xv.vol()[T - 3:])                                         # padded frame index = frame + 2: frames T-5 .. T-1
    so = hip.conv(sv, mod, VH.TAPS_311)
    torch.cuda.synchronize()
    assert torch.equal(out.interior()[T - 3:], so.interior()), "rows beyond the 4 GiB mark differ from the same rows computed on a short volume"
    # (b) sampled positions, fp32
    g = torch.Generator().manual_seed(5)
    cut_frame = int((((1 << 32) - 1) // (C * 2)) // 256 * 256 // xv.frame_rows)      # the real frame the first cut falls into (ranges start at the first real row)
    assert 2 < cut_frame < T - 2
    pos = [(0, 0, 0), (T - 1, H - 1, W - 1), (cut_frame, 0, 0), (cut_frame, H - 1, W - 1), (cut_frame + 1, 5, 7), (cut_frame - 1, 100, 200)]
    pos += [(int(torch.randint(0, T, (1,), generator=g)), int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))) for _ in range(250)]
    t_i, h_i, w_i = (torch.tensor(v, device=dev) for v in zip(*pos))
    xp = xv.vol()
    w = mod.weight.detach().to(dev).float()                                  # [co, ci, 3, 1, 1]
    ref = mod.bias.detach().to(dev).float()[None].repeat(len(pos), 1)
    for dt in range(3):
        ref += xp[t_i + dt, h_i + 1, w_i + 1].float() @ w[:, :, dt, 0, 0].t()
    _per_op_ok(out.interior()[t_i, h_i, w_i, :CO].float(), ref, "(3,1,1) 384->32 over a 4.66 GB volume, sampled positions")


def test_stride2_forms_match_torch():
    """The encoder's two strided convolutions through their stride-1 evaluation + subsampling (VaeHip.downsample)."""
    dev = _dev()
    torch.manual_seed(1)
    rs = V.Resample(64, "downsample3d")
    with torch.no_grad():
        for p in rs.parameters():
            p.copy_((p * 3).to(torch.bfloat16).float())
    x = (torch.randn(1, 64, 9, 8, 12) * 0.5).to(torch.bfloat16)
    hip = VH.VaeHip(nn.Identity(), dev)
    out = hip.downsample(rs, hip._to_vol(x.to(dev), 64))
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = rs(x.float())
    assert (out.T, out.H, out.W) == tuple(ref.shape[2:])
    got = out.interior().permute(3, 0, 1, 2)[None]
    rel = float((got.float().cpu() - ref).norm() / ref.norm())
    assert rel < 1e-2, rel          # two chained convolutions, the first one's output rounded to bf16


def test_upsample3d_matches_torch():
    dev = _dev()
    torch.manual_seed(2)
    rs = V.Resample(64, "upsample3d")
    with torch.no_grad():
        for p in rs.parameters():
            p.copy_((p * 3).to(torch.bfloat16).float())
    x = (torch.randn(1, 64, 4, 5, 6) * 0.5).to(torch.bfloat16)
    hip = VH.VaeHip(nn.Identity(), dev)
    out = hip.upsample(rs, hip._to_vol(x.to(dev), 64))
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = rs(x.float())
    assert (out.T, out.H, out.W) == tuple(ref.shape[2:])
    rel = float((out.interior()[..., :32].permute(3, 0, 1, 2)[None].float().cpu() - ref).norm() / ref.norm())
    assert rel < 1e-2, rel


def test_volume_norm_writes_zero_padding():
    dev = _dev()
    torch.manual_seed(3)
    norm = V.RMS_norm(96, images=False)
    with torch.no_grad():
        norm.gamma.copy_(torch.randn_like(norm.gamma))
    x = torch.randn(1, 96, 3, 5, 7).to(torch.bfloat16)
    hip = VH.VaeHip(nn.Identity(), dev)
    xv = hip._to_vol(x.to(dev), 96)
    xv.buf.view(torch.int16)[xv.buf.view(torch.int16) == 0] = 0x7FC0          # NaNs wherever the padding was: the kernel must overwrite them
    xv.interior().copy_(x[0].permute(1, 2, 3, 0))
    out = hip.norm_act(xv, norm, 1)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = F.silu(norm(x.float()))
    _per_op_ok(out.interior().permute(3, 0, 1, 2)[None], ref, "rms norm + SiLU")
    full = out.vol().float()
    full[VH.PT:, 1:-1, 1:-1] = 0
    assert torch.isfinite(full).all() and float(full.abs().max()) == 0.0, "padding frames / halo must be written as zeros"


@pytest.fixture(scope="module")
def tiny_vae():
    torch.manual_seed(4)
    ref = V.WanVAENet(dim=32, z_dim=16).eval()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    return ref


def _net_close(got, ref, what, rel_bound=2e-2):
    got, ref = got.float().cpu(), ref.float().cpu()
    rel = float((got - ref).norm() / ref.norm())
    cos = float(F.cosine_similarity(got.flatten(), ref.flatten(), dim=0))
    print(f"{what}: rel-L2 {rel:.3g}, cosine {cos:.6f}")
    assert rel <= rel_bound and cos >= 0.999, f"{what}: rel-L2 {rel}, cosine {cos}"


def test_tile_decoder_matches_fp32_network(tiny_vae):
    import copy
    dev = _dev()
    vae = V.WanVAE(copy.deepcopy(tiny_vae), dev)
    assert vae.hip is not None, "the HIP convolution path must be the default on the GPU"
    z = torch.randn(1, 16, 3, 6, 8).to(torch.bfloat16)
    with torch.no_grad():
        got = vae._net_decode(z.to(dev).contiguous(memory_format=torch.channels_last_3d))
        torch.cuda.synchronize()
        ref = tiny_vae.decode(z.float())
    assert got.shape == ref.shape
    _net_close(got, ref, "tile decoder (HIP convolutions) vs fp32 network")


def test_tile_encoder_matches_fp32_network(tiny_vae):
    import copy
    dev = _dev()
    vae = V.WanVAE(copy.deepcopy(tiny_vae), dev)
    x = (torch.rand(1, 3, 9, 32, 48) * 2 - 1).to(torch.bfloat16)
    with torch.no_grad():
        got = vae._net_encode(x.to(dev).contiguous(memory_format=torch.channels_last_3d))
        torch.cuda.synchronize()
        ref = tiny_vae.encode(x.float())
    assert got.shape == ref.shape
    _net_close(got, ref, "tile encoder (HIP convolutions) vs fp32 network")


def test_hip_path_equals_miopen_path_within_bf16(tiny_vae, monkeypatch):
    """The public encode / decode (tiled, blended) on the HIP convolutions vs the same calls on the stock MIOpen modules."""
    import copy
    dev = _dev()
    a = V.WanVAE(copy.deepcopy(tiny_vae), dev)
    monkeypatch.setenv("ICV_VAE_CONV", "miopen")
    b = V.WanVAE(copy.deepcopy(tiny_vae), dev)
    assert a.hip is not None and b.hip is None
    lat = torch.randn(16, 3, 12, 16)
    kw = dict(tiled=True, tile_size=(8, 8), tile_stride=(4, 4))
    va, vb = a.decode(lat, **kw), b.decode(lat, **kw)
    _net_close(va, vb, "tiled decode: HIP convolutions vs MIOpen")
    clip = torch.rand(3, 9, 96, 128) * 2 - 1
    kw = dict(tiled=True, tile_size=(8, 8), tile_stride=(4, 4))
    ea, eb = a.encode(clip, **kw), b.encode(clip, **kw)
    _net_close(ea, eb, "tiled encode: HIP convolutions vs MIOpen")
