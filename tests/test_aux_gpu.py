"""Stock-PyTorch components outside the loop, on the GPU: the layout / MIOpen set-up choices of vae.py do not change results."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_vae_channels_last_matches_default_layout(monkeypatch):
    from infinicube_amd.videogen.vae import WanVAE, WanVAENet
    torch.manual_seed(0)
    net = WanVAENet(dim=32)                       # small width: same graph, seconds of MIOpen search
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    video = torch.rand((3, 9, 64, 96)) * 2 - 1
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("ICV_VAE_CHANNELS_LAST", mode)
        n = WanVAENet(dim=32)
        n.load_state_dict(sd)
        vae = WanVAE(n, "cuda:0", torch.bfloat16)
        assert vae.channels_last == (mode == "1")
        z = vae.encode(video, tiled=False)
        y = vae.decode(z, tiled=False)
        outs[mode] = (z.float().cpu(), y.float().cpu())
    for a, b, what in ((outs["0"][0], outs["1"][0], "latent"), (outs["0"][1], outs["1"][1], "video")):
        assert a.shape == b.shape
        rel = float((a - b).norm() / a.norm().clamp_min(1e-6))
        assert rel < 3e-2, f"{what}: NDHWC vs NCDHW rel-L2 {rel} (bf16 kernels differ only in summation order)"
    assert outs["1"][0].shape == (16, 3, 8, 12) and outs["1"][1].shape == (3, 9, 64, 96)


@pytest.mark.parametrize("C,rows,act", [(96, 1000, 1), (192, 777, 1), (384, 4099, 0), (16, 50, 1), (2048, 3, 0)])
def test_rmsnorm_act_rows_kernel(hip_ops, C, rows, act):
    """icv_rmsnorm_act_rows (csrc/vae_ops.hip: the VAE's channel RMS norm + SiLU as one pass over NDHWC rows) against the
    composite definition in fp64 - F.normalize(x, dim = channel) * sqrt(C) * gamma, then SiLU - within one bf16 rounding."""
    import math
    from infinicube_amd import native
    g = torch.Generator(device="cuda:0").manual_seed(C + rows)
    x = (torch.randn((rows, C), device="cuda:0", generator=g) * 3).to(torch.bfloat16)
    x[0] = 0                                                                       # an all-zero pixel: eps guards the divide
    gamma = torch.rand((C,), device="cuda:0", generator=g) + 0.5
    out = torch.empty_like(x)
    native.check(hip_ops.lib.icv_rmsnorm_act_rows(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), rows, C, math.sqrt(C), 1e-12, act,
                                                  torch.cuda.current_stream().cuda_stream), "icv_rmsnorm_act_rows")
    xd = x.double()
    ref = xd / xd.norm(dim=1, keepdim=True).clamp_min(1e-12) * math.sqrt(C) * gamma.double()
    if act:
        ref = ref * torch.sigmoid(ref)
    err = (out.double() - ref).abs()
    assert torch.isfinite(out).all() and bool((err <= 2.0 ** -8 * ref.abs() + 1e-6).all()), f"max err {float(err.max())}"
    assert float(out[0].abs().max()) == 0.0


def test_vae_hip_norm_matches_stock_ops(monkeypatch):
    """The VAE with its norms (+ SiLU) on the HIP row kernel and the causal pad folded into the convolutions against the same
    network on stock ops only: the same function up to bf16 rounding points (the fused pass rounds once where the composite
    ops round four times), and the state-dict layout is untouched."""
    from infinicube_amd.videogen.vae import WanVAE, WanVAENet
    torch.manual_seed(0)
    net = WanVAENet(dim=32)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    video = torch.rand((3, 9, 64, 96)) * 2 - 1
    outs = {}
    for mode in ("stock", "hip"):
        monkeypatch.setenv("ICV_VAE_NORM", mode)
        monkeypatch.setenv("ICV_VAE_PAD", "copy" if mode == "stock" else "conv")
        n = WanVAENet(dim=32)
        n.load_state_dict(sd)
        vae = WanVAE(n, "cuda:0", torch.bfloat16)
        assert vae.hip_norm == (mode == "hip") and set(vae.net.state_dict()) == set(sd)
        z = vae.encode(video, tiled=True, tile_size=(4, 6), tile_stride=(2, 3))
        outs[mode] = (z.float().cpu(), vae.decode(z, tiled=True, tile_size=(4, 6), tile_stride=(2, 3)).float().cpu())
    for a, b, what in ((outs["stock"][0], outs["hip"][0], "latent"), (outs["stock"][1], outs["hip"][1], "video")):
        rel = float((a - b).norm() / a.norm().clamp_min(1e-6))
        assert a.shape == b.shape and rel < 3e-2, f"{what}: HIP norm vs stock ops rel-L2 {rel}"
