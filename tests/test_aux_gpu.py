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
