"""PSNR of DECODED frames (uint8, peak 255) — the quantity BASELINE.json's ">= 40 dB" is stated on — next to the
latent PSNR of oracle/wan_ref.py::psnr (peak = max |ref latent|).  Both arms go through the SAME decoder.  Two decoders:

* ``frame_psnr``: the stand-in ``PoolVAE`` (fixed linear 16 -> 3 channel map + nearest upsampling).  It maps latent error to
  pixel error linearly - but a 16 -> 3 projection discards most of a latent error and the clamp hides what lands outside [-1, 1]
  (VERDICT r5): it can only SHRINK an error.  Kept because earlier rounds' records are stated on it.
* ``wan_vae_frame_psnr`` (round 6): the PRODUCT's Wan-VAE ARCHITECTURE (vae.WanVAENet at the public Wan2.1 sizes: 96 base
  channels, z = 16, causal 3-D convolutions, 4x temporal / 8x spatial up-sampling, the tiled decode the pipeline runs, every
  convolution on libicvideo) with SEEDED random weights - no checkpoint exists offline.  A deep non-linear decoder of the real
  shape: a latent error is mixed over channels, space and time the way the real decoder mixes it.  What it cannot say is how
  the TRAINED decoder weighs latent directions; its output gain is arbitrary, so ONE affine map fitted on the reference arm
  (zero mean, standard deviation 0.35: the spread of natural frames in [-1, 1]) is applied to both arms before the usual clamp
  and 8-bit rounding."""
import math

import torch


def frames_u8(latent, vae):
    v = vae.decode(latent.float().cpu())
    return ((v.clamp(-1, 1) + 1.0) * 127.5).round().to(torch.uint8)


def frame_psnr(lat_a, lat_b, vae=None):
    if vae is None:
        from standins import PoolVAE
        vae = PoolVAE()
    a, b = frames_u8(lat_a, vae).double(), frames_u8(lat_b, vae).double()
    mse = float(((a - b) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * math.log10(255.0 ** 2 / mse)


_WAN_VAE = {}


def product_wan_vae(device):
    """The product's tiled Wan-VAE (public architecture sizes, bf16, HIP convolutions) with seeded random weights; one per device."""
    key = str(device)
    if key not in _WAN_VAE:
        from infinicube_amd.videogen.vae import WanVAE, WanVAENet
        g = torch.random.get_rng_state()
        torch.manual_seed(20260929)
        net = WanVAENet()
        torch.random.set_rng_state(g)
        _WAN_VAE[key] = WanVAE(net, device, torch.bfloat16)
    return _WAN_VAE[key]


def wan_vae_frame_psnr(lat, ref, device="cuda:0"):
    """(frame PSNR in dB, peak 255; fraction of the reference arm's pixels that the clamp touched) of ``lat`` vs ``ref`` decoded
    by the product's Wan-VAE architecture (see the module docstring)."""
    vae = product_wan_vae(device)
    with torch.no_grad():
        vr = vae.decode(ref.to(device=device, dtype=torch.float32), tiled=True).float()
        va = vae.decode(lat.to(device=device, dtype=torch.float32), tiled=True).float()
    mu, sd = vr.mean(), vr.std().clamp_min(1e-12)
    fa, fb = ((v - mu) / sd * 0.35 for v in (va, vr))
    clipped = float((fb.abs() > 1).float().mean())
    a, b = (((v.clamp(-1, 1) + 1.0) * 127.5).round().double() for v in (fa, fb))
    mse = float(((a - b) ** 2).mean())
    return (float("inf") if mse == 0 else 10.0 * math.log10(255.0 ** 2 / mse)), clipped
