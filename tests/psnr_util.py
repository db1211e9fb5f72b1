"""PSNR of DECODED frames (uint8, peak 255) — the quantity BASELINE.json's ">= 40 dB" is stated on — next to the
latent PSNR of oracle/wan_ref.py::psnr (peak = max |ref latent|).  Both arms go through the SAME decoder; the
stand-in ``PoolVAE`` (fixed linear 16 -> 3 channel map + nearest upsampling) is used because no Wan-VAE weights exist
offline: it cannot hide or amplify a difference selectively, it only maps latent error to pixel error linearly (then
the usual clamp and 8-bit rounding)."""
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
