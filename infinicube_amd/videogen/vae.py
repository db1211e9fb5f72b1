"""Wan2.1 3-D VAE loader (outside the hot loop; stock PyTorch-ROCm).

[R infinicube/videogen/inference.py:69,79] names the file ``Wan2.1_VAE.pth``.  Tiled encode/decode of
the VAE is SURVEY.md §8f row 4 ("next"): not restated in round 1.  ``from_pretrained`` fails loudly."""

import glob


def load_wan_vae(pattern, device):
    files = sorted(glob.glob(pattern))
    if not files:
        raise FileNotFoundError(f"Wan VAE checkpoint not found: {pattern!r} (skip_download=True: nothing is fetched)")
    raise NotImplementedError(
        "Wan-VAE on stock PyTorch-ROCm is a 'next' row (SURVEY.md §8f-4) and not built yet; construct "
        "WanVideoPipeline(vae=...) with any object exposing encode(video)/decode(latent)")
