#!/usr/bin/env python3
"""Writes tests/golden/oracle_tiny.npz: seeded inputs + oracle outputs (fp32) for the tiny config —
one DiT forward, the intermediate ops and a 3-step CFG loop.  PARITY UNPINNED vs the reference (its
diffsynth dependency is absent); these vectors pin the oracle against regressions and give the GPU
tests a committed target.  Run: python tests/golden/make_oracle_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from infinicube_amd.videogen import synthetic as syn  # noqa: E402
from infinicube_amd.videogen.config import TokenGrid, preset  # noqa: E402
from oracle import wan_ref as R  # noqa: E402


def main():
    cfg, grid = preset("tiny"), TokenGrid(5, 64, 96)
    sd = R.round_state_dict_to_bf16(syn.make_dit_state_dict(cfg))
    bsd = R.round_state_dict_to_bf16(syn.make_buffer_embedder_state_dict(cfg))
    noise = syn.make_latent_noise(grid)
    c1, c2 = syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2)
    bl = syn.make_buffer_latents(cfg, grid)
    buf = R.buffer_embed(bsd, bl)
    t, t_mod = R.time_embed(sd, cfg, 731.0)
    v = R.dit_forward(sd, cfg, noise, c1, 731.0, buf)
    tok = R.dit_forward(sd, cfg, noise, c1, 731.0, buf, return_tokens=True)
    trace = []
    fin = R.denoise_loop(sd, bsd, cfg, noise, c1, c2, bl, num_steps=3, trace=trace)
    np.savez_compressed(
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_tiny.npz"),
        buf_tokens=buf.numpy(), t=t.numpy(), t_mod=t_mod.numpy(), velocity=v.numpy(), tokens=tok.numpy(),
        loop_step0=trace[0].numpy(), loop_final=fin.numpy(), sigmas50=R.flow_match_sigmas(50).numpy(),
        rope_angle_sample=torch.view_as_real(R.rope_freqs_3d(128, grid.T, grid.Hp, grid.Wp)[37]).numpy())
    print("wrote oracle_tiny.npz")


if __name__ == "__main__":
    main()
