"""Pins the CPU oracle (oracle/wan_ref.py).  The reference holds NO golden vectors or numeric tests
for this path and its arithmetic lives in an absent third-party package, so
parity with the reference is UNPINNED; what can be pinned is (i) every oracle op against the stock
PyTorch primitive of the same published definition, (ii) closed-form properties, (iii) committed
regression vectors (tests/golden/oracle_tiny.npz), (iv) the published FLOP/shape tables."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from infinicube_amd.videogen import synthetic as syn
from infinicube_amd.videogen.config import (GRID_480P, GRID_720P, GRID_CFG1, TokenGrid, dit_forward_flops,
                                            infer_config_from_state_dict, preset)
from infinicube_amd.videogen.scheduler import FlowMatchScheduler, flow_match_sigmas
from oracle import wan_ref as R

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_tiny.npz"))
CFG, GRID = preset("tiny"), TokenGrid(5, 64, 96)


def test_token_counts_and_flops_match_baseline_tables():
    # BASELINE.md §2 / SURVEY.md §8: S and F_fwd per config
    assert (GRID_CFG1.S, GRID_480P.S, GRID_720P.S) == (2240, 37440, 86400)
    assert GRID_480P.latent_shape() == (16, 24, 60, 104)
    assert abs(dit_forward_flops(preset("1.3b"), 37440) / 1e15 - 0.3557) < 5e-4
    assert abs(dit_forward_flops(preset("14b"), 37440) / 1e15 - 2.0613) < 5e-4
    assert abs(dit_forward_flops(preset("1.3b"), 2240) / 1e12 - 6.88) < 1e-2


def test_sampler_table():
    s = R.flow_match_sigmas(50)
    assert torch.allclose(s, torch.tensor(flow_match_sigmas(50), dtype=torch.float64), atol=1e-12)
    assert np.allclose(s.numpy(), G["sigmas50"], atol=1e-12)
    assert abs(float(s[0]) - 1.0) < 1e-12 and abs(float(s[25]) - 5 * 0.5 / (1 + 4 * 0.5)) < 1e-12
    sch = FlowMatchScheduler(10)
    assert abs(sum(sch.dsigma(i) for i in range(10)) + 1.0) < 1e-12    # telescopes from sigma=1 to 0
    assert sch.timesteps[0] == 1000.0


def test_ops_against_stock_torch_primitives():
    g = torch.Generator().manual_seed(0)
    x = torch.randn((33, 256), generator=g)
    w, b = torch.randn(256, generator=g), torch.randn(256, generator=g)
    assert torch.allclose(R.layer_norm(x, w, b, 1e-6), F.layer_norm(x, (256,), w, b, 1e-6))
    ref = x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6) * w
    assert torch.allclose(R.rms_norm(x, w, 1e-6), ref)
    q, k, v = (torch.randn((40, 256), generator=g) for _ in range(3))
    qh, kh, vh = (t.reshape(40, 2, 128).transpose(0, 1) for t in (q, k, v))
    manual = (torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(128), -1) @ vh).transpose(0, 1).reshape(40, 256)
    assert torch.allclose(R.attention(q, k, v, 2), manual, atol=1e-5)
    # the chunked explicit form the full-size GPU checker runs (scale in the GEMM's alpha) against torch's SDPA
    sdpa = F.scaled_dot_product_attention(qh[None], kh[None], vh[None])[0]
    assert torch.allclose(R.attention_explicit(qh, kh, vh, 1.0 / math.sqrt(128)), sdpa, atol=1e-5)
    qb, kb, vb = (torch.randn((3, 70, 64), generator=g) for _ in range(3))
    assert torch.allclose(R.attention_explicit(qb, kb, vb, 0.31), F.scaled_dot_product_attention(qb[None], kb[None], vb[None], scale=0.31)[0], atol=1e-5)
    lat = torch.randn((16, 2, 8, 12), generator=g)
    wt = torch.randn((32, 16, 1, 2, 2), generator=g)
    tok = R.patchify_tokens(lat, wt, None)
    assert tok.shape == (2 * 4 * 6, 32)
    assert torch.allclose(tok[7], (wt.reshape(32, -1) @ lat[:, 0, 2:4, 2:4].reshape(-1)), atol=1e-4)  # token (0,1,1)
    h = torch.randn((2 * 4 * 6, 64), generator=g)
    assert R.unpatchify(h, (2, 4, 6), 16).shape == (16, 2, 8, 12)
    # unpatchify is the inverse of the token view used by patchify
    tokview = lat.reshape(16, 2, 4, 2, 6, 2).permute(1, 2, 4, 3, 5, 0).reshape(48, 64)   # (x y z c) order
    assert torch.equal(R.unpatchify(tokview, (2, 4, 6), 16), lat)


def test_rope_properties():
    assert R.rope_axis_dims(128) == (44, 42, 42)
    T, Hp, Wp = 3, 4, 5
    f = R.rope_freqs_3d(128, T, Hp, Wp)
    assert f.shape == (60, 64) and f.dtype == torch.complex128
    assert torch.allclose(f.abs(), torch.ones(60, 64, dtype=torch.float64))
    assert torch.allclose(torch.view_as_real(f[0]), torch.tensor([1.0, 0.0], dtype=torch.float64).expand(64, 2))
    # token (f,h,w) = (1,2,3): pair 0 rotates by f*1, pair 22 by h*1, pair 43 by w*1 (theta^0 = 1)
    tok = (1 * Hp + 2) * Wp + 3
    ang = torch.angle(f[tok])
    assert abs(float(ang[0]) - 1.0) < 1e-12 and abs(float(ang[22]) - 2.0) < 1e-12 and abs(float(ang[43]) - 3.0) < 1e-12
    assert abs(float(ang[1]) - 1.0 * 10000 ** (-2 / 44)) < 1e-12
    # rotation preserves norms and makes q.k depend only on relative position along an axis
    g = torch.Generator().manual_seed(1)
    x = torch.randn((60, 256), generator=g)
    y = R.rope_apply(x, f, 2)
    assert torch.allclose(y.norm(dim=1), x.norm(dim=1), rtol=1e-5)
    q = torch.randn(128, generator=g).repeat(60, 1)
    k = torch.randn(128, generator=g).repeat(60, 1)
    qr, kr = R.rope_apply(q, f, 1), R.rope_apply(k, f, 1)
    t = lambda a, b, c: (a * Hp + b) * Wp + c   # noqa: E731
    d1 = float(qr[t(0, 1, 1)] @ kr[t(1, 2, 3)])
    d2 = float(qr[t(1, 2, 1)] @ kr[t(2, 3, 3)])
    assert abs(d1 - d2) < 1e-3
    assert np.allclose(torch.view_as_real(R.rope_freqs_3d(128, GRID.T, GRID.Hp, GRID.Wp)[37]).numpy(), G["rope_angle_sample"])


def test_sinusoidal_and_time_embed():
    e = R.sinusoidal_embedding_1d(256, torch.tensor([500.0], dtype=torch.float64))
    assert e.shape == (1, 256) and abs(float(e[0, 0]) - math.cos(500.0)) < 1e-12 and abs(float(e[0, 128]) - math.sin(500.0)) < 1e-12
    assert abs(float(e[0, 127]) - math.cos(500.0 * 10000 ** (-127 / 128))) < 1e-12


def test_golden_regression_vectors():
    sd = R.round_state_dict_to_bf16(syn.make_dit_state_dict(CFG))
    bsd = R.round_state_dict_to_bf16(syn.make_buffer_embedder_state_dict(CFG))
    noise, c1, c2 = syn.make_latent_noise(GRID), syn.make_text_context(CFG, 1), syn.make_text_context(CFG, 2)
    bl = syn.make_buffer_latents(CFG, GRID)
    buf = R.buffer_embed(bsd, bl)
    assert np.allclose(buf.numpy(), G["buf_tokens"], atol=1e-6)
    t, t_mod = R.time_embed(sd, CFG, 731.0)
    assert np.allclose(t.numpy(), G["t"], atol=1e-5) and np.allclose(t_mod.numpy(), G["t_mod"], atol=1e-5)
    v = R.dit_forward(sd, CFG, noise, c1, 731.0, buf)
    assert np.allclose(v.numpy(), G["velocity"], atol=2e-5)
    fin = R.denoise_loop(sd, bsd, CFG, noise, c1, c2, bl, num_steps=3)
    assert np.allclose(fin.numpy(), G["loop_final"], atol=1e-4)
    # fp32 oracle vs fp64 oracle: the restatement itself is numerically tight
    v64 = R.dit_forward(sd, CFG, noise, c1, 731.0, R.buffer_embed(bsd, bl, torch.float64), dtype=torch.float64)
    assert float((v.double() - v64).abs().max()) < 1e-5


def test_buffer_embedder_variants_and_zero_init():
    bl = syn.make_buffer_latents(CFG, GRID)
    for variant in ("concat", "dual"):
        z = syn.make_buffer_embedder_state_dict(CFG, variant=variant, zero_init=True)
        assert float(R.buffer_embed(z, bl).abs().max()) == 0.0     # zero-init embedder adds nothing
        nz = syn.make_buffer_embedder_state_dict(CFG, variant=variant)
        assert float(R.buffer_embed(nz, bl).abs().max()) > 0.0


def test_infer_config_from_shapes():
    for name in ("tiny", "small"):
        cfg = preset(name)
        got = infer_config_from_state_dict(syn.make_dit_state_dict(cfg))
        assert (got.dim, got.ffn_dim, got.num_layers, got.num_heads, got.text_dim, got.freq_dim) == \
               (cfg.dim, cfg.ffn_dim, cfg.num_layers, cfg.num_heads, cfg.text_dim, cfg.freq_dim)
