"""Parity at the sizes BASELINE.json is quoted on (configs #2, #3, #5): S = 37 440 tokens (93 f 480x832
[R infinicube/inference/guidance_buffer_generation.py:79-82,744-745]) and S = 86 400 (93 f 720x1280).

  * self-attention alone at full S, sampled query rows vs the CPU oracle (bf16 default route, key-chunked
    carried-state route, e4m3 route);
  * ONE DiT block at the real Wan2.1-14B (t2v) and 14B-i2v dimensions over ALL tokens, checked on a token
    slice whose keys / values come from all S tokens (CPU oracle);
  * config #2 end to end: Wan2.1-1.3B, S = 37 440, 10 flow-match steps with CFG and guidance buffers rendered
    from a synthetic scene, against the SAME oracle code (oracle/wan_ref.py) executed in fp32 by stock PyTorch on
    the GPU with the explicit matmul-softmax-matmul attention — a checker that shares nothing with libicvideo.
Tolerances: SURVEY.md §8d (one forward: cosine >= 0.999, rel-L2 <= 2e-2; loop: PSNR >= 40 dB).
"""
import dataclasses
import math
import os
import time

import numpy as np
import pytest
import torch

from oracle import wan_ref as R
from infinicube_amd.videogen import synthetic as syn
from infinicube_amd.videogen.config import GRID_480P, GRID_720P, preset
from infinicube_amd.videogen.dit import WanDiT
from infinicube_amd.videogen.scheduler import FlowMatchScheduler
from infinicube_amd.videogen.seqpar import chunk_bounds

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LN2 = math.log(2.0)
FOLD = (1.0 / math.sqrt(128)) * math.log2(math.e)


def _rnd(shape, seed, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * std


def _attn_case(S, H, n_rows):
    """bf16 q, k (carrying the folded softmax scale, as the DiT hands it over), v at full S and the sampled rows.
    Some sampled queries get a strongly matching key deep in the sequence (tiles 3, S/128, the last) so the
    running reference of the lazy-max path is outgrown late, not only in the first tile."""
    d = H * 128
    q = _rnd((S, d), 901).to(torch.bfloat16)
    kf = _rnd((S, d), 902)
    v = _rnd((S, d), 903).to(torch.bfloat16)
    g = torch.Generator().manual_seed(904)
    rows = torch.sort(torch.randperm(S, generator=g)[:n_rows]).values
    rows[0], rows[-1] = 0, S - 1
    for j, key in enumerate((3 * 64 + 5, S // 2 + 17, S - 1, S - 64 * 9 - 3)):
        kf[key] = q[rows[7 * j + 1]].float() * (3.0 + j)
    k = (kf * FOLD).to(torch.bfloat16)
    return q, k, v, rows


def _check_rows(o, ref, what, rms_bound=2.0 ** -7, abs_floor=2.0 ** -5):
    got, ref = o.float().cpu(), ref.float()
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    rms = ref.pow(2).mean().sqrt()
    rms_err = (got - ref).pow(2).mean().sqrt()
    assert rms_err <= rms_bound * rms, f"{what}: rms err {rms_err:.4g} > {rms_bound:.3g} * rms {rms:.4g}"
    bad = (got - ref).abs() > (2.0 ** -7) * ref.abs() + abs_floor * rms
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} outside tol, max err {(got - ref).abs().max():.4g}"


@pytest.mark.parametrize("S", [GRID_480P.S, GRID_720P.S])
def test_attention_full_size(hip_ops, S):
    """K6 at the metric's own sequence lengths (585 / 1350 key tiles per row): the default route (attn7, unit scale,
    lazy max), the key-chunked carried-state route the sequence-parallel path uses (4 ramped chunks), and the e4m3
    route, each checked on 512 sampled query rows against the CPU oracle."""
    H = 2
    q, k, v, rows = _attn_case(S, H, 512)
    ref = R.attention(q[rows].float(), k.float(), v.float(), H, scale=LN2)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o = torch.zeros((S, H * 128), dtype=torch.bfloat16, device=DEV)
    hip_ops.attention(qd, kd, vd, o, H, LN2)
    _check_rows(o[rows.to(DEV)], ref, f"attention S={S}")
    o2 = torch.zeros_like(o)
    hip_ops.attention(qd, kd, vd, o2, H, LN2)
    assert torch.equal(o, o2), "full-size attention is not deterministic"
    # key chunks with carried (O, m, l) state
    acc = torch.empty((S, H * 128), device=DEV)
    ml = torch.empty((S, H, 2), device=DEV)
    b = chunk_bounds(S, 4)
    o3 = torch.zeros_like(o)
    for c in range(4):
        hip_ops.attention_chunk(qd, kd[b[c]:b[c + 1]], vd[b[c]:b[c + 1]], o3, acc, ml, H, LN2, first=(c == 0), last=(c == 3))
    _check_rows(o3[rows.to(DEV)], ref, f"chunked attention S={S}")
    # e4m3 operands (config #5's mode): vs the oracle with the same per-head power-of-two quantisation
    ws = hip_ops.attention_fp8_buffers(S, S, H * 128, H)
    o8 = torch.zeros_like(o)
    hip_ops.attention_fp8(qd, kd, vd, o8, H, ws)
    sub = rows[:128]
    ref8 = R.attention_fp8(q[sub].float(), k.float(), v.float(), H)
    got8 = o8[sub.to(DEV)].float().cpu()
    assert torch.isfinite(got8).all()
    rel8 = float((got8 - ref8).pow(2).mean().sqrt() / ref8.pow(2).mean().sqrt())
    rel8_exact = float((got8 - ref[:128]).pow(2).mean().sqrt() / ref[:128].pow(2).mean().sqrt())
    print(f"S={S}: fp8 attention rms err vs e4m3 oracle {rel8:.4f}, vs exact {rel8_exact:.4f}")
    assert rel8 <= 0.03, f"fp8 attention S={S}: rms err {rel8} vs the oracle with the same quantisation"


def _block_case(model, grid, n_slice, fp8=False):
    """One block at the model's real width over ALL S tokens on the GPU; the CPU oracle recomputes the tokens of one
    slice (LayerNorm/modulate and the K, V projections + RoPE for all tokens, everything else for the slice)."""
    cfg = dataclasses.replace(preset(model), num_layers=1)
    sd = syn.make_dit_state_dict(cfg, seed=0, dtype=torch.bfloat16)
    bsd = syn.make_buffer_embedder_state_dict(cfg, dtype=torch.bfloat16)
    sdr, bsdr = {k: v.float() for k, v in sd.items()}, {k: v.float() for k, v in bsd.items()}
    noise, ctx, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_buffer_latents(cfg, grid)
    clip = syn.make_clip_features(cfg) if cfg.has_image_input else None
    y = syn.make_cond_latents(cfg, grid) if cfg.has_image_input else None
    ts = 731.0
    return cfg, sd, bsd, sdr, bsdr, noise, ctx, bl, clip, y, ts


def _oracle_block_slice(sdr, bsdr, cfg, grid, noise, ctx, bl, clip, y, ts, sl, fp8):
    f32 = torch.float32
    lat = noise if y is None else torch.cat([noise, y], 0)
    x_all = R.patchify_tokens(lat, sdr["patch_embedding.weight"], sdr["patch_embedding.bias"]) + R.buffer_embed(bsdr, bl)
    _, t_mod = R.time_embed(sdr, cfg, ts)
    ctx_e = R.text_embed(sdr, ctx)
    ctx_img = R.image_embed(sdr, clip) if clip is not None else None
    freqs = R.rope_freqs_3d(cfg.head_dim, grid.T, grid.Hp, grid.Wp)
    p = "blocks.0"
    lin = R._lin8 if (fp8 and "wqkv" in fp8) else R._lin
    mod = sdr[f"{p}.modulation"].reshape(6, cfg.dim) + t_mod
    h_all = R.modulate(R.layer_norm(x_all, None, None, cfg.eps), mod[0], mod[1])
    k_all = R.rope_apply(R.rms_norm(lin(sdr, f"{p}.self_attn.k", h_all, f32), sdr[f"{p}.self_attn.norm_k.weight"], cfg.eps), freqs, cfg.num_heads)
    v_all = lin(sdr, f"{p}.self_attn.v", h_all, f32)
    del h_all
    x_out = R.dit_block(sdr, cfg, 0, x_all[sl], ctx_e, t_mod, freqs[sl], kv_override=lambda k, v: (k_all, v_all),
                        ctx_img=ctx_img, fp8=fp8)
    return x_all[sl], x_out


def _run_block(hip_ops, model, grid, sl, gemm_dtype="bf16", attn_dtype="bf16"):
    cfg, sd, bsd, sdr, bsdr, noise, ctx, bl, clip, y, ts = _block_case(model, grid, sl)
    # fp8: ALL six projections in e4m3 here (a kernel-level check at the real shapes against the oracle with the same
    # quantisation; the accuracy of the shipped default subset is the loop tests' business)
    m = WanDiT(cfg, sd, hip_ops, bsd, gemm_dtype=gemm_dtype, attn_dtype=attn_dtype, fp8_weights=WanDiT.FP8_WEIGHTS).prepare(grid, graphs=False)
    ck = m.encode_context(ctx, clip)
    bt = m.embed_buffers(bl)
    if y is not None:
        bt = m.embed_cond_latents(y, add_to=bt)
    lat = noise.to(DEV)
    m.forward_tokens(lat, ck, ts, bt, m.head_out[0], num_layers=0)
    x_in = m.x[sl].clone()
    m.forward_tokens(lat, ck, ts, bt, m.head_out[0], num_layers=1)
    torch.cuda.synchronize()
    x_out = m.x[sl].float().cpu()
    x_in = x_in.float().cpu()
    del m
    t0 = time.time()
    rx_in, rx_out = _oracle_block_slice(sdr, bsdr, cfg, grid, noise, ctx, bl, clip, y, ts, sl, fp8=(R.FP8_ALL if gemm_dtype == "fp8" else False))
    t_cpu = time.time() - t0
    u, ur = x_out - x_in, rx_out - rx_in                       # the block's update of the residual stream
    rel_in = float((x_in - rx_in).norm() / rx_in.norm())
    rel = float((u - ur).norm() / ur.norm())
    cos = float(torch.nn.functional.cosine_similarity(u.flatten().double(), ur.flatten().double(), dim=0))
    print(f"{model} S={grid.S} {gemm_dtype}/{attn_dtype}: block-update rel-L2 {rel:.4g}, cosine {cos:.6f}, "
          f"input rel-L2 {rel_in:.3g}, oracle {t_cpu:.1f}s on CPU")
    return rel_in, rel, cos


def test_layer_14b_full_S(hip_ops):
    """Config #3's layer: d = 5120, ffn = 13824, 40 heads, S = 37 440 — every kernel of a block (K1, K3-K10) at the
    benchmarked shape, checked on tokens [17000, 19048) (mid-grid RoPE offsets; keys / values from all tokens)."""
    rel_in, rel, cos = _run_block(hip_ops, "14b", GRID_480P, slice(17000, 17000 + 2048))
    assert rel_in <= 2.0 ** -8, f"patch + buffer embed at 14B width: rel-L2 {rel_in}"   # bf16 patch operand: 2^-9 relative rounding
    assert cos >= 0.999 and rel <= 2e-2, f"14B block at S=37440: rel-L2 {rel}, cosine {cos}"


@pytest.mark.parametrize("mode", ["bf16", "fp8"])
def test_layer_14b_i2v_720p(hip_ops, mode):
    """Config #5's layer: Wan2.1-14B i2v (36 input channels, CLIP cross-attention branch) at 93 f 720x1280,
    S = 86 400.  bf16 against the oracle; fp8 (e4m3 projections + e4m3 self-attention, what
    torch_dtype=float8_e4m3fn selects) against the oracle with the same row quantisation of the projections."""
    sl = slice(40000, 40000 + 1024)
    if mode == "bf16":
        rel_in, rel, cos = _run_block(hip_ops, "14b-i2v", GRID_720P, sl)
        assert rel_in <= 2.0 ** -8
        assert cos >= 0.999 and rel <= 2e-2, f"14B i2v block at S=86400: rel-L2 {rel}, cosine {cos}"
    else:
        rel_in, rel, cos = _run_block(hip_ops, "14b-i2v", GRID_720P, sl, gemm_dtype="fp8", attn_dtype="fp8")
        assert rel_in <= 2.0 ** -8
        assert cos >= 0.998 and rel <= 6e-2, f"14B i2v fp8 block at S=86400: rel-L2 {rel}, cosine {cos}"


@pytest.mark.parametrize("world,attn_dtype", [(4, "bf16"), (8, "bf16"), (8, "fp8"), (4, "bf16+arrival"), (8, "bf16+arrival")])
def test_layer_14b_sequence_parallel_shards_full_S(hip_ops, world, attn_dtype):
    """Config #4's per-rank work at its real size on ONE GPU: a Wan2.1-14B block at S = 37 440 computed as the token shards
    of a `world`-rank sequence-parallel run (4 = the shards of the cfg2 x sp4 layout `bench.py --gpus 8` builds, 8 = plain
    sp8) — K|V row matrix [n, 2d], ramped 4-chunk exchange, carried-state attention over 37 440 keys with strided K / V
    halves, RoPE at the shard's token offset — with the exchange served from the unsharded run's K / V (the only part a
    1-GPU box cannot do; real ranks: tests/test_multigpu_rccl.py).  First, middle and last shard against the rows of the
    unsharded HIP block (a self-comparison: only the softmax merge order differs; e4m3 attention: per-chunk K / V scales), and the
    middle shard ALSO against oracle/wan_ref.py's rows for 1024 of its tokens - oracle parity at the exact per-rank shapes
    (n = 9 360 and n = 4 680 query rows against 37 440 keys in 4 ramped chunks).
    "+arrival" (round 6): the same shards through ONE arrival-gated attention launch over the pieces (csrc/attn7p.hip: own rows in
    place, then 4 chunks x (world - 1) peers = up to 29 pieces, chunk bounds on the 64-key tile grid) instead of the 4 chunk launches."""
    from infinicube_amd.videogen.seqpar import ShardPlan
    arrival = attn_dtype.endswith("+arrival")
    attn_dtype = attn_dtype.split("+")[0]
    cfg, grid, chunks = dataclasses.replace(preset("14b"), num_layers=1), GRID_480P, 4
    sd = syn.make_dit_state_dict(cfg, seed=0, device=DEV, dtype=torch.bfloat16)
    bsd = syn.make_buffer_embedder_state_dict(cfg, device=DEV, dtype=torch.bfloat16)
    noise, ctx, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_buffer_latents(cfg, grid)
    kw = dict(attn_dtype=attn_dtype)
    rec = []
    raw, raw8 = hip_ops.attention, hip_ops.attention_fp8

    def rec_attention(q, k, v, o, heads, scale):
        if k.shape[0] == grid.S:
            rec.append((k.clone(), v.clone()))
        raw(q, k, v, o, heads, scale)

    def rec_attention8(q, k, v, o, heads, ws):
        rec.append((k.clone(), v.clone()))
        raw8(q, k, v, o, heads, ws)

    hip_ops.attention, hip_ops.attention_fp8 = rec_attention, rec_attention8
    try:
        full = WanDiT(cfg, sd, hip_ops, bsd, **kw).prepare(grid, graphs=False)
        lat = noise.to(DEV)
        fck, fbt = full.encode_context(ctx), full.embed_buffers(bl)
        full.forward_tokens(lat, fck, 731.0, fbt, full.head_out[0], num_layers=0)
        x_in = full.x.clone()                                   # residual stream entering the block (patch + buffer embed)
        full.forward_tokens(lat, fck, 731.0, fbt, full.head_out[0], num_layers=1)
        torch.cuda.synchronize()
        want = full.x.clone()
        del fck, fbt
    finally:
        hip_ops.attention, hip_ops.attention_fp8 = raw, raw8
    del full
    kf, vf = rec[0]
    for r in (0, world // 2, world - 1):
        plan = ShardPlan.make(grid.S, world, r)
        n = plan.n_tok

        class ServedGather:            # seqpar.KVGather's interface; the peers' rows come from the unsharded run
            def __init__(self):
                self.r0, self.n_collectives, self.timing = 0, 0, None

            def start(self, rows, out):
                dd, m = rows.shape[1] // 2, rows.shape[0]
                assert out.shape[0] == world * m
                mine = kf[plan.tok0 + self.r0: plan.tok0 + self.r0 + m]
                assert torch.equal(mine, rows[:, :dd]), "this shard's K rows differ from the unsharded run's (RoPE offset / row layout)"
                for rk in range(world):
                    out[rk * m:(rk + 1) * m, :dd].copy_(kf[rk * n + self.r0: rk * n + self.r0 + m])
                    out[rk * m:(rk + 1) * m, dd:].copy_(vf[rk * n + self.r0: rk * n + self.r0 + m])
                self.r0 += m
                self.n_collectives += 1
                return ()

            def wait(self, handle):
                pass

            # arrival-driven consumer: every piece is there (no flag); own rows are read in place
            def enable_arrival(self, ops):
                pass

            def arrival(self, handle):
                return None, [(j, -1, 0) for j in range(world) if j != plan.rank]

            def consumed(self, handle):
                pass

        g = ServedGather()
        m = WanDiT(cfg, sd, hip_ops, bsd, **kw).prepare(grid, plan, kv_gather=g, sp_chunks=chunks, graphs=False,
                                                        kv_exchange="allgather+arrival" if arrival else None)
        assert m.attn_arrival == arrival and (not arrival or all(b % 64 == 0 for b in m.sp_bounds[1:-1]))
        m.forward_tokens(lat, m.encode_context(ctx), 731.0, m.embed_buffers(bl), m.head_out[0], num_layers=1)
        torch.cuda.synchronize()
        assert g.n_collectives == chunks and g.r0 == n
        sl = slice(plan.tok0, plan.tok0 + n)
        got, ref = m.x - x_in[sl], want[sl] - x_in[sl]          # the block's UPDATE of the residual stream
        rel = float((got - ref).norm() / ref.norm())
        print(f"14B block, S={grid.S}, shard {r} of {world} ({n} tokens, attention {attn_dtype}): block update vs the unsharded block rel-L2 {rel:.3g}")
        assert torch.isfinite(got).all() and rel < (3e-2 if attn_dtype == "fp8" else 2e-3), f"shard {r}/{world}: rel-L2 {rel}"
        if r == world // 2 and attn_dtype == "bf16":
            # ... and against the ORACLE itself at this shard shape (not only against the unsharded HIP run): oracle/wan_ref.py in
            # fp32 on the CPU recomputes 1024 of this shard's tokens (keys / values from all 37 440 tokens) - same bars as the
            # unsharded block test (test_layer_14b_full_S)
            o0 = plan.tok0 + min(1000, n - 1024)
            osl = slice(o0, o0 + 1024)
            sdr = {k: v.float().cpu() for k, v in sd.items()}
            bsdr = {k: v.float().cpu() for k, v in bsd.items()}
            rx_in, rx_out = _oracle_block_slice(sdr, bsdr, cfg, grid, noise, ctx, bl, None, None, 731.0, osl, fp8=False)
            loc = slice(o0 - plan.tok0, o0 - plan.tok0 + 1024)
            u, ur = (m.x[loc] - x_in[osl]).float().cpu(), rx_out - rx_in
            rel_o = float((u - ur).norm() / ur.norm())
            cos_o = float(torch.nn.functional.cosine_similarity(u.flatten().double(), ur.flatten().double(), dim=0))
            print(f"14B block, shard {r} of {world} ({n} query rows x {grid.S} keys in {chunks} ramped chunks) vs the fp32 ORACLE rows: block-update rel-L2 {rel_o:.4g}, cosine {cos_o:.6f}")
            assert cos_o >= 0.999 and rel_o <= 2e-2, f"shard shape n = {n} vs the oracle: rel-L2 {rel_o}, cosine {cos_o}"
            del sdr, bsdr
        del m


def test_config4_rank_forward_full_depth_on_one_gpu(hip_ops):
    """Config #4's per-rank work at FULL depth on one GPU: the whole Wan2.1-14B forward (40 layers, S = 37 440) of ONE rank of the
    8-GPU job — token shard 1/4 of the cfg2 x sp4 layout `bench.py --gpus 8` builds, and shard 1/8 of plain sp8 — with every
    layer's K|V exchange served from the unsharded forward's own K / V (recorded per layer: 31 GB, which 288 GB of HBM hold).
    What real ranks add is the transport only (tests/test_multigpu_rccl.py).  Bar: the shard's velocity tokens against the
    unsharded forward's rows — rel-L2 <= 8e-3 (measured 4.6e-3), cosine >= 0.9999 ("~ bit-close: only the softmax merge order changes", 40 layers
    deep)."""
    from infinicube_amd.videogen.seqpar import ShardPlan
    cfg, grid, chunks = preset("14b"), GRID_480P, 4
    sd = syn.make_dit_state_dict(cfg, seed=0, device=DEV, dtype=torch.bfloat16)
    bsd = syn.make_buffer_embedder_state_dict(cfg, device=DEV, dtype=torch.bfloat16)
    noise, ctx, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_buffer_latents(cfg, grid)
    rec = []
    raw = hip_ops.attention

    def rec_attention(q, k, v, o, heads, scale):
        if k.shape[0] == grid.S:
            rec.append((k.clone(), v.clone()))
        raw(q, k, v, o, heads, scale)

    hip_ops.attention = rec_attention
    try:
        full = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid, graphs=False)
        lat = noise.to(DEV)
        ck, bt = full.encode_context(ctx), full.embed_buffers(bl)
        full.forward_tokens(lat, ck, 731.0, bt, full.head_out[0])
        torch.cuda.synchronize()
        want = full.head_out[0].clone()
    finally:
        hip_ops.attention = raw
    del full, ck, bt
    assert len(rec) == cfg.num_layers
    for world, r in ((4, 1), (8, 5)):
        plan = ShardPlan.make(grid.S, world, r)
        n = plan.n_tok

        class ServedGather:            # seqpar.KVGather's interface; the peers' rows of EVERY layer come from the unsharded run
            def __init__(self):
                self.layer, self.r0, self.n_collectives, self.timing = 0, 0, 0, None

            def start(self, rows, out):
                dd, m = rows.shape[1] // 2, rows.shape[0]
                kf, vf = rec[self.layer]
                for rk in range(world):
                    out[rk * m:(rk + 1) * m, :dd].copy_(kf[rk * n + self.r0: rk * n + self.r0 + m])
                    out[rk * m:(rk + 1) * m, dd:].copy_(vf[rk * n + self.r0: rk * n + self.r0 + m])
                out[r * m:(r + 1) * m].copy_(rows)            # this rank's own rows are its own (they carry the shard's drift)
                self.r0 += m
                self.n_collectives += 1
                if self.r0 == n:
                    self.layer, self.r0 = self.layer + 1, 0
                return ()

            def wait(self, handle):
                pass

        g = ServedGather()
        m = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid, plan, kv_gather=g, sp_chunks=chunks, graphs=False)
        m.forward_tokens(lat, m.encode_context(ctx), 731.0, m.embed_buffers(bl), m.head_out[0])
        torch.cuda.synchronize()
        assert g.n_collectives == chunks * cfg.num_layers and g.layer == cfg.num_layers
        got, ref = m.head_out[0].float(), want[plan.tok0: plan.tok0 + n].float()
        rel = float((got - ref).norm() / ref.norm())
        cos = float(torch.nn.functional.cosine_similarity(got.flatten().double(), ref.flatten().double(), dim=0))
        print(f"config #4 on one GPU: 14B forward at full depth, rank {r} of a {world}-rank sequence-parallel group ({n} tokens): velocity tokens vs the "
              f"unsharded forward's rows rel-L2 {rel:.3g}, cosine {cos:.6f}")
        assert torch.isfinite(got).all() and rel <= 8e-3 and cos >= 0.9999, f"rank {r}/{world}: rel-L2 {rel}, cosine {cos}"
        del m


from psnr_util import frame_psnr, wan_vae_frame_psnr  # noqa: E402

# the frame bar through the product's Wan-VAE ARCHITECTURE (seeded weights; tests/psnr_util.py): north_star's ">= 40 dB" is a statement about FRAMES
WAN_VAE_FRAME_BAR = 40.0


def test_config2_wan_1p3b_93f_480p(hip_ops):
    """BASELINE.json config #2: Wan2.1-1.3B t2v, 93 frames 480x832 (S = 37 440), "real voxel guidance buffers": a synthetic voxel world
    (point cloud with Waymo classes) ray-cast by the product's voxel renderer into depth / class / instance maps, turned
    into the coordinate and colour buffers by the product's buffer kernels (uint8), stand-in VAE encode, 10 flow-match steps with CFG
    (round 4 ran 4; the stated 50 steps are a recorded opt-in run, profiles/r05/parity_config2_50_steps.txt).  HIP loop vs oracle/wan_ref.py run in
    fp32 on the GPU by stock PyTorch.  Bars: final-latent PSNR >= 40 dB, decoded-frame PSNR (peak 255) >= 40 dB."""
    from infinicube_amd.utils.buffer_utils import generate_coordinate_buffer_from_memory_global_norm
    from infinicube_amd.utils.semantic_utils import generate_rgb_semantic_buffer, semantic_to_color
    from infinicube_amd.videogen.pipeline import _video_to_tensor
    from standins import PoolVAE
    from PIL import Image
    cfg, grid = preset("1.3b"), GRID_480P
    from infinicube_amd.utils.voxel_render import render_voxel_buffers
    torch.manual_seed(0); np.random.seed(0)
    # stage 2's own flow on the GPU: voxel world -> ray-cast depth / class / instance maps -> the two guidance buffers
    pts, psem, pinst, cam, poses = syn.make_voxel_world(grid)
    depth, sem, inst = render_voxel_buffers(cam, poses, pts, psem, pinst)
    assert depth.shape == (93, 480, 832) and 0.02 < float((depth == 0).float().mean()) < 0.6
    coord_u8 = generate_coordinate_buffer_from_memory_global_norm(depth, cam, poses, return_uint8=True).cpu().numpy()
    sem_rgb = (semantic_to_color(sem) * 255).astype(np.uint8)
    sem_u8 = generate_rgb_semantic_buffer(sem_rgb, inst.cpu().numpy().astype(np.uint16))
    assert coord_u8.shape == sem_u8.shape == (93, 480, 832, 3)
    assert (coord_u8 == 255).all(-1).mean() > 0.05, "the scene has sky (coordinate buffer = 1.0 there)"
    vae = PoolVAE()
    lats = [vae.encode(_video_to_tensor([Image.fromarray(f) for f in buf], grid.height, grid.width)) for buf in (sem_u8, coord_u8)]
    bl = torch.cat(lats, 0).float()
    sd = syn.make_dit_state_dict(cfg, seed=0, dtype=torch.bfloat16)
    bsd = syn.make_buffer_embedder_state_dict(cfg, dtype=torch.bfloat16)
    noise = syn.make_latent_noise(grid)
    c1, c2 = syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2)
    steps = 10
    m = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid)
    lat = noise.clone().to(DEV)
    t0 = time.time()
    m.denoise(lat, m.encode_context(c1), m.encode_context(c2), m.embed_buffers(bl), FlowMatchScheduler(steps), 5.0)
    torch.cuda.synchronize()
    t_hip = time.time() - t0
    lat = lat.cpu()
    del m
    torch.cuda.empty_cache()
    # the checker: the oracle's own code on GPU tensors, fp32 (no bf16 anywhere except the shared weight values)
    sdr = {k: v.float().to(DEV) for k, v in sd.items()}
    bsdr = {k: v.float().to(DEV) for k, v in bsd.items()}
    t0 = time.time()
    ref = R.denoise_loop(sdr, bsdr, cfg, noise.to(DEV), c1.to(DEV), c2.to(DEV), bl.to(DEV), num_steps=steps)
    torch.cuda.synchronize()
    t_ref = time.time() - t0
    ref = ref.cpu()
    p = R.psnr(lat, ref)
    pf = frame_psnr(lat, ref, vae)
    del sdr, bsdr
    torch.cuda.empty_cache()
    pw, clipped = wan_vae_frame_psnr(lat, ref, DEV)
    cos = float(torch.nn.functional.cosine_similarity((lat - noise).flatten().double(), (ref - noise).flatten().double(), dim=0))
    line = (f"config #2, {steps}-step loop: HIP {t_hip:.1f}s, fp32 torch oracle on GPU {t_ref:.1f}s; latent PSNR {p:.1f} dB (range-free SNR {R.snr_db(lat, ref):.1f} dB); "
            f"frame PSNR through the product's Wan-VAE architecture (seeded weights) {pw:.1f} dB ({100 * clipped:.1f} % of the pixels clamped), "
            f"through the pooling stand-in {pf:.1f} dB; update cosine {cos:.5f}")
    print(line)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_config2_loop.txt", "w") as f:
        f.write(line + "\n")
    assert p >= 40.0 and pf >= 40.0 and cos >= 0.999, f"config #2 parity: latent {p:.1f} dB, frames {pf:.1f} dB, cosine {cos}"
    assert pw >= WAN_VAE_FRAME_BAR, f"config #2: frame PSNR through the Wan-VAE architecture {pw:.1f} dB < {WAN_VAE_FRAME_BAR} dB"


def test_config3_wan_14b_full_depth_forwards_and_loop(hip_ops):
    """BASELINE.json config #3 at FULL depth and size - Wan2.1-14B (40 layers, d = 5120), 93 f 480x832 (S = 37 440) - against
    oracle/wan_ref.py executed in fp32 by stock PyTorch on the GPU on the same bf16-rounded weight values, with ONE oracle run
    (four oracle steps = eight 40-layer fp32 forwards, ~4 GPU-minutes) serving three checks:
      * one FORWARD: the cond and the uncond velocity of the first step (x = noise, t = 1000), each cosine >= 0.999 and
        rel-L2 <= 2e-2; the CFG-combined velocity v_u + 5 (v_c - v_u) amplifies the difference of two nearly
        equal forwards five times, so it is held to cosine >= 0.999 / rel-L2 <= 5e-2;
      * the LOOP: a complete 4-step CFG-5 flow-match schedule from noise to sigma 0, product loop (WanDiT.denoise) vs the
        oracle's: final-latent PSNR >= 40 dB (north star) and decoded-frame PSNR >= 40 dB through the same pooling VAE;
      * the e4m3 MODE (config #5's kernels: FP8_DEFAULT projections + e4m3 self-attention, what torch_dtype=float8_e4m3fn selects)
        over the same loop against the UNQUANTISED oracle: >= 40 dB as well.
    (Rounds 2-3 ran a one-step test and a 4-step loop on separate oracle runs - 5 oracle steps, 330 s; round 4 a 2-step loop; the
    50-step runs of this config are recorded opt-in runs: profiles/r05/parity_config3_50_steps.txt.)"""
    from standins import PoolVAE
    cfg, grid, steps = preset("14b"), GRID_480P, 4
    sd = syn.make_dit_state_dict(cfg, seed=0, device=DEV, dtype=torch.bfloat16)
    bsd = syn.make_buffer_embedder_state_dict(cfg, device=DEV, dtype=torch.bfloat16)
    noise = syn.make_latent_noise(grid)
    c1, c2, bl = syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2), syn.make_buffer_latents(cfg, grid)
    sched = FlowMatchScheduler(steps)
    ts0 = float(sched.timesteps[0])
    gshape = (grid.T, grid.Hp, grid.Wp)
    got = {}
    for mode in ("bf16", "fp8"):
        kw = {} if mode == "bf16" else dict(gemm_dtype="fp8", attn_dtype="fp8")
        m = WanDiT(cfg, sd, hip_ops, bsd, **kw).prepare(grid, graphs=False)
        lat = noise.clone().to(DEV)
        ck, cu, bt = m.encode_context(c1), m.encode_context(c2), m.embed_buffers(bl)
        if mode == "bf16":      # the two forwards of the first step on their own (what bench.py's step is made of)
            m.forward_tokens(lat, ck, ts0, bt, m.head_out[0])
            m.forward_tokens(lat, cu, ts0, bt, m.head_out[1])
            vc_hip = R.unpatchify(m.head_out[0].cpu(), gshape, cfg.out_dim)
            vu_hip = R.unpatchify(m.head_out[1].cpu(), gshape, cfg.out_dim)
        torch.cuda.synchronize()
        t0 = time.time()
        m.denoise(lat, ck, cu, bt, FlowMatchScheduler(steps), 5.0)
        torch.cuda.synchronize()
        got[mode] = (lat.cpu(), time.time() - t0)
        del m, ck, cu, bt, lat
        torch.cuda.empty_cache()
    sdr = {k: v.float() for k, v in sd.items()}
    bsdr = {k: v.float() for k, v in bsd.items()}
    del sd, bsd
    torch.cuda.empty_cache()
    # the oracle's loop (oracle/wan_ref.denoise_loop, spelled out so that the first step's two velocities can be kept)
    t0 = time.time()
    buf = R.buffer_embed(bsdr, bl.to(DEV))
    x = noise.to(DEV).clone()
    first = None
    for i in range(steps):
        ts = float(sched.timesteps[i])
        v_c = R.dit_forward(sdr, cfg, x, c1.to(DEV), ts, buf)
        v_u = R.dit_forward(sdr, cfg, x, c2.to(DEV), ts, buf)
        v = v_u + 5.0 * (v_c - v_u)
        if first is None:
            first = (v_c.cpu(), v_u.cpu(), v.cpu())
        x = x + v * sched.dsigma(i)
        del v_c, v_u, v
    ref = x.cpu()
    torch.cuda.synchronize()
    t_ref = time.time() - t0
    del sdr, bsdr, buf, x
    torch.cuda.empty_cache()
    chk = R.denoise_loop   # (same arithmetic: tests/test_oracle.py pins denoise_loop; this spelling only keeps intermediates)
    assert chk is not None

    def cmp(a, b):
        return (float((a - b).norm() / b.norm()),
                float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0)))

    (rc, cc), (ru, cu_) = cmp(vc_hip, first[0]), cmp(vu_hip, first[1])
    rv, cv = cmp(vu_hip + 5.0 * (vc_hip - vu_hip), first[2])
    lines = [f"config #3, first step's forwards at full depth: cond rel-L2 {rc:.4g} cos {cc:.6f}; uncond rel-L2 {ru:.4g} cos {cu_:.6f}; "
             f"CFG velocity rel-L2 {rv:.4g} cos {cv:.6f}"]
    print(lines[-1])
    assert cc >= 0.999 and rc <= 2e-2 and cu_ >= 0.999 and ru <= 2e-2, f"config #3 forward parity: cond {rc}/{cc}, uncond {ru}/{cu_}"
    assert cv >= 0.999 and rv <= 5e-2, f"config #3 CFG velocity: rel-L2 {rv}, cosine {cv}"
    for mode, (lat, t_hip) in got.items():
        p = R.psnr(lat, ref)
        pf = frame_psnr(lat, ref, PoolVAE())
        pw, clipped = wan_vae_frame_psnr(lat, ref, DEV)
        lines.append(f"config #3, {steps}-step loop at full depth, product {mode}: HIP {t_hip:.1f}s, fp32 torch oracle on GPU {t_ref:.1f}s; latent PSNR {p:.1f} dB "
                     f"(range-free SNR {R.snr_db(lat, ref):.1f} dB); frame PSNR through the product's Wan-VAE architecture (seeded weights) {pw:.1f} dB "
                     f"({100 * clipped:.1f} % of the pixels clamped), through the pooling stand-in {pf:.1f} dB")
        print(lines[-1])
        assert torch.isfinite(lat).all() and p >= 40.0 and pf >= 40.0, f"config #3 {steps}-step loop ({mode}): latent PSNR {p:.1f} dB, frame PSNR {pf:.1f} dB"
        assert pw >= WAN_VAE_FRAME_BAR, f"config #3 ({mode}): frame PSNR through the Wan-VAE architecture {pw:.1f} dB < {WAN_VAE_FRAME_BAR} dB"
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_config3_loop_full_depth.txt", "w") as f:
        f.write("\n".join(lines) + "\n")


SLOW = int(os.environ.get("ICV_SLOW_TESTS", "0") or 0)


@pytest.mark.parametrize("model,need", [("1.3b", 1), ("14b", 2)])
def test_configs_2_and_3_at_50_steps(hip_ops, model, need):
    """BASELINE.json configs #2 / #3 at the step count they state: 93 f 480x832, FIFTY flow-match steps with CFG (the
    pipeline's default, which the reference never overrides [R infinicube/videogen/inference.py:216-226]), bf16 product and
    the e4m3 mode against the fp32 oracle loop run by stock PyTorch on the GPU.  Bars: latent and decoded-frame PSNR >= 40 dB.
    Opt-in (the oracle's 100 forwards take ~10 GPU-minutes for 1.3B, ~50 for 14B): ICV_SLOW_TESTS=1 runs 1.3B, =2 both;
    recorded in profiles/r05/parity_config{2,3}_50_steps.txt."""
    if SLOW < need:
        pytest.skip(f"50 oracle steps at S = 37 440 for Wan2.1-{model}: run with ICV_SLOW_TESTS={need} (recorded in profiles/r05/parity_config{2 if model == '1.3b' else 3}_50_steps.txt)")
    from standins import PoolVAE
    cfg, grid, steps = preset(model), GRID_480P, 50
    sd = syn.make_dit_state_dict(cfg, seed=0, device=DEV, dtype=torch.bfloat16)
    bsd = syn.make_buffer_embedder_state_dict(cfg, device=DEV, dtype=torch.bfloat16)
    noise = syn.make_latent_noise(grid)
    c1, c2, bl = syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2), syn.make_buffer_latents(cfg, grid)
    got = {}
    arms = tuple(os.environ.get("ICV_SLOW_ARMS", "bf16,fp8").split(","))      # a GPU lease is at most one hour: the 14B run fits with one arm
    for mode in arms:
        kw = {} if mode == "bf16" else dict(gemm_dtype="fp8", attn_dtype="fp8")
        m = WanDiT(cfg, sd, hip_ops, bsd, **kw).prepare(grid, graphs=False)
        lat = noise.clone().to(DEV)
        ck, cu, bt = m.encode_context(c1), m.encode_context(c2), m.embed_buffers(bl)
        torch.cuda.synchronize()
        t0 = time.time()
        m.denoise(lat, ck, cu, bt, FlowMatchScheduler(steps), 5.0)
        torch.cuda.synchronize()
        got[mode] = (lat.cpu(), time.time() - t0)
        del m, ck, cu, bt, lat
        torch.cuda.empty_cache()
    sdr = {k: v.float() for k, v in sd.items()}
    bsdr = {k: v.float() for k, v in bsd.items()}
    del sd, bsd
    torch.cuda.empty_cache()
    # The 14B oracle loop takes ~49 GPU-minutes and a lease is one hour: its final latent (a pure function of the seeds above) is
    # written to gpurun_out/ and, when a copy is found under oracle/_ref/ (git-ignored, travels to the GPU box), reused - the second
    # arm then runs in its own lease against the SAME oracle run.
    cache_name = f"oracle_latent_config{2 if model == '1.3b' else 3}_{steps}_steps.pt"
    cached = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", cache_name)
    # what the cached latent is a function of: the oracle's source, the config, and the seeded inputs (a stale blob must not pass)
    import hashlib
    hk = hashlib.sha256()
    hk.update(open(R.__file__, "rb").read())
    hk.update(repr((cfg, grid, steps, 5.0)).encode())
    for t in (noise, c1, c2, bl, sdr["blocks.0.self_attn.q.weight"].cpu(), sdr["head.head.weight"].cpu()):
        hk.update(t.detach().float().contiguous().numpy().tobytes())
    key = hk.hexdigest()
    blob = torch.load(cached) if os.path.exists(cached) else None
    if blob is not None and "key" not in blob and os.environ.get("ICV_ADOPT_UNKEYED_ORACLE") == "1":
        # one-time adoption of a blob written before the key existed (round 5's 49-minute oracle run; oracle/wan_ref.py and the seeds are
        # unchanged since): it is re-saved WITH the key under gpurun_out/
        print(f"adopting the un-keyed cached oracle latent {cached} (made by round 5's run of this test) under key {key[:12]}")
        blob["key"] = key
        os.makedirs("gpurun_out", exist_ok=True)
        torch.save(blob, os.path.join("gpurun_out", cache_name))
    if blob is not None and blob.get("key") != key:
        print(f"cached oracle latent {cached} was made from other inputs / another oracle (key {str(blob.get('key'))[:12]} != {key[:12]}): recomputing")
        blob = None
    if blob is not None:
        ref, t_ref = blob["latent"], blob["seconds"]
        del sdr, bsdr
    else:
        t0 = time.time()
        ref = R.denoise_loop(sdr, bsdr, cfg, noise.to(DEV), c1.to(DEV), c2.to(DEV), bl.to(DEV), num_steps=steps).cpu()
        torch.cuda.synchronize()
        t_ref = time.time() - t0
        os.makedirs("gpurun_out", exist_ok=True)
        torch.save(dict(latent=ref, seconds=t_ref, key=key), os.path.join("gpurun_out", cache_name))
    lines = []
    for mode, (lat, t_hip) in got.items():
        p, pf = R.psnr(lat, ref), frame_psnr(lat, ref, PoolVAE())
        pw, clipped = wan_vae_frame_psnr(lat, ref, DEV)
        lines.append(f"config #{2 if model == '1.3b' else 3}, Wan2.1-{model} 93f 480x832 (S = {grid.S}), {steps} steps CFG 5, product {mode}: HIP {t_hip:.1f}s, "
                     f"fp32 torch oracle on GPU {t_ref:.1f}s{' (same oracle run, reused)' if os.path.exists(cached) else ''}; latent PSNR {p:.1f} dB "
                     f"(range-free SNR {R.snr_db(lat, ref):.1f} dB); frame PSNR through the product's Wan-VAE architecture (seeded weights) {pw:.1f} dB "
                     f"({100 * clipped:.1f} % clamped), through the pooling stand-in {pf:.1f} dB")
        print(lines[-1])
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/parity_config{2 if model == '1.3b' else 3}_50_steps{'_' + '_'.join(arms) if len(arms) < 2 else ''}.txt", "w") as f:
        f.write("\n".join(lines) + "\n")
    for mode, (lat, _) in got.items():
        assert torch.isfinite(lat).all() and R.psnr(lat, ref) >= 40.0 and frame_psnr(lat, ref, PoolVAE()) >= 40.0, f"50 steps ({mode}) under the 40 dB bar: {lines}"


def test_config5_wan_14b_i2v_720p_loop_full_depth(hip_ops):
    """Config #5's LOOP on one GPU at full depth and size: Wan2.1-14B image-to-video (CLIP cross-attention branch, 36 input
    channels), 93 frames 720x1280 (S = 86 400), a complete 3-step CFG-5 flow-match schedule from noise to sigma 0 (6 forwards of
    40 layers), bf16 product and the e4m3 mode against oracle/wan_ref.denoise_loop in fp32 by stock PyTorch on the GPU.
    Opt-in (the checker needs ~17 GPU-minutes): ICV_SLOW_TESTS=2; recorded in profiles/r03/parity_config5_loop_full_depth.txt."""
    if SLOW < 2:
        pytest.skip("6 fp32 oracle forwards at S = 86 400 (~17 GPU-minutes): run with ICV_SLOW_TESTS=2 (recorded in profiles/r03/parity_config5_loop_full_depth.txt)")
    from standins import PoolVAE
    cfg, grid, steps = preset("14b-i2v"), GRID_720P, 3
    sd = syn.make_dit_state_dict(cfg, seed=0, device=DEV, dtype=torch.bfloat16)
    bsd = syn.make_buffer_embedder_state_dict(cfg, device=DEV, dtype=torch.bfloat16)
    noise, c1, c2, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2), syn.make_buffer_latents(cfg, grid)
    clip, y = syn.make_clip_features(cfg), syn.make_cond_latents(cfg, grid)
    got = {}
    for mode in ("bf16", "fp8"):
        kw = {} if mode == "bf16" else dict(gemm_dtype="fp8", attn_dtype="fp8")
        m = WanDiT(cfg, sd, hip_ops, bsd, **kw).prepare(grid, graphs=False)
        ck, cu = m.encode_context(c1, clip), m.encode_context(c2, clip)
        add = m.embed_cond_latents(y, add_to=m.embed_buffers(bl))
        lat = noise.clone().to(DEV)
        torch.cuda.synchronize()
        t0 = time.time()
        m.denoise(lat, ck, cu, add, FlowMatchScheduler(steps), 5.0)
        torch.cuda.synchronize()
        got[mode] = (lat.cpu(), time.time() - t0)
        del m, ck, cu, add, lat
        torch.cuda.empty_cache()
    sdr = {k: v.float() for k, v in sd.items()}
    bsdr = {k: v.float() for k, v in bsd.items()}
    del sd, bsd
    torch.cuda.empty_cache()
    t0 = time.time()
    ref = R.denoise_loop(sdr, bsdr, cfg, noise.to(DEV), c1.to(DEV), c2.to(DEV), bl.to(DEV), num_steps=steps, clip_fea=clip.to(DEV), y=y.to(DEV)).cpu()
    torch.cuda.synchronize()
    t_ref = time.time() - t0
    lines = []
    for mode, (lat, t_hip) in got.items():
        p, pf = R.psnr(lat, ref), frame_psnr(lat, ref, PoolVAE())
        lines.append(f"config #5 on one GPU, Wan2.1-14B i2v 93f 720x1280 (S = {grid.S}), {steps}-step CFG-5 loop at full depth, product {mode}: HIP {t_hip:.1f}s, "
                     f"fp32 torch oracle on GPU {t_ref:.1f}s; latent PSNR {p:.1f} dB (SNR {R.snr_db(lat, ref):.1f} dB), decoded-frame PSNR {pf:.1f} dB")
        print(lines[-1])
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_config5_loop_full_depth.txt", "w") as f:
        f.write("\n".join(lines) + "\n")
    for mode, (lat, _) in got.items():
        assert torch.isfinite(lat).all() and R.psnr(lat, ref) >= 40.0 and frame_psnr(lat, ref, PoolVAE()) >= 40.0, f"config #5 loop ({mode}) under the 40 dB bar: {lines}"


def test_config5_wan_14b_i2v_720p_one_forward_full_depth(hip_ops):
    """Config #5 at FULL depth and size: Wan2.1-14B image-to-video (36 input channels, CLIP cross-attention branch), 93 frames
    720x1280 (S = 86 400), ONE conditional forward of all 40 layers; bf16 product and the e4m3 mode (torch_dtype =
    float8_e4m3fn's kernels) against oracle/wan_ref.py run UNQUANTISED in fp32 by stock PyTorch on the GPU."""
    cfg, grid = preset("14b-i2v"), GRID_720P
    sd = syn.make_dit_state_dict(cfg, seed=0, device=DEV, dtype=torch.bfloat16)
    bsd = syn.make_buffer_embedder_state_dict(cfg, device=DEV, dtype=torch.bfloat16)
    noise, c1, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_buffer_latents(cfg, grid)
    clip, y = syn.make_clip_features(cfg), syn.make_cond_latents(cfg, grid)
    sched = FlowMatchScheduler(50)
    ts = float(sched.timesteps[0])
    gshape = (grid.T, grid.Hp, grid.Wp)
    got = {}
    for mode in ("bf16", "fp8"):
        kw = {} if mode == "bf16" else dict(gemm_dtype="fp8", attn_dtype="fp8")
        m = WanDiT(cfg, sd, hip_ops, bsd, **kw).prepare(grid, graphs=False)
        ck = m.encode_context(c1, clip)
        add = m.embed_cond_latents(y, add_to=m.embed_buffers(bl))
        lat = noise.clone().to(DEV)
        torch.cuda.synchronize()
        t0 = time.time()
        m.forward_tokens(lat, ck, ts, add, m.head_out[0])
        torch.cuda.synchronize()
        got[mode] = (R.unpatchify(m.head_out[0].cpu(), gshape, cfg.out_dim), time.time() - t0)
        del m, ck, add
        torch.cuda.empty_cache()
    sdr = {k: v.float() for k, v in sd.items()}
    bsdr = {k: v.float() for k, v in bsd.items()}
    del sd, bsd
    torch.cuda.empty_cache()
    t0 = time.time()
    buf = R.buffer_embed(bsdr, bl.to(DEV))
    v = R.dit_forward(sdr, cfg, noise.to(DEV), c1.to(DEV), ts, buf, clip_fea=clip.to(DEV), y=y.to(DEV)).cpu()
    torch.cuda.synchronize()
    print(f"config #5, Wan2.1-14B i2v, 93 f 720x1280, S={grid.S}, one forward of {cfg.num_layers} layers; fp32 torch oracle on the GPU: {time.time() - t0:.1f} s")
    res = {}
    for mode, (vh, t) in got.items():
        rel = float((vh - v).norm() / v.norm())
        cos = float(torch.nn.functional.cosine_similarity(vh.flatten().double(), v.flatten().double(), dim=0))
        p = R.psnr(noise + vh * sched.dsigma(0), noise + v * sched.dsigma(0))
        res[mode] = (rel, cos)
        print(f"  product {mode:4s}: {t:6.2f} s   velocity rel-L2 {rel:.4g}  cosine {cos:.6f}   latent PSNR after one Euler step {p:.1f} dB")
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_config5_full_depth.txt", "w") as f:
        f.write("".join(f"config #5 (14B i2v, 93f 720x1280, S={grid.S}, 40 layers, one forward) product {mo}: velocity rel-L2 {r[0]:.4g} cosine {r[1]:.6f}\n" for mo, r in res.items()))
    assert res["bf16"][1] >= 0.999 and res["bf16"][0] <= 2e-2, f"config #5 bf16 forward: {res['bf16']}"
    assert res["fp8"][1] >= 0.998 and res["fp8"][0] <= 8e-2, f"config #5 e4m3 forward vs the unquantised oracle: {res['fp8']}"
