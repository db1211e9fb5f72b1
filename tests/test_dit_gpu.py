"""End-to-end parity of the HIP DiT path (host driver + C ABI kernels) against the CPU oracle.

Bars: one forward cosine >= 0.999 and rel-L2 <= 2e-2; denoise loop final-latent
PSNR >= 40 dB vs the fp32 oracle fed the same bf16-rounded weights.
"""
import os

import pytest
import torch

from oracle import wan_ref as R
from oracle_ops import OracleOps
from infinicube_amd.videogen import synthetic as syn
from infinicube_amd.videogen.config import TokenGrid, preset
from infinicube_amd.videogen.dit import WanDiT
from infinicube_amd.videogen.scheduler import FlowMatchScheduler

pytestmark = pytest.mark.gpu


def _setup(name, grid, variant="concat"):
    cfg = preset(name)
    sd = syn.make_dit_state_dict(cfg)
    bsd = syn.make_buffer_embedder_state_dict(cfg, variant=variant)
    return cfg, sd, bsd, R.round_state_dict_to_bf16(sd), R.round_state_dict_to_bf16(bsd)


@pytest.mark.parametrize("name,grid,variant", [
    ("tiny", TokenGrid(5, 64, 96), "concat"),
    ("tiny", TokenGrid(9, 80, 112), "dual"),
    ("small", TokenGrid(17, 128, 160), "concat"),
])
def test_forward_parity(hip_ops, name, grid, variant):
    cfg, sd, bsd, sdr, bsdr = _setup(name, grid, variant)
    noise, ctx = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1)
    bl = syn.make_buffer_latents(cfg, grid)
    m = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid)
    ck = m.encode_context(ctx)
    bt = m.embed_buffers(bl)
    lat = noise.to("cuda:0")
    m.forward_tokens(lat, ck, 731.0, bt, m.head_out[0])
    torch.cuda.synchronize()
    v = R.unpatchify(m.head_out[0].cpu(), (grid.T, grid.Hp, grid.Wp), cfg.out_dim)
    vref = R.dit_forward(sdr, cfg, noise, ctx, 731.0, R.buffer_embed(bsdr, bl))
    rel = float((v - vref).norm() / vref.norm())
    cos = float(torch.nn.functional.cosine_similarity(v.flatten(), vref.flatten(), dim=0))
    assert cos >= 0.999 and rel <= 2e-2, f"forward parity: cos={cos} rel-L2={rel}"
    # the buffer conditioning must actually matter (non-zero embedder) and be applied
    m.forward_tokens(lat, ck, 731.0, None, m.head_out[1])
    torch.cuda.synchronize()
    assert float((m.head_out[0] - m.head_out[1]).abs().max()) > 1e-3
    # rounding-point-exact CPU emulation of the same pipeline agrees much tighter
    e = WanDiT(cfg, sd, OracleOps(), bsd).prepare(grid)
    e.forward_tokens(noise.clone(), e.encode_context(ctx), 731.0, e.embed_buffers(bl), e.head_out[0])
    rel_e = float((m.head_out[0].cpu() - e.head_out[0]).norm() / e.head_out[0].norm())
    assert rel_e <= 1e-2, f"HIP vs bf16-emulated host pipeline rel-L2={rel_e}"


def test_denoise_loop_psnr(hip_ops):
    grid = TokenGrid(9, 64, 96)
    cfg, sd, bsd, sdr, bsdr = _setup("tiny", grid)
    noise = syn.make_latent_noise(grid)
    c1, c2 = syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2)
    bl = syn.make_buffer_latents(cfg, grid)
    steps = 10
    m = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid)
    lat = noise.clone().to("cuda:0")
    m.denoise(lat, m.encode_context(c1), m.encode_context(c2), m.embed_buffers(bl), FlowMatchScheduler(steps), 5.0)
    torch.cuda.synchronize()
    ref = R.denoise_loop(sdr, bsdr, cfg, noise, c1, c2, bl, num_steps=steps)
    p = R.psnr(lat.cpu(), ref)
    assert p >= 40.0, f"final-latent PSNR {p:.1f} dB < 40 dB"
    from psnr_util import frame_psnr
    pf = frame_psnr(lat.cpu(), ref)
    assert pf >= 40.0, f"decoded-frame PSNR (peak 255) {pf:.1f} dB < 40 dB"
    # the context-free stem shared between the two CFG forwards (product default) vs two full forwards: bit-identical
    m2 = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid)
    m2.share_stem = False
    lat_ns = noise.clone().to("cuda:0")
    m2.denoise(lat_ns, m2.encode_context(c1), m2.encode_context(c2), m2.embed_buffers(bl), FlowMatchScheduler(steps), 5.0)
    assert m.share_stem and torch.equal(lat_ns, lat), "sharing the stem between the CFG forwards changed the result"
    # determinism: same seed/buffers/prompt => bit-identical latents
    lat2 = noise.clone().to("cuda:0")
    m.denoise(lat2, m.encode_context(c1), m.encode_context(c2), m.embed_buffers(bl), FlowMatchScheduler(steps), 5.0)
    assert torch.equal(lat, lat2)


def test_hipgraph_replay_equals_eager_launches(hip_ops):
    """Launch-bound sizes replay each DiT forward as one hipGraph (WanDiT.prepare(graphs=...)): bit-identical to
    issuing the kernels one by one, across steps (the per-step scalars stay outside the graph) and CFG branches."""
    import time
    cfg, grid = preset("small"), TokenGrid(17, 128, 160)
    sd, bsd = syn.make_dit_state_dict(cfg), syn.make_buffer_embedder_state_dict(cfg)
    noise, c1, c2, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2), syn.make_buffer_latents(cfg, grid)
    res, times = {}, {}
    for mode in (False, True):
        m = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid, graphs=mode)
        assert m._graphs_on == mode
        ck, cu, bt = m.encode_context(c1), m.encode_context(c2), m.embed_buffers(bl)
        lat = noise.to("cuda:0")
        m.denoise(lat, ck, cu, bt, FlowMatchScheduler(3), 5.0)          # first steps: eager + capture
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.denoise(lat, ck, cu, bt, FlowMatchScheduler(8), 5.0)          # replays (same buffers -> same graphs)
        torch.cuda.synchronize()
        times[mode], res[mode] = time.perf_counter() - t0, lat.clone()
        if mode:
            assert len(m._graphs) == 2                                  # one graph per CFG branch
    print(f"8 steps, S={grid.S}: eager launches {times[False] * 1e3:.1f} ms, hipGraph replay {times[True] * 1e3:.1f} ms")
    assert torch.equal(res[False], res[True])
    # default off; "auto" = on for small token counts only
    assert not WanDiT(cfg, sd, hip_ops, bsd).prepare(grid)._graphs_on
    assert WanDiT(cfg, sd, hip_ops, bsd).prepare(grid, graphs="auto")._graphs_on and WanDiT.GRAPH_MAX_TOKENS < 37440


def test_dual_stream_cfg_equals_sequential(hip_ops, monkeypatch):
    """ICV_DUAL_STREAM=1: cond / uncond forwards on two HIP streams in twin engines sharing the weights — same latents."""
    cfg, grid = preset("small"), TokenGrid(17, 128, 160)
    sd, bsd = syn.make_dit_state_dict(cfg), syn.make_buffer_embedder_state_dict(cfg)
    noise, c1, c2, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2), syn.make_buffer_latents(cfg, grid)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("ICV_DUAL_STREAM", mode)
        m = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid)
        assert m.dual_stream == (mode == "1")
        lat = noise.to("cuda:0")
        m.denoise(lat, m.encode_context(c1), m.encode_context(c2), m.embed_buffers(bl), FlowMatchScheduler(5), 5.0)
        torch.cuda.synchronize()
        res[mode] = lat.clone()
    assert torch.equal(res["0"], res["1"])


def test_i2v_forward_and_loop_parity(hip_ops):
    """BASELINE.json config #5's image-conditioning branch at test size (in_dim 36, 257 CLIP tokens)."""
    cfg, grid = preset("tiny-i2v"), TokenGrid(9, 64, 96)
    sd, bsd = syn.make_dit_state_dict(cfg), syn.make_buffer_embedder_state_dict(cfg)
    sdr, bsdr = R.round_state_dict_to_bf16(sd), R.round_state_dict_to_bf16(bsd)
    noise, c1, c2 = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2)
    bl, clip, y = syn.make_buffer_latents(cfg, grid), syn.make_clip_features(cfg), syn.make_cond_latents(cfg, grid)
    m = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid)
    ck, cu = m.encode_context(c1, clip), m.encode_context(c2, clip)
    add = m.embed_cond_latents(y, add_to=m.embed_buffers(bl))
    lat = noise.to("cuda:0")
    m.forward_tokens(lat, ck, 731.0, add, m.head_out[0])
    torch.cuda.synchronize()
    v = R.unpatchify(m.head_out[0].cpu(), (grid.T, grid.Hp, grid.Wp), cfg.out_dim)
    vref = R.dit_forward(sdr, cfg, noise, c1, 731.0, R.buffer_embed(bsdr, bl), clip_fea=clip, y=y)
    rel = float((v - vref).norm() / vref.norm())
    cos = float(torch.nn.functional.cosine_similarity(v.flatten(), vref.flatten(), dim=0))
    assert cos >= 0.999 and rel <= 2e-2, f"i2v forward parity: cos={cos} rel-L2={rel}"
    steps = 6
    m.denoise(lat, ck, cu, add, FlowMatchScheduler(steps), 5.0)
    torch.cuda.synchronize()
    ref = R.denoise_loop(sdr, bsdr, cfg, noise, c1, c2, bl, num_steps=steps, clip_fea=clip, y=y)
    p = R.psnr(lat.cpu(), ref)
    assert p >= 40.0, f"i2v final-latent PSNR {p:.1f} dB < 40 dB"


def test_fp8_gemm_mode_forward_and_loop(hip_ops):
    """BASELINE.json config #5 at test size: i2v DiT with the six per-layer projections on the fp8 MFMA.
    Parity bar for the fp8 mode: against the oracle run with the SAME e4m3 row quantisation (kernel
    arithmetic), cosine >= 0.999 / rel-L2 <= 2e-2 per forward and PSNR >= 40 dB for the loop; the gap to
    the unquantised oracle is the quantisation itself and is only reported / bounded loosely."""
    cfg, grid = preset("tiny-i2v"), TokenGrid(9, 64, 96)
    sd, bsd = syn.make_dit_state_dict(cfg), syn.make_buffer_embedder_state_dict(cfg)
    sdr, bsdr = R.round_state_dict_to_bf16(sd), R.round_state_dict_to_bf16(bsd)
    noise, c1, c2 = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2)
    bl, clip, y = syn.make_buffer_latents(cfg, grid), syn.make_clip_features(cfg), syn.make_cond_latents(cfg, grid)
    m = WanDiT(cfg, sd, hip_ops, bsd, gemm_dtype="fp8", fp8_weights=WanDiT.FP8_WEIGHTS).prepare(grid)
    ck, cu = m.encode_context(c1, clip), m.encode_context(c2, clip)
    add = m.embed_cond_latents(y, add_to=m.embed_buffers(bl))
    lat = noise.to("cuda:0")
    m.forward_tokens(lat, ck, 731.0, add, m.head_out[0])
    torch.cuda.synchronize()
    v = R.unpatchify(m.head_out[0].cpu(), (grid.T, grid.Hp, grid.Wp), cfg.out_dim)
    kw = dict(clip_fea=clip, y=y)
    ref8 = R.dit_forward(sdr, cfg, noise, c1, 731.0, R.buffer_embed(bsdr, bl), fp8=True, **kw)
    ref = R.dit_forward(sdr, cfg, noise, c1, 731.0, R.buffer_embed(bsdr, bl), **kw)
    rel8 = float((v - ref8).norm() / ref8.norm())
    cos8 = float(torch.nn.functional.cosine_similarity(v.flatten(), ref8.flatten(), dim=0))
    rel = float((v - ref).norm() / ref.norm())
    print(f"fp8 forward: vs fake-quant oracle rel-L2 {rel8:.2e} cos {cos8:.6f}; vs unquantised oracle rel-L2 {rel:.2e}")
    assert cos8 >= 0.999 and rel8 <= 2e-2, f"fp8 forward vs fake-quant oracle: cos={cos8} rel-L2={rel8}"
    assert rel <= 0.15, f"fp8 forward vs unquantised oracle rel-L2 {rel}"
    steps = 6
    m.denoise(lat, ck, cu, add, FlowMatchScheduler(steps), 5.0)
    torch.cuda.synchronize()
    refl8 = R.denoise_loop(sdr, bsdr, cfg, noise, c1, c2, bl, num_steps=steps, fp8=True, **kw)
    refl = R.denoise_loop(sdr, bsdr, cfg, noise, c1, c2, bl, num_steps=steps, **kw)
    p8, p = R.psnr(lat.cpu(), refl8), R.psnr(lat.cpu(), refl)
    print(f"fp8 loop: PSNR vs fake-quant oracle {p8:.1f} dB, vs unquantised oracle {p:.1f} dB")
    assert p8 >= 40.0, f"fp8 loop PSNR vs fake-quant oracle {p8:.1f} dB < 40 dB"


@pytest.mark.parametrize("chunks,model,gemm_dtype", [(1, "tiny", "bf16"), (3, "tiny", "bf16"), (3, "tiny-i2v", "fp8"), (2, "tiny", "fp8+attn")])
def test_sequence_parallel_path_on_one_gpu(hip_ops, chunks, model, gemm_dtype):
    """The world>1 code path (token shards, RoPE offsets, chunked K/V gather feeding the carried-state
    attention kernel, per-shard Euler update) driven on ONE GPU: two shard engines run with a stand-in
    for the RCCL all-gather that serves the other shard's K/V rows from the unsharded run.
    Compared with the UNSHARDED HIP OUTPUT (a self-comparison: it proves the sharded schedule computes what the unsharded one
    does, not parity with the oracle - that is test_forward_parity / test_denoise_loop_psnr for the unsharded path, and
    tests/test_fullsize_gpu.py::test_layer_14b_sequence_parallel_shards_full_S for the shard shapes directly against the oracle)."""
    from infinicube_amd.videogen.seqpar import ShardPlan
    grid = TokenGrid(9, 64, 96)
    attn_dtype = "fp8" if gemm_dtype.endswith("+attn") else "bf16"      # 4th case: e4m3 self-attention, per-chunk K/V scales
    gemm_dtype = gemm_dtype.split("+")[0]
    cfg, sd, bsd, _, _ = _setup(model, grid)
    noise, ctx, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_buffer_latents(cfg, grid)
    clip = syn.make_clip_features(cfg) if cfg.has_image_input else None
    ycond = syn.make_cond_latents(cfg, grid) if cfg.has_image_input else None

    def additive(mm):
        bt = mm.embed_buffers(bl)
        return mm.embed_cond_latents(ycond, add_to=bt) if ycond is not None else bt

    rec = []
    raw = hip_ops.attention

    def recording_attention(q, k, v, o, heads, scale):
        if k.shape[0] == grid.S:
            rec.append((k.clone(), v.clone()))
        raw(q, k, v, o, heads, scale)

    hip_ops.attention = recording_attention
    raw8 = hip_ops.attention_fp8

    def recording_attention8(q, k, v, o, heads, ws):
        rec.append((k.clone(), v.clone()))
        raw8(q, k, v, o, heads, ws)

    hip_ops.attention_fp8 = recording_attention8
    try:
        full = WanDiT(cfg, sd, hip_ops, bsd, gemm_dtype=gemm_dtype, attn_dtype=attn_dtype, fp8_weights=WanDiT.FP8_WEIGHTS).prepare(grid, graphs=False)   # ops are wrapped: no capture
        lat = noise.to("cuda:0")
        full.forward_tokens(lat, full.encode_context(ctx, clip), 300.0, additive(full), full.head_out[0])
        torch.cuda.synchronize()
    finally:
        hip_ops.attention = raw
        hip_ops.attention_fp8 = raw8
    outs = []
    for r in range(2):
        plan = ShardPlan.make(grid.S, 2, r)
        n = plan.n_tok

        class FakeGather:
            def __init__(self):
                self.layer, self.calls, self.r0 = 0, 0, 0

            def start(self, rows, out):     # rows [m, 2d] = k | v of this chunk, out [world*m, 2d]
                dd = rows.shape[1] // 2
                k_rows, v_rows, k_out, v_out = rows[:, :dd], rows[:, dd:], out[:, :dd], out[:, dd:]
                kf, vf = rec[self.layer]
                m = k_rows.shape[0]
                r0 = self.r0
                mine = kf[plan.tok0 + r0: plan.tok0 + r0 + m]
                if self.layer == 0:   # before any chunked attention the shard's K is bit-identical
                    assert torch.equal(mine, k_rows), "local K rows differ from the unsharded run"
                else:                 # later layers differ by the chunked online-softmax rounding only
                    assert float((mine.float() - k_rows.float()).norm() / mine.float().norm()) < 1e-2
                k_out.copy_(torch.cat([kf[rk * n + r0: rk * n + r0 + m] for rk in range(2)], 0))
                v_out.copy_(torch.cat([vf[rk * n + r0: rk * n + r0 + m] for rk in range(2)], 0))
                self.calls += 1
                self.r0 += m
                if self.calls % chunks == 0:
                    self.layer += 1
                    self.r0 = 0
                return ()

            def wait(self, handle):
                pass

        m = WanDiT(cfg, sd, hip_ops, bsd, gemm_dtype=gemm_dtype, attn_dtype=attn_dtype, fp8_weights=WanDiT.FP8_WEIGHTS).prepare(grid, plan, kv_gather=FakeGather(), sp_chunks=chunks)
        m.forward_tokens(lat, m.encode_context(ctx, clip), 300.0, additive(m), m.head_out[0])
        torch.cuda.synchronize()
        outs.append(m.head_out[0].clone())
    got, want = torch.cat(outs, 0), full.head_out[0]
    rel = float((got - want).norm() / want.norm())
    # e4m3 attention: per-chunk K / V scales and a different P rounding reference -> fp8-level agreement
    assert rel < (6e-2 if attn_dtype == "fp8" else 5e-3), f"sharded vs unsharded forward rel-L2 {rel}"


def test_copy_engine_transport_one_rank_rehearsal(hip_ops):
    """KVGather mode "ipc" (csrc/ipc.hip) on the one rank a one-GPU process has: the symmetric heap is carved for the engine's
    and its CFG-pair twin's K|V rows, exported (hipMemGetAddressRange + hipIpcGetMemHandle of torch's allocation), the flag
    segment is created, registered and unlinked, the start-up self-test runs, and an 8-step CFG loop on the sequence-parallel
    schedule (publish -> own-rows copy -> wait, acquire before every K|V GEMM; 3 chunks x 2 branches x layers tickets through
    the 32-slot ring) is BIT-IDENTICAL to the same schedule on the collective transport.  The peers' half of the protocol is
    tests/test_multigpu_rccl.py::test_shared_gpu_copy_engine_transport_is_bit_identical_to_allgather."""
    import glob
    from infinicube_amd.videogen.seqpar import _IpcHeap
    grid = TokenGrid(9, 64, 96)
    cfg, sd, bsd, _, _ = _setup("tiny", grid)
    noise, c1, c2, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2), syn.make_buffer_latents(cfg, grid)
    lats = {}
    for mode in ("allgather", "ipc"):
        m = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid, force_sp=True, kv_exchange=mode, sp_chunks=3)
        assert m.sp_on and m.kv_gather.mode == mode
        lat = noise.clone().to("cuda:0")
        m.denoise(lat, m.encode_context(c1), m.encode_context(c2), m.embed_buffers(bl), FlowMatchScheduler(8), 5.0)
        torch.cuda.synchronize()
        lats[mode] = lat.clone()
        if mode == "ipc":
            heap = m.kv_gather._heap
            assert heap is not None and heap in _IpcHeap._live
            assert m.kv_loc.data_ptr() >= heap.mem.data_ptr() and m._pair is not None and m._pair.kv_loc.data_ptr() > m.kv_loc.data_ptr()
            n_tickets = hip_ops.lib.icv_ipc_tickets(heap.handle)
            assert n_tickets == 2 + m.kv_gather.n_collectives, "two self-test exchanges + one ticket per chunk exchange"
            assert n_tickets > 2 * 32, "the loop must wrap the flag-slot ring at least twice"
            assert not glob.glob("/dev/shm/icv_kv_*"), "the flag segment's name must not outlive the set-up"
            m.kv_gather.close()
            assert heap not in _IpcHeap._live
    assert torch.isfinite(lats["ipc"]).all() and torch.equal(lats["ipc"], lats["allgather"])


@pytest.mark.parametrize("mode", ["allgather+arrival", "ipc+arrival"])
def test_arrival_driven_attention_one_rank_rehearsal(hip_ops, mode):
    """The arrival-driven self-attention (csrc/attn7p.hip: ONE launch per layer over the K|V pieces) on the one rank a one-GPU
    process has: the sequence-parallel schedule with the engine's own rows as the only piece.  One piece of whole tiles in memory
    order is the plain kernel's tile sequence, so a 12-step CFG loop must be BIT-IDENTICAL to the same schedule on the chunked
    carried-state launches with ONE chunk (a single launch over the same rows; 3 chunks re-associate the fp32 sums).  With
    "ipc": the transport runs without its own-rows copy (icv_ipc_configure), tickets are released by icv_ipc_gather_consumed
    (the 32-slot ring wraps), and no device-side wait may have given up.  The peers' half: tests/test_multigpu_rccl.py."""
    grid = TokenGrid(9, 64, 96)
    cfg, sd, bsd, _, _ = _setup("tiny", grid)
    noise, c1, c2, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2), syn.make_buffer_latents(cfg, grid)
    lats = {}
    for kv, chunks in ((mode.split("+")[0], 1), (mode, 3)):
        m = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid, force_sp=True, kv_exchange=kv, sp_chunks=chunks)
        assert m.sp_on and m.attn_arrival == kv.endswith("+arrival") and m.kv_gather.mode == kv.split("+")[0]
        lat = noise.clone().to("cuda:0")
        m.denoise(lat, m.encode_context(c1), m.encode_context(c2), m.embed_buffers(bl), FlowMatchScheduler(12), 5.0)
        torch.cuda.synchronize()
        lats[kv] = lat.clone()
        if m.attn_arrival:
            assert int(m.sp_err.item()) == 0
            m.check_exchange()
            if kv.startswith("ipc"):
                assert hip_ops.lib.icv_ipc_tickets(m.kv_gather._heap.handle) > 32, "the loop must wrap the flag-slot ring"
        m.kv_gather.close()
    a, b = lats[mode.split("+")[0]], lats[mode]
    assert torch.isfinite(b).all() and torch.equal(a, b), f"arrival-driven vs one chunked launch: max |d| {float((a - b).abs().max())}"


@pytest.mark.parametrize("mode", ["ipc+arrival", "allgather+arrival"])
def test_arrival_gated_e4m3_chunks_one_rank_rehearsal(hip_ops, mode):
    """The e4m3 mode of the arrival-driven schedule on one rank: every chunk launch gates on its blobs inside the kernel
    (icv_attention_fp8_fwd_pieces_gated) and reads this rank's own blob where it was quantised.  One rank = one piece per chunk in the
    same order, so the loop must be BIT-IDENTICAL to the host-waited chunk launches when both arms cut the same chunk bounds (the gated
    arm cuts on the 64-key grid); otherwise the two differ by the re-association of the fp32 sums only."""
    grid = TokenGrid(9, 64, 96)
    cfg, sd, bsd, _, _ = _setup("tiny", grid)
    noise, c1, c2, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2), syn.make_buffer_latents(cfg, grid)
    lats = {}
    for kv in (mode.split("+")[0], mode):
        m = WanDiT(cfg, sd, hip_ops, bsd, gemm_dtype="fp8", attn_dtype="fp8").prepare(grid, force_sp=True, kv_exchange=kv, sp_chunks=3)
        assert m.sp_on and m.fp8_wire and m.attn_arrival == kv.endswith("+arrival")
        lat = noise.clone().to("cuda:0")
        m.denoise(lat, m.encode_context(c1), m.encode_context(c2), m.embed_buffers(bl), FlowMatchScheduler(12), 5.0)
        torch.cuda.synchronize()
        lats[kv] = (lat.clone(), list(m.sp_bounds))
        if m.attn_arrival:
            assert int(m.sp_err.item()) == 0
            m.check_exchange()
        m.kv_gather.close()
    (a, ba), (b, bb) = lats[mode.split("+")[0]], lats[mode]
    assert torch.isfinite(b).all()
    if ba == bb:
        assert torch.equal(a, b), f"gated vs host-waited chunk launches: max |d| {float((a - b).abs().max())}"
    else:                                          # different chunk bounds re-associate the fp32 sums
        assert R.psnr(b.cpu(), a.cpu()) >= 50.0, f"gated vs host-waited chunk launches (other chunk bounds {bb} vs {ba}): {R.psnr(b.cpu(), a.cpu()):.1f} dB"


def test_copy_engine_transport_refuses_misuse(hip_ops):
    """icv_ipc_* error paths on one rank: rows outside the symmetric heap, a heap that is too small for what is carved from it,
    more than ICV_IPC_SLOTS exchanges without a wait, a wait for a ticket that is not in flight - each a clean error, not a hang."""
    import ctypes
    from infinicube_amd import native
    from infinicube_amd.videogen.seqpar import KVGather, ShardPlan
    kg = KVGather(ShardPlan.make(64, 1, 0), None, "ipc")
    kg.reserve(1 << 20, "cuda:0")
    heap = kg._heap
    rows = kg.local_rows(64, 256, torch.bfloat16, hip_ops.alloc)
    out = torch.empty_like(rows)
    with pytest.raises(ValueError, match="not inside the symmetric heap"):
        kg.start(torch.zeros_like(rows), out)
    with pytest.raises(RuntimeError, match="heap exhausted"):
        kg.local_rows(1 << 20, 256, torch.bfloat16, hip_ops.alloc)
    base = hip_ops.lib.icv_ipc_tickets(heap.handle)
    handles = [kg.start(rows, out) for _ in range(native.IPC_SLOTS - (base % native.IPC_SLOTS))]       # fill what is left of the ring ...
    handles += [kg.start(rows, out) for _ in range(base % native.IPC_SLOTS)]                           # ... exactly IPC_SLOTS un-waited tickets
    assert len(handles) == native.IPC_SLOTS
    with pytest.raises(native.NativeError, match="exchanges in flight without a wait"):
        kg.start(rows, out)
    for h in handles:
        kg.wait(h)
    kg.wait(kg.start(rows, out))                     # the ring is free again
    with pytest.raises(native.NativeError, match="not in flight"):
        native.check(hip_ops.lib.icv_ipc_gather_wait(heap.handle, 10 ** 6, torch.cuda.current_stream().cuda_stream), "icv_ipc_gather_wait")
    torch.cuda.synchronize()
    assert torch.equal(out, rows)
    kg.close()


@pytest.mark.parametrize("transport", ["torch", "native"])
def test_sequence_parallel_gather_on_rccl_stream(hip_ops, transport):
    """transport = "torch": torch.distributed's collective; "native": libicvideo's own RCCL communicator and side stream
    (icv_comm_unique_id / icv_comm_create / icv_allgather_kv, seqpar._NativeComm), id shipped through the nccl group.
    The same sharded path, but each shard's OWN K/V rows travel through a real RCCL collective
    (`nccl` backend, a one-rank group: the only RCCL this one-GPU box can run) issued async on RCCL's
    stream while the compute stream keeps projecting Q; peer rows come from the unsharded run.  Checks
    the stream hand-off KVGather relies on: the collective must see K/V written by earlier kernels on
    the compute stream, and the chunk attention must not start before `wait()`."""
    import torch.distributed as dist
    from infinicube_amd.videogen.seqpar import ShardPlan
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29571", world_size=1, rank=0,
                                device_id=torch.device("cuda", 0))
    grid = TokenGrid(9, 64, 96)
    cfg, sd, bsd, _, _ = _setup("tiny", grid)
    noise, ctx, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_buffer_latents(cfg, grid)
    rec = []
    raw = hip_ops.attention

    def recording_attention(q, k, v, o, heads, scale):
        if k.shape[0] == grid.S:
            rec.append((k.clone(), v.clone()))
        raw(q, k, v, o, heads, scale)

    hip_ops.attention = recording_attention
    try:
        full = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid, graphs=False)   # ops are wrapped: no capture
        lat = noise.to("cuda:0")
        full.forward_tokens(lat, full.encode_context(ctx), 300.0, full.embed_buffers(bl), full.head_out[0])
        torch.cuda.synchronize()
    finally:
        hip_ops.attention = raw
    chunks, outs = 3, []
    from infinicube_amd.videogen.seqpar import _NativeComm
    ncomm = _NativeComm(dist, None, [0], 0, 1) if transport == "native" else None
    for r in range(2):
        plan = ShardPlan.make(grid.S, 2, r)
        n = plan.n_tok

        class RcclOwnRows:
            def __init__(self):
                self.layer, self.calls, self.r0 = 0, 0, 0

            def start(self, rows, out):     # rows [m, 2d] = k | v of this chunk, out [world*m, 2d]
                dd = rows.shape[1] // 2
                kf, vf = rec[self.layer]
                m, r0, peer = rows.shape[0], self.r0, 1 - r
                out[peer * m:(peer + 1) * m, :dd].copy_(kf[peer * n + r0: peer * n + r0 + m])
                out[peer * m:(peer + 1) * m, dd:].copy_(vf[peer * n + r0: peer * n + r0 + m])
                if ncomm is not None:
                    h = (ncomm.allgather(rows, out[r * m:(r + 1) * m]),)
                else:
                    h = (dist.all_gather_into_tensor(out[r * m:(r + 1) * m], rows, async_op=True),)
                self.calls += 1
                self.r0 += m
                if self.calls % chunks == 0:
                    self.layer, self.r0 = self.layer + 1, 0
                return h

            def wait(self, handle):
                for w in handle:
                    w.wait()

        m = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid, plan, kv_gather=RcclOwnRows(), sp_chunks=chunks)
        m.forward_tokens(lat, m.encode_context(ctx), 300.0, m.embed_buffers(bl), m.head_out[0])
        torch.cuda.synchronize()
        outs.append(m.head_out[0].clone())
    got, want = torch.cat(outs, 0), full.head_out[0]
    rel = float((got - want).norm() / want.norm())
    assert rel < 5e-3, f"sharded (RCCL own-rows) vs unsharded forward rel-L2 {rel}"



def test_config1_wan_1p3b_17f_256p_10_steps(hip_ops):
    """BASELINE.json config #1 at the REAL Wan2.1-1.3B dimensions (d=1536, 12 heads, 30 layers, text 512x4096):
    17 frames 256x448 (S = 2240 tokens), 10 flow-match steps with CFG, guidance-buffer tokens from the dummy
    buffers' stand-in latents; HIP loop vs the fp32 oracle fed the same bf16-rounded weights (run by stock PyTorch on the GPU,
    cross-checked against the CPU execution of the same code on a full CFG step of the same model on a smaller grid).
    Bar: final-latent PSNR >= 40 dB (north star), per-step velocity cosine >= 0.999."""
    import time
    from infinicube_amd.videogen.config import GRID_CFG1
    cfg, grid = preset("1.3b"), GRID_CFG1
    sd = syn.make_dit_state_dict(cfg, seed=0, dtype=torch.bfloat16)
    bsd = syn.make_buffer_embedder_state_dict(cfg, dtype=torch.bfloat16)
    sdr = {k: v.float() for k, v in sd.items()}
    bsdr = {k: v.float() for k, v in bsd.items()}
    noise = syn.make_latent_noise(grid)
    c1, c2, bl = syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2), syn.make_buffer_latents(cfg, grid)
    steps = 10
    m = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid)
    lat = noise.clone().to("cuda:0")
    t0 = time.time()
    m.denoise(lat, m.encode_context(c1), m.encode_context(c2), m.embed_buffers(bl), FlowMatchScheduler(steps), 5.0)
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    # the checker: oracle/wan_ref.py's loop executed in fp32 by stock PyTorch ON THE GPU (seconds instead of two minutes on
    # the host cores), tied to the same code on the CPU by one full CFG step (identical function, different BLAS)
    dev = "cuda:0"
    t0 = time.time()
    ref = R.denoise_loop({k: v.to(dev) for k, v in sdr.items()}, {k: v.to(dev) for k, v in bsdr.items()}, cfg, noise.to(dev), c1.to(dev), c2.to(dev),
                         bl.to(dev), num_steps=steps).cpu()
    # ... on a SMALL grid (5 latent frames of 8 x 12 tokens: the same 30 layers, 1.3B widths and code path; one full CFG step takes
    # about a second on the host instead of twenty at S = 2240)
    from infinicube_amd.videogen.config import TokenGrid
    g_small = TokenGrid(17, 128, 192)
    n_small, bl_small = syn.make_latent_noise(g_small), syn.make_buffer_latents(cfg, g_small)
    one_gpu = R.denoise_loop({k: v.to(dev) for k, v in sdr.items()}, {k: v.to(dev) for k, v in bsdr.items()}, cfg, n_small.to(dev), c1.to(dev), c2.to(dev),
                             bl_small.to(dev), num_steps=1).cpu()
    torch.cuda.synchronize()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    one_cpu = R.denoise_loop(sdr, bsdr, cfg, n_small, c1, c2, bl_small, num_steps=1)
    t_cpu = time.time() - t0
    assert float((one_gpu - one_cpu).norm() / one_cpu.norm()) < 1e-4, "the oracle on GPU tensors and on CPU tensors disagree"
    # ... and AT THE FULL GRID against the CPU path itself (round 6; ADVICE r5 flagged the small-grid tie as a weakening): the final latent
    # of this very config from oracle/wan_ref.denoise_loop on the CPU (226 s on 8 Xeon threads; tools/cpu_config1_full.py --save-golden,
    # profiles/r06/cpu_config1_full.txt) is a committed fixture - config #1 is "the reference's own CPU-runnable case" (BASELINE.json)
    import numpy as np
    gold_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config1_cpu_oracle_latent.npz")
    gold = np.load(gold_path)
    assert int(gold["steps"]) == steps and np.array_equal(gold["noise_head"], noise.flatten()[:16].numpy()), "the golden latent was made from other inputs"
    cpu_ref = torch.from_numpy(gold["latent"])
    rel_cpu = float((ref - cpu_ref).norm() / cpu_ref.norm())
    p_cpu = R.psnr(lat.cpu(), cpu_ref)
    print(f"config #1, full grid: oracle on the GPU vs the committed CPU-oracle latent rel-L2 {rel_cpu:.3g}; HIP loop vs the CPU-oracle latent {p_cpu:.1f} dB")
    assert rel_cpu < 1e-4, f"the GPU-executed oracle drifted from the CPU path over 10 steps: rel-L2 {rel_cpu}"
    assert p_cpu >= 40.0, f"config #1 vs the CPU path: {p_cpu:.1f} dB"
    p = R.psnr(lat.cpu(), ref)
    cos = float(torch.nn.functional.cosine_similarity((lat.cpu() - noise).flatten(), (ref - noise).flatten(), dim=0))
    from psnr_util import frame_psnr
    pf = frame_psnr(lat.cpu(), ref)
    print(f"config #1: HIP {t_gpu:.2f}s, oracle (fp32 torch on the GPU, 10 steps + one step on the CPU) {t_cpu:.1f}s, latent PSNR {p:.1f} dB, decoded-frame PSNR {pf:.1f} dB, update cosine {cos:.5f}")
    assert p >= 40.0 and pf >= 40.0 and cos >= 0.999, f"config #1 parity: latent {p:.1f} dB, frames {pf:.1f} dB, cosine {cos}"
    # The fp8 mode (what torch_dtype=float8_e4m3fn selects: e4m3 self-attention + the DEFAULT e4m3 projection set = QKV) at
    # the same REAL depth must also meet the 40 dB bar against the UNQUANTISED fp32 oracle (14B depth: test_fullsize_gpu.py).
    del m

    def run(**kw):
        mm = WanDiT(cfg, sd, hip_ops, bsd, **kw).prepare(grid)
        l8 = noise.clone().to("cuda:0")
        mm.denoise(l8, mm.encode_context(c1), mm.encode_context(c2), mm.embed_buffers(bl), FlowMatchScheduler(steps), 5.0)
        torch.cuda.synchronize()
        l8 = l8.cpu()
        return R.psnr(l8, ref), float(torch.nn.functional.cosine_similarity((l8 - noise).flatten().double(), (ref - noise).flatten().double(), dim=0))

    assert WanDiT.FP8_DEFAULT == ("wqkv",) == R.FP8_DEFAULT
    pd, cd = run(gemm_dtype="fp8", attn_dtype="fp8")
    print(f"config #1, fp8 mode (default set {WanDiT.FP8_DEFAULT} + e4m3 self-attention): PSNR vs unquantised fp32 oracle {pd:.1f} dB, cosine {cd:.5f}")
    assert pd >= 40.0 and cd >= 0.995, f"the default fp8 mode misses the 40 dB bar at 1.3B depth: {pd:.1f} dB, cosine {cd}"
    # reported (and loosely bounded): every projection in e4m3, and the e4m3 self-attention alone
    p6, c6 = run(gemm_dtype="fp8", attn_dtype="fp8", fp8_weights=WanDiT.FP8_WEIGHTS)
    print(f"config #1, all six projections + self-attention in e4m3: {p6:.1f} dB, cosine {c6:.5f}")
    assert p6 >= 25.0 and c6 >= 0.98
    pa, _ = run(attn_dtype="fp8")
    print(f"config #1, bf16 projections + fp8 self-attention: {pa:.1f} dB")
    assert pa >= 50.0


def test_generator_end_to_end_on_gpu(tmp_path, monkeypatch):
    """The whole drop-in path on the real kernels: WanVideoGenerator(ckpt) -> checkpoint overlay ->
    generate(uint8 buffers) -> WanVideoPipeline -> WanDiT/HipOps -> frames (+ mp4 hook), with the stand-in
    text encoder / VAE (no checkpoints exist offline).  Checks determinism and buffer conditioning."""
    import contextlib
    import io
    from safetensors.torch import save_file
    import infinicube_amd.videogen.inference as inf
    from infinicube.videogen import WanVideoGenerator
    from infinicube_amd.videogen.pipeline import DiTHolder, WanVideoPipeline
    from standins import HashTextEncoder, PoolVAE
    cfg, grid = preset("tiny"), TokenGrid(9, 64, 96)
    sd, bsd = syn.make_dit_state_dict(cfg), syn.make_buffer_embedder_state_dict(cfg)
    path = str(tmp_path / "step-1.safetensors")
    save_file({**{"buffer_embedder." + k: v for k, v in bsd.items()}, "dit.head.modulation": sd["head.modulation"] * 1.1}, path)

    def factory(torch_dtype, device, model_configs):
        pipe = WanVideoPipeline(device, torch_dtype, DiTHolder(sd, cfg), HashTextEncoder(cfg), PoolVAE())
        pipe.num_inference_steps = 3
        return pipe

    saved = {}
    monkeypatch.setattr(inf, "save_video", lambda fr, p, fps, quality: saved.update(n=len(fr), p=p))
    with contextlib.redirect_stdout(io.StringIO()):
        g = WanVideoGenerator(path, device="cuda:0", use_wan_1pt3b=True, pipeline_factory=factory)
        sem, co = syn.make_dummy_buffers(grid)
        f1 = g.generate(sem, co, seed=0, output_path=str(tmp_path / "video_480p_front.mp4"))
        f2 = g.generate(sem, co, seed=0)
        sem2 = sem.copy(); sem2[:, :32] = 20
        f3 = g.generate(sem2, co, seed=0)
    assert len(f1) == 9 and f1[0].size == (96, 64) and saved["n"] == 9
    import numpy as np
    a1, a2, a3 = (np.stack([np.asarray(x) for x in f]) for f in (f1, f2, f3))
    assert np.array_equal(a1, a2), "same seed + buffers + prompt must reproduce the same frames"
    assert not np.array_equal(a1, a3), "guidance buffers must condition the output"


def test_pipeline_i2v_with_the_real_vae_and_clip_modules_on_gpu(monkeypatch):
    """The pipeline end to end on the GPU with the REAL module classes either side of the loop at small widths: the Wan-VAE
    (tiled encode of both guidance buffers and of the image-conditioning clip, tiled decode - every convolution on libicvideo's
    kernel) and the CLIP vision tower (patch embedding as a matmul), the image-to-video branch of the DiT in between.  Same call
    twice = the same frames; the same call with the VAE's convolutions on MIOpen = the same video to bf16 rounding through a
    2-step loop."""
    from PIL import Image
    import numpy as np
    from infinicube_amd.videogen.clip_vision import ClipVisionEncoder
    from infinicube_amd.videogen.pipeline import DiTHolder, WanVideoPipeline
    from infinicube_amd.videogen.vae import WanVAE, WanVAENet
    from standins import HashTextEncoder
    cfg, grid = preset("tiny-i2v"), TokenGrid(9, 64, 96)
    sd = syn.make_dit_state_dict(cfg)
    torch.manual_seed(21)
    net = WanVAENet(dim=32, z_dim=16).eval()
    clip = ClipVisionEncoder(image_size=224, patch=14, dim=cfg.img_dim, heads=2, layers=2, use_blocks=1).to("cuda:0", torch.bfloat16).eval()
    rng = np.random.default_rng(0)
    img = Image.fromarray(rng.integers(0, 255, (grid.height, grid.width, 3), dtype=np.uint8), mode="RGB")
    sem, co = syn.make_dummy_buffers(grid)
    to_pil = lambda a: [Image.fromarray(f, mode="RGB") for f in a]      # noqa: E731

    def run(conv):
        import copy
        monkeypatch.setenv("ICV_VAE_CONV", conv)
        vae = WanVAE(copy.deepcopy(net), "cuda:0", torch.bfloat16)
        assert (vae.hip is not None) == (conv == "hip")
        pipe = WanVideoPipeline("cuda:0", torch.bfloat16, DiTHolder(sd, cfg), HashTextEncoder(cfg), vae, image_encoder=clip)
        pipe.initialize_buffer_embedder(16, zero_init=False)
        out = pipe(prompt="a street", negative_prompt="bad", semantic_buffer_video=to_pil(sem), coordinate_buffer_video=to_pil(co), input_image=img,
                   height=grid.height, width=grid.width, num_frames=grid.num_frames, seed=0, num_inference_steps=2, tiled=True,
                   tile_size=(6, 8), tile_stride=(3, 4))
        return np.stack([np.asarray(f) for f in out]).astype(np.int16)

    a, a2, b = run("hip"), run("hip"), run("miopen")
    assert a.shape == (grid.num_frames, grid.height, grid.width, 3)
    assert np.array_equal(a, a2), "the same call must reproduce the same frames"
    d = np.abs(a - b)
    print(f"i2v pipeline with the real VAE / CLIP: HIP vs MIOpen convolutions: mean |d| {d.mean():.3f}, max {d.max()} of 255")
    assert d.mean() < 2.0, f"HIP-convolution pipeline drifts from the MIOpen one: mean |d| {d.mean():.2f}"


@pytest.mark.parametrize("name,mode", [("tiny", "bf16"), ("tiny-i2v", "bf16"), ("tiny", "fp8"), ("tiny-i2v", "fp8"),
                                       ("tiny", "sp"), ("tiny", "sp-torch"), ("tiny-i2v", "sp+fp8")])
def test_native_forward_matches_python_driver(hip_ops, name, mode, monkeypatch):
    """icv_dit_create / icv_dit_bind / icv_dit_forward (the whole forward enqueued by ONE C call) against the per-op
    driver in dit.py: the same launchers in the same order, so the velocity tokens and a whole CFG denoise loop are
    bit-identical, with and without the shared stem; binding errors are reported through the ABI.
    Modes: bf16; fp8 = e4m3 projections (icv_dit_bind "<name>_s") + e4m3 self-attention (icv_dit_set_fp8); sp = the
    sequence-parallel schedule (icv_dit_set_seqpar: K|V rows through icv_allgather_kv on libicvideo's own RCCL
    communicator and side stream) rehearsed on ONE rank — the only RCCL world a 1-GPU box can build; the N-rank form runs
    in tests/test_multigpu_rccl.py when the box has the GPUs; sp-torch = the Python arm exchanges through its local-copy
    path while the C arm uses RCCL."""
    import ctypes
    from infinicube_amd import native
    cfg, grid = preset(name), TokenGrid(9, 64, 96)
    sd, bsd = syn.make_dit_state_dict(cfg), syn.make_buffer_embedder_state_dict(cfg)
    noise, c1, c2 = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2)
    bl = syn.make_buffer_latents(cfg, grid)
    clip = syn.make_clip_features(cfg) if cfg.has_image_input else None
    y = syn.make_cond_latents(cfg, grid) if cfg.has_image_input else None
    fp8, sp = "fp8" in mode, mode.startswith("sp")
    kw = dict(gemm_dtype="fp8", attn_dtype="fp8") if fp8 else {}
    if fp8 and sp:
        # like with like: the C driver gathers bf16 rows and quantises each gathered chunk; the per-op driver's default since
        # round 5 is e4m3 ON THE WIRE (quantised once with group-wide scales), which the C driver does not implement - and refuses
        eng = WanDiT(cfg, sd, hip_ops, bsd, **kw).prepare(grid, graphs=False, force_sp=True, sp_chunks=3, kv_exchange="native")
        eng.native_forward = True
        assert eng.fp8_wire and not eng._native_eligible(), "the C driver must stand down for the e4m3 wire format"
        eng.kv_gather.close()
        monkeypatch.setenv("ICV_FP8_WIRE", "bf16")
    outs = {}
    for native_on in (False, True):
        m = WanDiT(cfg, sd, hip_ops, bsd, **kw).prepare(grid, graphs=False, force_sp=sp, sp_chunks=3,
                                                       kv_exchange=("allgather" if mode == "sp-torch" else "native") if sp else None)
        m.native_forward = native_on
        assert m._native_eligible() == native_on and m.sp_on == sp
        ck, cu = m.encode_context(c1, clip), m.encode_context(c2, clip)
        add = m.embed_buffers(bl)
        if y is not None:
            add = m.embed_cond_latents(y, add_to=add)
        lat = noise.clone().to("cuda:0")
        m.forward_tokens(lat, ck, 731.0, add, m.head_out[0])
        m.forward_tokens(lat, ck, 731.0, add, m.head_out[1], num_layers=1)
        one = m.head_out.clone()
        m.denoise(lat, ck, cu, add, FlowMatchScheduler(3), 5.0)          # stem shared between the CFG forwards
        torch.cuda.synchronize()
        outs[native_on] = (one.cpu(), lat.cpu())
        if sp and not native_on:
            assert m.kv_gather.n_collectives > 0
        if native_on and mode == "bf16":
            h = m._native_ctx()
            m.native_profile(True)                   # event pairs around the self-attention launches, read back as (ms, count)
            m.forward_tokens(lat, ck, 500.0, add, m.head_out[0])
            ms, n = m.native_profile_read()
            assert n == cfg.num_layers and 0.0 < ms < 1e3
            assert m.native_profile_read() == (0.0, 0)
            m.native_profile(False)
            assert hip_ops.lib.icv_dit_bind(h, b"no_such_tensor", -1, lat.data_ptr()) != 0
            assert b"unknown tensor" in hip_ops.lib.icv_last_error()
            assert hip_ops.lib.icv_dit_bind(h, b"wqkv", cfg.num_layers, lat.data_ptr()) != 0
        if native_on and mode == "sp":               # under the sequence-parallel schedule every KEY-CHUNK launch is timed
            m.native_profile(True)
            m.forward_tokens(lat, ck, 500.0, add, m.head_out[0])
            ms, n = m.native_profile_read()
            assert n == cfg.num_layers * 3 and 0.0 < ms < 1e3, (ms, n)
            m.native_profile(False)
    assert torch.isfinite(outs[True][1]).all()
    assert torch.equal(outs[True][0], outs[False][0]), "icv_dit_forward differs from the per-op driver (one forward)"
    assert torch.equal(outs[True][1], outs[False][1]), "icv_dit_forward differs from the per-op driver (CFG loop, shared stem)"
    if sp and not fp8:     # the rehearsed schedule against the plain single-rank one: chunked softmax merge order only
        m = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid, graphs=False)
        lat = noise.clone().to("cuda:0")
        m.denoise(lat, m.encode_context(c1, clip), m.encode_context(c2, clip), m.embed_buffers(bl), FlowMatchScheduler(3), 5.0)
        ref = lat.cpu()
        assert float((outs[True][1] - ref).norm() / ref.norm()) < 2e-3
    bad = native.DitConfig(dim=100, ffn_dim=64, heads=1, layers=1, n_tok=4, tok0=0, T=1, Hp=2, Wp=2, k_patch=64, out_cols=64, eps=1e-6)
    hh = ctypes.c_void_p()
    assert hip_ops.lib.icv_dit_create(ctypes.byref(bad), ctypes.byref(hh)) != 0


@pytest.mark.parametrize("name,mode", [("tiny", "bf16"), ("small", "bf16"), ("tiny-i2v", "bf16"), ("tiny", "fp8"), ("tiny-i2v", "fp8"),
                                       ("small", "bf16+sp"), ("tiny", "fp8+sp")])
def test_cfg_batched_forward_pair_is_bit_identical(hip_ops, name, mode):
    """The two CFG forwards of a step as ONE batch of 2n rows (WanDiT.forward_pair, the single-rank default) against two
    sequential forwards on the HIP kernels: every token-local kernel computes a row independently of the row count and the
    branch-specific launches (RoPE, self-attention, cross-attention) see the same operands, so a whole CFG loop is
    bit-identical — with and without the shared stem.  "+sp": the same under the sequence-parallel schedule on one rank
    (force_sp: what every rank of the `sp` layout runs - projections over 2n rows, exchange + chunked attention per branch)."""
    sp = mode.endswith("+sp")
    mode = mode.split("+")[0]
    cfg, grid = preset(name), (TokenGrid(17, 128, 160) if name == "small" else TokenGrid(9, 64, 96))
    sd, bsd = syn.make_dit_state_dict(cfg), syn.make_buffer_embedder_state_dict(cfg)
    noise, c1, c2, bl = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2), syn.make_buffer_latents(cfg, grid)
    clip = syn.make_clip_features(cfg) if cfg.has_image_input else None
    y = syn.make_cond_latents(cfg, grid) if cfg.has_image_input else None
    kw = dict(gemm_dtype="fp8", attn_dtype="fp8", fp8_weights=WanDiT.FP8_WEIGHTS) if mode == "fp8" else {}
    res = {}
    for batch in (False, True):
        for share in (False, True):
            m = WanDiT(cfg, sd, hip_ops, bsd, **kw).prepare(grid, graphs=False, force_sp=sp, sp_chunks=3)
            m.cfg_batch, m.share_stem = batch, share
            add = m.embed_buffers(bl)
            if y is not None:
                add = m.embed_cond_latents(y, add_to=add)
            lat = noise.clone().to("cuda:0")
            m.denoise(lat, m.encode_context(c1, clip), m.encode_context(c2, clip), add, FlowMatchScheduler(3), 5.0)
            torch.cuda.synchronize()
            assert (m._pair is not None) == batch and m.sp_on == sp
            res[(batch, share)] = lat.cpu()
    assert torch.isfinite(res[(True, False)]).all()
    for k, v in res.items():
        assert torch.equal(v, res[(False, False)]), f"cfg_batch / share_stem = {k} changes the result"


def test_native_forward_under_graph_replay(hip_ops):
    """icv_dit_forward inside hipGraph capture (graphs=True: every forward after the first is a replay): the C driver only
    enqueues on the capture stream, so the replayed loop equals the eager per-op loop bit for bit."""
    cfg, grid = preset("tiny"), TokenGrid(9, 64, 96)
    sd, bsd = syn.make_dit_state_dict(cfg), syn.make_buffer_embedder_state_dict(cfg)
    noise, c1, c2 = syn.make_latent_noise(grid), syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2)
    bl = syn.make_buffer_latents(cfg, grid)
    outs = []
    for native_on, graphs in ((False, False), (True, True)):
        m = WanDiT(cfg, sd, hip_ops, bsd).prepare(grid, graphs=graphs)
        m.native_forward = native_on
        ck, cu, add = m.encode_context(c1), m.encode_context(c2), m.embed_buffers(bl)
        lat = noise.clone().to("cuda:0")
        m.denoise(lat, ck, cu, add, FlowMatchScheduler(4), 5.0)
        torch.cuda.synchronize()
        if graphs:
            assert m._graphs_on and len(m._graphs) >= 1
        outs.append(lat.cpu())
    assert torch.equal(outs[0], outs[1])
