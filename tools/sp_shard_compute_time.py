"""Compute-only projection of the N-GPU step time from ONE GPU (run on the GPU box).

A rank of an N-GPU run does, per layer, exactly the kernels of `WanDiT._forward_body` on its token shard with the K|V rows of
the other ranks arriving from the exchange.  Everything except the exchange can be timed on one GPU: this tool builds a
few Wan2.1-14B layers at S = 37 440, runs the sequence-parallel schedule of one rank of a `world`-rank group with the
exchange SERVED from local memory (a device copy of random rows stands in for the arrived peers' rows; no transfer time),
and reports ms per layer-forward -> the step time a rank would need if every transfer hid completely under compute:

    layout at N GPUs        forwards per rank per step     shard
    cfg+sp  N = 2           1                              world 1 (no exchange at all)
    cfg+sp  N = 4           1                              world 2
    cfg+sp  N = 8           1                              world 4
    sp      N = 8           2                              world 8

It is a LOWER bound on the step time and an upper bound on the scaling: RCCL's kernels take CUs while they run and whatever
part of a transfer is not hidden adds to it.  The first real measurement is the driver's 8-GPU run; `bench.py --gpus N` reports
`exposed_kv_wait_ms_per_step` to compare with this table.
"""
import dataclasses
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infinicube_amd.videogen import synthetic as syn
from infinicube_amd.videogen.config import GRID_480P, GRID_720P, preset
from infinicube_amd.videogen.dit import WanDiT
from infinicube_amd.videogen.ops import HipOps
from infinicube_amd.videogen.seqpar import ShardPlan

DEV = "cuda:0"
LAYERS = int(os.environ.get("LAYERS", "4"))
ops = HipOps(DEV)
# MODEL=14b|14b-i2v  GRID=480p|720p  GEMM=bf16|fp8  ATTN=bf16|fp8 (with ICV_FP8_WIRE=e4m3|bf16: what travels in the fp8 mode)
MODEL, GEMM, ATTN = os.environ.get("MODEL", "14b"), os.environ.get("GEMM", "bf16"), os.environ.get("ATTN", "bf16")
cfg_full = preset(MODEL)
cfg, grid = dataclasses.replace(cfg_full, num_layers=LAYERS), (GRID_720P if os.environ.get("GRID", "480p") == "720p" else GRID_480P)
sd = syn.make_dit_state_dict(cfg, seed=0, device=DEV, dtype=torch.bfloat16)
bsd = syn.make_buffer_embedder_state_dict(cfg, device=DEV, dtype=torch.bfloat16)
noise, ctx, bl = syn.make_latent_noise(grid).to(DEV), syn.make_text_context(cfg, 1), syn.make_buffer_latents(cfg, grid)
ctx2 = syn.make_text_context(cfg, 2)
peers = torch.randn((grid.S, 2 * cfg.dim), device=DEV).to(torch.bfloat16) * 0.3      # stand-in for the arrived K|V rows


class ServedGather:
    """seqpar.KVGather's interface with the transfer taken out: the peers' rows are ALREADY in the gathered buffer (constant random
    rows: what is left of an exchange once it has landed) and this rank's own rows are placed by a copy on a SIDE stream, as
    RCCL's own stream would (an all-gather writes the local shard too) - nothing of the exchange runs on the compute stream.
    (Round 3's version copied the peers' rows on the compute stream at every start(): ~0.35 ms per layer that no real rank pays.)"""
    timing, n_collectives = None, 0

    def __init__(self, world):
        self.world = world
        self.side = torch.cuda.Stream()
        self.filled = set()
        self.arrival_on = False

    # the arrival-driven attention (ONE launch per layer over the pieces, csrc/attn7p.hip): every piece is there, no flag to wait for;
    # own rows are read in place, so nothing is copied at all
    def enable_arrival(self, ops):
        self.arrival_on = True

    def arrival(self, handle):
        return None, [(j, -1, 0) for j in range(1, self.world)]

    def consumed(self, handle):
        pass

    def allreduce_max(self, t):           # e4m3 wire format: the abs-max exchange (2 x heads floats) - nothing to time on one rank
        pass

    def start(self, rows, out):
        m = rows.shape[0]
        if out.data_ptr() not in self.filled:            # first use of this slice of the gathered buffer: the peers' rows
            if rows.dtype == torch.uint8:                # e4m3 blobs: every "peer" piece = a copy of this rank's valid blob
                for j in range(out.shape[0] // m):
                    out[j * m:(j + 1) * m].copy_(rows)
            else:
                out.copy_(peers[: out.shape[0]])
            self.filled.add(out.data_ptr())
        if self.arrival_on:
            return ()
        ready = torch.cuda.Event()
        ready.record()
        self.side.wait_event(ready)
        with torch.cuda.stream(self.side):
            out[:m].copy_(rows)
            done = torch.cuda.Event()
            done.record()
        return (done,)

    def wait(self, handle):
        for ev in handle:
            torch.cuda.current_stream().wait_event(ev)


def time_forward(world, chunks, iters=3, pair=False, arrival=False):
    """ms per layer of ONE forward on the shard (pair=False), or of BOTH CFG forwards issued as WanDiT.forward_pair (pair=True:
    what a rank of the `sp` layout runs per step - projections over 2n rows, exchange + attention per branch)."""
    plan = ShardPlan.make(grid.S, world, 0)
    m = WanDiT(cfg, sd, ops, bsd, gemm_dtype=GEMM, attn_dtype=ATTN).prepare(grid, plan, kv_gather=ServedGather(world) if world > 1 else None,
                                                                           sp_chunks=chunks, graphs=False,
                                                                           kv_exchange="allgather+arrival" if arrival and world > 1 else None)
    assert m.attn_arrival == bool(arrival and world > 1)
    clip = syn.make_clip_features(cfg) if cfg.has_image_input else None
    ck, bt = m.encode_context(ctx, clip), m.embed_buffers(bl)
    if cfg.has_image_input:
        bt = m.embed_cond_latents(syn.make_cond_latents(cfg, grid), add_to=bt)
    cu = m.encode_context(ctx2, clip) if pair else None

    def run():
        if pair and m._pair_ok():
            m.forward_pair(noise, ck, cu, 500.0, bt, m.head_out)
        elif pair:                      # the product's own rule (2n rows would span >= 4 GiB of an operand: 720p on one GPU): two forwards
            m.forward_tokens(noise, ck, 500.0, bt, m.head_out[0])
            m.forward_tokens(noise, cu, 500.0, bt, m.head_out[1])
        else:
            m.forward_tokens(noise, ck, 500.0, bt, m.head_out[0])

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    del m
    torch.cuda.empty_cache()
    return e0.elapsed_time(e1) / iters / LAYERS          # ms per layer (patch embed + head amortised: < 0.1 %)


rows = []
ONLY = os.environ.get("ONLY")          # "world:chunks[:pair]" -> time just that shard shape (for a rocprofv3 kernel trace of it)
if ONLY:
    w, c, *rest = ONLY.split(":")
    t = time_forward(int(w), int(c), iters=int(os.environ.get("ITERS", "3")), pair="pair" in rest, arrival="arrival" in rest)
    print(f"{MODEL} S={grid.S} gemm {GEMM} attn {ATTN} wire {os.environ.get('ICV_FP8_WIRE', 'e4m3') if ATTN == 'fp8' else 'bf16'}: shard 1/{w} chunks {c}{' pair' if rest else ''}: {t:.3f} ms per layer")
    sys.exit(0)
L = cfg_full.num_layers
# one GPU: the product's default step = the CFG-batched pair (bench.py's `cfg_forwards_batched`)
base1 = time_forward(1, 1)
base = time_forward(1, 1, pair=True)
one_gpu_step = L * base
print(f"1 GPU: {base:.2f} ms per layer for both forwards of a step (pair-batched; two separate forwards: {2 * base1:.2f}) -> {one_gpu_step:.0f} ms per step x {L} layers")
for n_gpus, layout, world, pair in ((2, "cfg+sp", 1, False), (4, "cfg+sp", 2, False), (8, "cfg+sp (auto)", 4, False), (8, "sp", 8, True), (8, "sp unpaired", 8, False),
                                    (4, "sp", 4, True), (2, "sp", 2, True)):
    for chunks, arrival in (((1, False),) if world == 1 else ((4, True), (4, False), (2, False)) if ATTN == "bf16" else ((4, False), (2, False))):
        if pair:
            t = time_forward(world, chunks, pair=True, arrival=arrival)          # both forwards of the step
        else:
            t = (base1 if world == 1 else time_forward(world, chunks, arrival=arrival)) * (2 if layout.startswith("sp") else 1)
        step = L * t
        rows.append(dict(n_gpus=n_gpus, layout=layout, sp_world=world, sp_chunks=chunks, attention="one arrival-gated launch" if arrival else "chunk launches",
                         ms_per_layer_per_step=t, compute_only_ms_per_step=step,
                         compute_only_steps_per_s=1e3 / step, compute_only_scaling=one_gpu_step / step))
        print(f"N = {n_gpus} {layout:14s} shard 1/{world} chunks {chunks}{' ARRIVAL' if arrival else '        '}: {t:6.2f} ms per layer per step -> {step:7.1f} ms per step, "
              f"{1e3 / step:.3f} steps/s, x{one_gpu_step / step:.2f} of one GPU (compute only, exchange fully hidden)")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dict(model=cfg_full.name, gemm=GEMM, attn=ATTN, S=grid.S, layers_timed=LAYERS, one_gpu_ms_per_step=one_gpu_step, rows=rows), open("gpurun_out/sp_compute_only_projection.json", "w"), indent=1)
