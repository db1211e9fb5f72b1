"""One rank of an N-process sequence-parallel run, launched by tests/test_multigpu_rccl.py (real RCCL ranks, one per GPU,
when the box has >= 2 GPUs) and by its CPU twin (gloo ranks with the test-only oracle operator set).  Environment: RANK,
WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT as torch.distributed.run sets them.

    python tests/rank_worker.py --backend nccl|gloo --ops hip|oracle --scenario loop|layer --model tiny|14b \
           --frames F --height H --width W --parallelism sp|cfg+sp|auto --kv-exchange allgather|p2p|native --out result.pt

  loop  : a complete CFG denoising loop (3 steps) of the named model under the layout; rank 0 saves the gathered latent.
  layer : ONE block of the named model (e.g. Wan2.1-14B at S = 37 440) under plain token sharding; every rank's slice of
          the residual stream after the block is gathered and rank 0 saves it.
World 1 (no process group) computes the single-process result the N-rank one is compared with.
"""
import argparse
import dataclasses
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

from infinicube_amd.videogen import synthetic as syn  # noqa: E402
from infinicube_amd.videogen.config import TokenGrid, preset  # noqa: E402
from infinicube_amd.videogen.dit import WanDiT  # noqa: E402
from infinicube_amd.videogen.scheduler import FlowMatchScheduler  # noqa: E402
from infinicube_amd.videogen.seqpar import BranchExchange, ParallelLayout, gather_latent  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--ops", default="hip", choices=["hip", "oracle"])
    ap.add_argument("--scenario", default="loop", choices=["loop", "layer"])
    ap.add_argument("--model", default="tiny")
    ap.add_argument("--frames", type=int, default=9)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=96)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--parallelism", default="auto")
    ap.add_argument("--kv-exchange", default="allgather")
    ap.add_argument("--sp-chunks", type=int, default=3)
    ap.add_argument("--gemm-dtype", default="bf16")
    ap.add_argument("--attn-dtype", default="bf16")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (gloo backend): the multi-process path on a 1-GPU box")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", str(rank)))
    import torch.distributed as dist
    if a.ops == "hip":
        from infinicube_amd.videogen.ops import HipOps
        torch.cuda.set_device(local)
        ops, dev = HipOps(f"cuda:{local}"), torch.device("cuda", local)
    else:
        from oracle_ops import OracleOps
        torch.set_num_threads(2)
        ops, dev = OracleOps(), torch.device("cpu")
    if world > 1:
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    try:
        cfg, grid = preset(a.model), TokenGrid(a.frames, a.height, a.width)
        if a.scenario == "layer":
            cfg = dataclasses.replace(cfg, num_layers=1)
        on_dev = dict(device=dev, dtype=torch.bfloat16) if a.ops == "hip" else {}
        sd = syn.make_dit_state_dict(cfg, seed=0, **on_dev)
        bsd = syn.make_buffer_embedder_state_dict(cfg, **on_dev)
        noise = syn.make_latent_noise(grid)
        c1, c2, bl = syn.make_text_context(cfg, 1), syn.make_text_context(cfg, 2), syn.make_buffer_latents(cfg, grid)
        lay = ParallelLayout.make(world, rank, "sp" if a.scenario == "layer" else a.parallelism, use_cfg=True)
        plan = lay.shard_plan(grid.S)
        m = WanDiT(cfg, sd, ops, bsd, gemm_dtype=a.gemm_dtype, attn_dtype=a.attn_dtype).prepare(
            grid, plan, group=lay.sp_group, sp_chunks=a.sp_chunks, graphs=False, kv_exchange=a.kv_exchange if world > 1 else None)
        del sd, bsd
        lat = noise.clone().to(dev)
        info = dict(world=world, mode=lay.mode, sp_world=lay.sp_world, kv_exchange=a.kv_exchange,
                    backend=(dist.get_backend() if world > 1 else None))
        if a.scenario == "loop":
            bt = m.embed_buffers(bl)
            if lay.mode == "cfg+sp":
                m.denoise(lat, m.encode_context(c1) if lay.branch == 0 else None, m.encode_context(c2) if lay.branch == 1 else None,
                          bt, FlowMatchScheduler(a.steps), 5.0, branch_exchange=BranchExchange(lay))
            else:
                m.denoise(lat, m.encode_context(c1), m.encode_context(c2), bt, FlowMatchScheduler(a.steps), 5.0)
            res = gather_latent(lat, plan, grid, group=lay.sp_group)
            info["kv_collectives"] = m.kv_gather.n_collectives if m.kv_gather is not None else 0
        else:
            m.forward_tokens(lat, m.encode_context(c1), 731.0, m.embed_buffers(bl), m.head_out[0], num_layers=1)
            mine = m.x.contiguous()                      # residual stream after the block, this rank's tokens
            if world > 1:
                res = torch.empty((grid.S, cfg.dim), dtype=mine.dtype, device=mine.device)
                dist.all_gather_into_tensor(res, mine)
            else:
                res = mine
            info["kv_collectives"] = m.kv_gather.n_collectives if m.kv_gather is not None else 0
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        assert torch.isfinite(res).all(), "non-finite result"
        if world > 1:
            # every rank must end with the same full result
            mine_sum = res.double().sum().reshape(1).to("cpu" if a.backend == "gloo" else dev)
            sums = [torch.empty_like(mine_sum) for _ in range(world)]
            dist.all_gather(sums, mine_sum)
            assert all(float(s) == float(sums[0]) for s in sums), f"ranks disagree on the gathered result: {[float(s) for s in sums]}"
        if rank == 0:
            torch.save(dict(result=res.float().cpu(), info=info), a.out)
    finally:
        if world > 1:
            from infinicube_amd.videogen.seqpar import _NativeComm
            _NativeComm.close_all()
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
