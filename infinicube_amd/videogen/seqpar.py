"""Token-sequence parallelism for the DiT (BASELINE.json config #4).

Every op of a DiT block except self-attention is token-local, so the (f h w) token axis is cut
into ``world`` contiguous shards (one per GPU / process).  The only exchange is the per-layer
all-gather of the post-RoPE K and V shards (RCCL over xGMI through torch.distributed; the `nccl`
backend IS RCCL on ROCm).  The reference has no such path — it runs one dense sequence on one
GPU [R infinicube/inference/guidance_buffer_generation.py:759-766].
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch


@dataclass(frozen=True)
class ShardPlan:
    world: int
    rank: int
    S: int          # global token count
    tok0: int       # first global token of this shard
    n_tok: int      # tokens in this shard

    @staticmethod
    def make(S: int, world: int = 1, rank: int = 0) -> "ShardPlan":
        if world < 1 or not (0 <= rank < world):
            raise ValueError(f"bad (world, rank) = ({world}, {rank})")
        if S % world:
            raise ValueError(
                f"token count S={S} is not divisible by world={world}; choose a GPU count that divides "
                f"the sequence (93f 480p: S=37440 = 2^6*3^2*5*13)")
        n = S // world
        return ShardPlan(world, rank, S, rank * n, n)


class KVGather:
    """Chunked, asynchronous all-gather of the local K and V shards.

    ``start(k_rows, v_rows, k_out, v_out)`` enqueues the all-gather of ONE row-chunk of the local
    shard ([m, d] from every rank -> [world*m, d], rank-major) and returns a handle; ``wait(handle)``
    makes the compute stream wait for that chunk only.  With the ``nccl`` backend (= RCCL over xGMI) the
    collectives run on RCCL's own stream, so chunk c+1 is in flight while attention consumes chunk c
    (dit.WanDiT._sp_attention).  Attention is invariant to key order, so a chunk does not have to be a
    contiguous range of global tokens."""

    def __init__(self, plan: ShardPlan, group=None):
        self.plan = plan
        self.group = group
        if plan.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("KVGather with world>1 needs torch.distributed to be initialised")
            self.dist = dist

    def start(self, k_rows: torch.Tensor, v_rows: torch.Tensor, k_out: torch.Tensor, v_out: torch.Tensor):
        if self.plan.world == 1:
            raise RuntimeError("KVGather used with world == 1 (attend over the local buffers directly)")
        assert k_rows.is_contiguous() and v_rows.is_contiguous() and k_out.is_contiguous() and v_out.is_contiguous()
        assert k_out.shape[0] == self.plan.world * k_rows.shape[0]
        return (self.dist.all_gather_into_tensor(k_out, k_rows, group=self.group, async_op=True),
                self.dist.all_gather_into_tensor(v_out, v_rows, group=self.group, async_op=True))

    def wait(self, handle) -> None:
        for w in handle:
            w.wait()   # nccl: the CURRENT STREAM waits (host does not block); gloo: host blocks


def chunk_bounds(n_rows: int, chunks: int):
    """Row boundaries of the K/V gather chunks of one shard (identical on every rank).  The sizes RAMP
    (weights 1, 3, 6, 6, ...): only the first chunk's transfer is exposed before attention can start,
    so it is small; later chunks are large so launches stay efficient and each transfer hides under the
    previous chunk's attention."""
    chunks = max(1, min(chunks, n_rows))
    w = [1.0, 3.0][:chunks] + [6.0] * max(0, chunks - 2)
    tot, acc, out = sum(w), 0.0, [0]
    for c in range(chunks):
        acc += w[c]
        out.append(n_rows if c == chunks - 1 else max(out[-1] + 1, int(round(n_rows * acc / tot))))
    return out


def gather_latent(latent: torch.Tensor, plan: ShardPlan, grid, group=None) -> torch.Tensor:
    """After the loop each rank has stepped only its own tokens' latent patches; exchange them so
    every rank holds the full latent (once per generation, 4.8 MB at 480p)."""
    if plan.world == 1:
        return latent
    import torch.distributed as dist
    C, T, H8, W8 = latent.shape
    Hp, Wp = H8 // 2, W8 // 2
    # view latent as tokens: [T, Hp, Wp, C, 2, 2] -> [S, C*4]
    tok = latent.reshape(C, T, Hp, 2, Wp, 2).permute(1, 2, 4, 0, 3, 5).reshape(T * Hp * Wp, C * 4).contiguous()
    mine = tok[plan.tok0: plan.tok0 + plan.n_tok].contiguous()
    full = torch.empty_like(tok)
    dist.all_gather_into_tensor(full, mine, group=group)
    return full.reshape(T, Hp, Wp, C, 2, 2).permute(3, 0, 1, 4, 2, 5).reshape(C, T, H8, W8).contiguous()
