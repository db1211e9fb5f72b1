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
    """All-gather of the local K and V shards [n_tok, d] into full [S, d] buffers."""

    def __init__(self, plan: ShardPlan, group=None):
        self.plan = plan
        self.group = group
        if plan.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("KVGather with world>1 needs torch.distributed to be initialised")
            self.dist = dist

    def __call__(self, k_loc: torch.Tensor, v_loc: torch.Tensor, k_full: torch.Tensor, v_full: torch.Tensor):
        if self.plan.world == 1:
            raise RuntimeError("KVGather called with world == 1 (attend over the local buffers directly)")
        assert k_loc.is_contiguous() and v_loc.is_contiguous() and k_full.is_contiguous() and v_full.is_contiguous()
        # rank-major concatenation == global token order because shards are contiguous token ranges
        self.dist.all_gather_into_tensor(k_full, k_loc, group=self.group)
        self.dist.all_gather_into_tensor(v_full, v_loc, group=self.group)


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
