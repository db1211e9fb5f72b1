"""Host driver of the buffer-conditioned Wan2.1 DiT forward and denoising loop.

This replaces what ``self.pipe(...)`` does inside its sampling loop
[R infinicube/videogen/inference.py:216-226] — the diffsynth fork's ``WanModel.forward`` per step,
twice (cond / uncond), plus CFG and the flow-match Euler update.
All arithmetic runs in libicvideo HIP kernels through ``ops`` (ops.HipOps); this file owns the
HBM layout, the step-invariant caches and the sequence-parallel schedule:

  HBM layout (per rank, n = local tokens, d = model dim)
    x        f32  [n, d]        residual stream (fp32: 288 GB of HBM3E makes this free, and it
                                removes the bf16 round-off random walk over 3 adds x L layers x 100 forwards)
    h        bf16 [n, d]        LN/modulate output -> GEMM A operand
    qkv      bf16 [3, n, d]     q | k | v planes written by ONE fused QKV GEMM (split epilogue)
    kv_loc   bf16 [n, 2d]       world > 1: this shard's keys and values as ONE matrix (row = k(d) | v(d)) written by the
                                K/V half of the QKV GEMM, so one collective per row-chunk moves both
    kv_full  bf16 [S, 2d]       world > 1: the gathered rows of every rank (chunk-major, rank-major inside)
    att      bf16 [n, d]        attention output -> O-projection A operand
    ff       bf16 [n, ffn]      GELU(FFN1) output
    weights  bf16 [N, K]        torch Linear layout == the K-contiguous B operand of the MFMA GEMM
  Step-invariant caches: guidance-buffer tokens f32 [n, d] (once per generation; K1), text
  embedding + per-layer cross-attention K/V bf16 [L, text_len, d] per CFG branch (K9), RoPE tables.
  i2v (BASELINE.json config #5, [EXT] Wan2.1 i2v): the conditioning latent y (4 mask + 16 first-frame
  channels) is step-invariant and the patch embedding is linear, so its tokens W[:,16:]·patch(y) are
  computed ONCE and folded into the cached buffer tokens — the per-step patch GEMM stays 16-channel;
  the CLIP tokens give a second cached K/V set [L, 257, d] whose attention is summed onto the text one.
  fp8 (config #5's "fp8 MFMA weights", ``gemm_dtype="fp8"``): the six per-layer projections (QKV, O, cross-q,
  cross-o, FFN1, FFN2 = 99.6 % of the GEMM flops) run on the K=128 fp8 MFMA at twice the bf16 rate.  Weights are
  quantised once at load to e4m3 + one f32 scale per output channel (14 GB instead of 28); activations get one
  scale per token: LN/modulate emits e4m3 directly (h8), the bf16 attention / GELU outputs take one
  quantise pass (att8, ff8).  Residual stream, attention, norms, embeddings and the head stay as above.
"""

from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from .config import TokenGrid, WanDiTConfig
from .ops import BF16, EPI_BF16, EPI_F32, EPI_GELU_BF16, EPI_RESID_F32, F32, FP8, RopeTable
from .scheduler import FlowMatchScheduler
from .seqpar import BranchExchange, KVGather, ShardPlan, chunk_bounds  # noqa: F401

ARRIVAL_SUFFIX = "+arrival"      # K|V exchange mode "<transport>+arrival": ONE arrival-gated attention launch per layer (csrc/attn7p.hip)


def split_kv_mode(mode: Optional[str]):
    """("ipc+arrival") -> ("ipc", True).  ICV_ATTN_ARRIVAL=1 / 0 forces the arrival-driven attention on / off for every mode."""
    arrival = bool(mode) and mode.endswith(ARRIVAL_SUFFIX)
    base = mode[: -len(ARRIVAL_SUFFIX)] if arrival else mode
    env = os.environ.get("ICV_ATTN_ARRIVAL")
    if env in ("0", "1"):
        arrival = env == "1"
    return (base or None), arrival


class _ChunkBufs(list):
    """The gathered row-chunks of one exchange + the local rows they were cut from (the arrival-driven attention reads its own rows in place)."""
    own = None

ACT_SILU = 1


@dataclass
class ContextKV:
    """Cross-attention keys/values of one prompt for all layers (step-invariant)."""
    k: torch.Tensor  # bf16 [L, text_len, d]
    v: torch.Tensor  # bf16 [L, text_len, d]
    k_img: Optional[torch.Tensor] = None  # i2v: bf16 [L, img_len, d]
    v_img: Optional[torch.Tensor] = None


class WanDiT:
    FP8_WEIGHTS = ("wqkv", "wo", "xq_w", "xo_w", "f0_w", "f2_w")
    # Default e4m3 set of the fp8 mode.  Every quantised GEMM adds ~3-5 % relative noise to its output (e4m3 has 3 mantissa
    # bits), and which projections can afford it was measured at the REAL depths against the bf16 path (random-init weights,
    # CFG 5; tools/fp8_subset_study.py):
    #   1.3B (30 layers, 10 steps, profiles/r02/fp8_projection_subsets_psnr.txt): each FFN GEMM alone 43.3 dB, every other
    #        projection 52-57 dB, all six 39.8 dB, all but FFN2 42.6 dB;
    #   14B (40 layers, d = 5120, S = 37 440, 4-step loop, profiles/r03/fp8_projection_subsets_psnr_14b_depth.txt): e4m3
    #        self-attention alone 47.1 dB, cross-q 46.4, cross-o 43.9, QKV 42.8 - but O 34.0, FFN1 30.8, FFN2 31.0 (their
    #        outputs are added straight onto the residual stream, 40 layers deep), round 2's set (all but FFN2) 28.9 dB.
    # The bar is >= 40 dB at the depth of the model that is run, so the default is the QKV projection (26 % of the GEMM flops)
    # plus e4m3 self-attention (57 % of a bf16 step): asserted against the fp32 oracle at both depths
    # (test_config1_*: 1.3B; test_config3_wan_14b_full_depth_forwards_and_loop: 14B).  ICV_FP8_WEIGHTS / fp8_weights= selects any other
    # subset for a caller whose checkpoint tolerates more.
    FP8_DEFAULT = ("wqkv",)

    def __init__(self, cfg: WanDiTConfig, state_dict: Dict[str, torch.Tensor], ops,
                 buffer_embedder_sd: Optional[Dict[str, torch.Tensor]] = None, gemm_dtype: str = "bf16",
                 attn_dtype: str = "bf16", fp8_weights: Optional[tuple] = None):
        self.cfg = cfg.validate()
        self.ops = ops
        if gemm_dtype not in ("bf16", "fp8"):
            raise ValueError(f"gemm_dtype must be 'bf16' or 'fp8', got {gemm_dtype!r}")
        self.fp8 = gemm_dtype == "fp8"
        # which of the six per-layer projections run in e4m3 when gemm_dtype == "fp8" (default FP8_DEFAULT; ICV_FP8_WEIGHTS=
        # "wqkv,f0_w,..." or the fp8_weights argument overrides it: e4m3 has 3 mantissa bits, so every quantised GEMM adds
        # ~3-5 % relative noise to its output however fine the scales are - DESIGN.md §7 and profiles/r03/fp8_projection_subsets_psnr_14b_depth.txt have the accuracy / speed table)
        env = os.environ.get("ICV_FP8_WEIGHTS")
        sel = fp8_weights if fp8_weights is not None else (tuple(x for x in env.split(",") if x) if env else self.FP8_DEFAULT)
        bad = [x for x in sel if x not in self.FP8_WEIGHTS]
        if bad:
            raise ValueError(f"fp8_weights: unknown projection(s) {bad}; choose from {self.FP8_WEIGHTS}")
        self.fp8_set = tuple(sel) if self.fp8 else ()
        if attn_dtype not in ("bf16", "fp8"):
            raise ValueError(f"attn_dtype must be 'bf16' or 'fp8', got {attn_dtype!r}")
        # fp8 self-attention (e4m3 Q/K/V/P on the K=64 scaled MFMA, csrc/attn8.hip), single-rank and sequence-parallel
        # (each gathered K/V chunk is quantised on its own); cross-attention (512 + 257 keys, 1 % of the flops) stays bf16
        self.attn_fp8 = attn_dtype == "fp8"
        if self.fp8 and (cfg.dim % 128 or cfg.ffn_dim % 128):
            raise ValueError("fp8 GEMMs need dim and ffn_dim to be multiples of 128")
        d = cfg.dim
        sd = state_dict
        W = lambda name: ops.to_device(sd[name], BF16)   # noqa: E731  matrices: bf16 in HBM
        V = lambda name: ops.to_device(sd[name], F32)    # noqa: E731  vectors: fp32

        kp = cfg.in_dim * cfg.patch_elems
        kx = cfg.out_dim * cfg.patch_elems               # columns of the noise channels (c-major: c*4 + y*2 + z)
        pw = W("patch_embedding.weight").reshape(d, kp)
        self.k_patch = ((kx + 63) // 64) * 64            # GEMM K granularity
        self.patch_w = self._pad_k(pw[:, :kx], self.k_patch)
        self.patch_b = V("patch_embedding.bias")
        self.cond_w = None
        if kp > kx:                                      # i2v: columns of the step-invariant y channels
            self.k_cond = ((kp - kx + 63) // 64) * 64
            self.cond_w = self._pad_k(pw[:, kx:], self.k_cond)
        if cfg.has_image_input:
            self.img_ln0 = (V("img_emb.proj.0.weight"), V("img_emb.proj.0.bias"))
            self.img1_w, self.img1_b = W("img_emb.proj.1.weight"), V("img_emb.proj.1.bias")
            self.img3_w, self.img3_b = W("img_emb.proj.3.weight"), V("img_emb.proj.3.bias")
            self.img_ln4 = (V("img_emb.proj.4.weight"), V("img_emb.proj.4.bias"))
        self.text0_w, self.text0_b = W("text_embedding.0.weight"), V("text_embedding.0.bias")
        self.text2_w, self.text2_b = W("text_embedding.2.weight"), V("text_embedding.2.bias")
        self.time0_w, self.time0_b = W("time_embedding.0.weight"), V("time_embedding.0.bias")
        self.time2_w, self.time2_b = W("time_embedding.2.weight"), V("time_embedding.2.bias")
        self.tproj_w, self.tproj_b = W("time_projection.1.weight"), V("time_projection.1.bias")
        self.head_w, self.head_b = W("head.head.weight"), V("head.head.bias")
        self.head_mod = V("head.modulation").reshape(2, d).contiguous()
        # softmax scale folded into the K norm weights (see the module docstring); attention then runs at unit scale
        self.k_fold = (1.0 / math.sqrt(cfg.head_dim)) * math.log2(math.e)
        self.attn_scale = math.log(2.0)
        self.layers = []
        mods = []
        for i in range(cfg.num_layers):
            p = f"blocks.{i}"
            sa, ca = f"{p}.self_attn", f"{p}.cross_attn"
            lw = dict(
                wqkv=torch.cat([W(f"{sa}.q.weight"), W(f"{sa}.k.weight"), W(f"{sa}.v.weight")], 0).contiguous(),
                bqkv=torch.cat([V(f"{sa}.q.bias"), V(f"{sa}.k.bias"), V(f"{sa}.v.bias")], 0).contiguous(),
                nq=V(f"{sa}.norm_q.weight"), nk=V(f"{sa}.norm_k.weight") * self.k_fold,
                wo=W(f"{sa}.o.weight"), bo=V(f"{sa}.o.bias"),
                n3w=V(f"{p}.norm3.weight"), n3b=V(f"{p}.norm3.bias"),
                xq_w=W(f"{ca}.q.weight"), xq_b=V(f"{ca}.q.bias"),
                xkv_w=torch.cat([W(f"{ca}.k.weight"), W(f"{ca}.v.weight")], 0).contiguous(),
                xkv_b=torch.cat([V(f"{ca}.k.bias"), V(f"{ca}.v.bias")], 0).contiguous(),
                xnq=V(f"{ca}.norm_q.weight"), xnk=V(f"{ca}.norm_k.weight") * self.k_fold,
                xo_w=W(f"{ca}.o.weight"), xo_b=V(f"{ca}.o.bias"),
                f0_w=W(f"{p}.ffn.0.weight"), f0_b=V(f"{p}.ffn.0.bias"),
                f2_w=W(f"{p}.ffn.2.weight"), f2_b=V(f"{p}.ffn.2.bias"),
            )
            if self.fp8:                                  # e4m3 rows + per-output-channel scale; bf16 copy dropped
                for nm in self.fp8_set:
                    lw[nm] = self._quantize_weight(lw[nm])
            if cfg.has_image_input:
                lw["xkv_img_w"] = torch.cat([W(f"{ca}.k_img.weight"), W(f"{ca}.v_img.weight")], 0).contiguous()
                lw["xkv_img_b"] = torch.cat([V(f"{ca}.k_img.bias"), V(f"{ca}.v_img.bias")], 0).contiguous()
                lw["xnk_img"] = V(f"{ca}.norm_k_img.weight") * self.k_fold
            self.layers.append(lw)
            mods.append(V(f"{p}.modulation").reshape(6 * d))
        self.modulation = torch.stack(mods, 0).contiguous()  # f32 [L, 6d]
        self.buffer_embedder = None
        if buffer_embedder_sd is not None:
            self.load_buffer_embedder(buffer_embedder_sd)
        self.plan: Optional[ShardPlan] = None

    # ------------------------------------------------------------------------------------
    def _quantize_weight(self, w: torch.Tensor):
        q = torch.empty(w.shape, dtype=FP8, device=w.device)
        sc = torch.empty((w.shape[0],), dtype=F32, device=w.device)
        self.ops.quantize_rows(w, q, sc)
        return (q, sc)

    # GEMM operands are either a bf16 tensor or an (e4m3 rows, f32 row scales) pair; these three helpers keep
    # forward_tokens identical for both GEMM dtypes.
    def _norm(self, w, rows: Optional[slice] = None, **kw):
        """K3 / K8: LayerNorm(+affine)(+modulate) of the residual stream into the A operand of the GEMM with weight
        ``w`` (e4m3 rows + scales when that weight is quantised, bf16 otherwise); ``rows``: a row range of the workspace."""
        r = slice(None) if rows is None else rows
        if isinstance(w, tuple):
            self.ops.ln_modulate_fp8(self.x[r], self.h8[r], self.h8s[r], **kw)
            return (self.h8[r], self.h8s[r])
        self.ops.ln_modulate(self.x[r], self.h[r], **kw)
        return self.h[r]

    def _operand(self, t: torch.Tensor, q8: Optional[torch.Tensor], s8: Optional[torch.Tensor], w, rows: Optional[slice] = None):
        """bf16 activation produced by attention / the GELU epilogue -> A operand of the GEMM with weight ``w``."""
        r = slice(None) if rows is None else rows
        if isinstance(w, tuple):
            self.ops.quantize_rows(t[r], q8[r], s8[r])
            return (q8[r], s8[r])
        return t[r]

    def _mm(self, a, w, bias, out, epi, rows: Optional[slice] = None, **kw):
        """out = epilogue(a @ w[rows].T + bias[rows])."""
        if rows is not None:
            w = (w[0][rows], w[1][rows]) if isinstance(w, tuple) else w[rows]
            bias = bias[rows]
        if isinstance(w, tuple):
            self.ops.gemm_fp8(a[0], a[1], w[0], w[1], bias, out, epi, **kw)
        else:
            self.ops.gemm(a, w, bias, out, epi, **kw)

    @staticmethod
    def _pad_k(w: torch.Tensor, k_to: int) -> torch.Tensor:
        if w.shape[1] == k_to:
            return w.contiguous()
        out = torch.zeros((w.shape[0], k_to), dtype=w.dtype, device=w.device)
        out[:, : w.shape[1]] = w
        return out

    def load_buffer_embedder(self, bsd: Dict[str, torch.Tensor]):
        """Guidance-buffer embedder weights (strict).  Variant recovered from the key names:
        'proj.*' = one conv over the channel-concatenated buffers (H1), '{semantic,coordinate}_proj.*'
        = one conv per buffer (H2).  SURVEY.md §8a K1 — the fork's real layout is unknown ([EXT])."""
        d, ops = self.cfg.dim, self.ops
        keys = set(bsd.keys())
        if keys == {"proj.weight", "proj.bias"}:
            names = ["proj"]
        elif keys == {"semantic_proj.weight", "semantic_proj.bias", "coordinate_proj.weight", "coordinate_proj.bias"}:
            names = ["semantic_proj", "coordinate_proj"]
        else:
            raise KeyError(f"buffer embedder: unexpected keys {sorted(keys)} (strict load)")
        convs = []
        for nm in names:
            w = ops.to_device(bsd[f"{nm}.weight"], BF16)
            cin = w.shape[1]
            k = cin * self.cfg.patch_elems
            kpad = ((k + 63) // 64) * 64
            convs.append(dict(w=self._pad_k(w.reshape(d, k), kpad), b=ops.to_device(bsd[f"{nm}.bias"], F32),
                              cin=cin, k=kpad))
        self.buffer_embedder = convs

    # ------------------------------------------------------------------------------------
    GRAPH_MAX_TOKENS = 8192   # graphs="auto": only sizes whose forward is made of many short kernels (cfg #1: S = 2240)

    def prepare(self, grid: TokenGrid, plan: Optional[ShardPlan] = None, kv_gather=None, sp_chunks: int = 4,
                group=None, graphs=False, kv_exchange: Optional[str] = None, force_sp: bool = False):
        """Allocate the per-generation workspace for this token grid / shard.  ``group`` = the process group the
        K/V all-gather runs in (seqpar.ParallelLayout.sp_group; None = the default group).
        ``graphs``: replay each DiT forward (its ~25 launches x L layers) as ONE hipGraph instead of issuing the
        kernels from Python: True / False / "auto" (= single-rank runs of at most GRAPH_MAX_TOKENS tokens on a GPU;
        env ICV_GRAPHS=0|1 overrides).  The time embedding and the fused Euler step take a per-step scalar by value
        and stay outside the graph.  Off by default: measured on MI355X the loop is not host-bound even at S = 400
        (10.7 us per launch eagerly AND replayed - the cost is the GPU-side dispatch of many tiny kernels), so replay
        buys nothing today; it is kept, tested bit-identical, for hosts that are slower at issuing launches.
        ``force_sp`` (env ICV_FORCE_SP=1): run the sequence-parallel schedule even on ONE rank — K|V into the [n, 2d] row
        matrix, a one-rank exchange, chunked attention with carried state: the one-GPU rehearsal of the N-GPU path."""
        cfg, ops = self.cfg, self.ops
        self.grid = grid
        self.plan = plan or ShardPlan.make(grid.S)
        self.sp_on = self.plan.world > 1 or force_sp or os.environ.get("ICV_FORCE_SP", "0") == "1"
        if self.plan.S != grid.S:
            raise ValueError("shard plan does not match the token grid")
        n, d, S = self.plan.n_tok, cfg.dim, grid.S
        self.rope = RopeTable.build(grid.T, grid.Hp, grid.Wp, ops.device, cfg.head_dim)
        a = ops.alloc
        self.x = a((n, d), F32)
        self.x_stem = a((n, d), F32)       # residual stream after layer 0's self-attention block, shared by the CFG branches
        self.h = a((n, d), BF16)
        self.qkv = a((3, n, d), BF16)
        self.att = a((n, d), BF16)
        self.ff = a((n, cfg.ffn_dim), BF16)
        self.attn8_ws = None
        if self.attn_fp8:      # world > 1: the K/V side of the workspace holds one gathered chunk at a time
            kv_rows = n if not self.sp_on else self.plan.world * max(
                b1 - b0 for b0, b1 in zip(chunk_bounds(n, sp_chunks)[:-1], chunk_bounds(n, sp_chunks)[1:]))
            self.attn8_ws = ops.attention_fp8_buffers(n, kv_rows, d, cfg.num_heads)
        self.h8 = self.h8s = self.att8 = self.att8s = self.ff8 = self.ff8s = None
        if self.fp8:
            self.h8, self.h8s = a((n, d), FP8), a((n,), F32)
            self.att8, self.att8s = a((n, d), FP8), a((n,), F32)
            self.ff8, self.ff8s = a((n, cfg.ffn_dim), FP8), a((n,), F32)
        self.patches = torch.zeros((n, self.k_patch), dtype=BF16, device=ops.device)
        self.head_out = a((2, n, cfg.out_dim * cfg.patch_elems), F32)
        self.head_own = a((n, cfg.out_dim * cfg.patch_elems), F32)   # cfg+sp: this rank's branch before the swap
        self.mod = a((cfg.num_layers, 6 * d), F32)
        self.hmod = a((2, d), F32)
        self.t_sin = a((1, cfg.freq_dim), F32)
        self.t_a = a((1, d), F32)
        self.t_e = a((1, d), F32)
        self.t_mod = a((1, 6 * d), F32)
        self._t_cached = None
        env = os.environ.get("ICV_GRAPHS")
        if env is not None:
            graphs = env == "1"
        if graphs == "auto":
            graphs = n <= self.GRAPH_MAX_TOKENS
        self._graphs_on = bool(graphs) and not self.sp_on and self._is_gpu()
        self._graphs = {}
        # ICV_DUAL_STREAM=1: run the cond / uncond forwards of a step concurrently on two HIP streams (single-rank only).
        # Off by default: measured -5 % at 14B / 480p and neutral at 1.3B — two chip-filling kernels at once break the
        # XCD-local K/V and weight reuse of each other more than they fill each other's tail waves.
        self.dual_stream = os.environ.get("ICV_DUAL_STREAM", "0") == "1" and not self.sp_on and self._is_gpu()
        # ICV_NATIVE_FORWARD=1 / self.native_forward = True: one C call (icv_dit_forward) enqueues the whole forward instead of
        # ~13 C-ABI calls per layer from Python — the same launchers in the same order, so bit-identical, in every mode it covers
        # (bf16 / e4m3 projections and attention on one rank; sequence-parallel over an RCCL transport with bf16 rows on the wire and the
        # chunked launches - _native_eligible() says no to the e4m3 wire format, the copy-engine transport and the arrival-driven
        # attention, and the per-op driver runs).  Off by default: host issue time is 0.3 % of a 14B step either way.
        self.native_forward = os.environ.get("ICV_NATIVE_FORWARD", "0") == "1"
        if getattr(self, "_native", None) is not None:      # a new workspace: the old context points at freed buffers
            self.ops.lib.icv_dit_destroy(self._native)
        self._native = None
        # ICV_SHARE_STEM=0 switches off the sharing of the context-free stem between the two CFG forwards (A/B, tests)
        self.share_stem = os.environ.get("ICV_SHARE_STEM", "1") == "1"
        self._twin = None
        # ICV_CFG_BATCH=0: run the two CFG forwards of a step one after the other instead of as one batch of 2n rows
        # (forward_pair; bit-identical either way)
        self.cfg_batch = os.environ.get("ICV_CFG_BATCH", "1") == "1"
        self._pair = None
        if self.sp_on:
            old = getattr(self, "kv_gather", None)
            if old is not None and old is not kv_gather and hasattr(old, "close"):
                old.close()                                        # a copy-engine transport owns a heap and streams
            base_mode, arrival = split_kv_mode(kv_exchange)
            self.kv_gather = kv_gather or KVGather(self.plan, group, base_mode)
            self._sp_set_arrival(arrival)
            self.sp_bounds = chunk_bounds(n, sp_chunks, self.sp_align)
            # e4m3 attention under sequence parallelism: ship e4m3 K|V (each rank quantises its own rows once) instead of bf16
            # rows that every rank re-quantises (ICV_FP8_WIRE=bf16 restores that order of operations)
            self.fp8_wire = self._fp8_wire_wanted()
            self._sp_local_rows()                                  # kv_loc: local k | v rows (one exchange moves both)
            self.kv_full = a((self.plan.world * n, 2 * d), BF16)   # gathered rows (chunk-major, rank-major inside)
            self.sp_acc = a((n, d), F32)                           # carried O accumulator between key chunks
            self.sp_ml = a((n, cfg.num_heads, 2), F32)             # carried (running max, row sum)
        else:
            self.kv_loc, self.kv_full, self.kv_gather = None, None, None
            self.attn_arrival, self.sp_err, self.sp_align = False, None, 1
        return self

    def _sp_local_rows(self):
        """(Re)place the matrices this rank writes its K|V rows into where the current transport wants them: plain workspace
        for the collective modes, the symmetric heap the peers pull from for KVGather mode "ipc" (seqpar._IpcHeap) - sized
        for this engine's [n, 2d] rows plus the [2n, 2d] rows of its CFG-batched pair twin (forward_pair).  With e4m3 on the
        wire (``fp8_wire``) what travels is one e4m3 blob per row chunk (ops.attention_fp8_quantize_kv): those blobs are the
        rows that must live where the transport can reach them, one set per CFG branch of the pair twin."""
        n, d, kg, H = self.plan.n_tok, self.cfg.dim, self.kv_gather, self.cfg.num_heads
        U8 = torch.uint8
        self._kv8_chunks, tot8 = [], 0
        if getattr(self, "fp8_wire", False):
            for b0, b1 in zip(self.sp_bounds[:-1], self.sp_bounds[1:]):
                rows8 = self.ops.attention_fp8_blob_bytes(b1 - b0, H) // d          # blob as [rows8, d] bytes
                self._kv8_chunks.append((tot8, rows8, b0, b1))
                tot8 += rows8
        if hasattr(kg, "reserve"):
            kg.reserve(3 * n * 2 * d * 2 + 3 * tot8 * d + 4096, self.ops.device)
        if getattr(self, "attn_arrival", False):
            kg.enable_arrival(self.ops)      # after reserve(): the copy-engine transport's flags belong to its heap
        if hasattr(kg, "local_rows"):
            rows = lambda r, cols=2 * d, dt=BF16: kg.local_rows(r, cols, dt, self.ops.alloc)      # noqa: E731
        else:
            rows = lambda r, cols=2 * d, dt=BF16: self.ops.alloc((r, cols), dt)                  # noqa: E731
        self.kv_loc = rows(n)
        self._kv_rows = rows
        self.kv8 = self._sp_wire_set(rows, tot8) if tot8 else None
        if getattr(self, "_pair", None) is not None:
            self._pair.kv_loc = rows(2 * n)
            self._pair.kv8 = [self._sp_wire_set(rows, tot8) for _ in range(2)] if tot8 else None

    def _sp_wire_set(self, rows, tot8):
        """(local blobs [tot8, d] bytes, gathered blobs [world * tot8, d] bytes chunk-major, abs-max table f32 [3, H])."""
        d = self.cfg.dim
        return (rows(tot8, d, torch.uint8), self.ops.alloc((self.plan.world * tot8, d), torch.uint8), self.ops.alloc((3, self.cfg.num_heads), F32))

    def _fp8_wire_wanted(self) -> bool:
        return bool(self.attn_fp8 and os.environ.get("ICV_FP8_WIRE", "e4m3") == "e4m3" and hasattr(self.ops, "attention_fp8_quantize_kv")
                    and hasattr(self.kv_gather, "allreduce_max"))

    def _sp_set_arrival(self, want: bool):
        """Arrival-driven self-attention.  bf16: ONE launch per layer over all pieces (csrc/attn7p.hip).  e4m3 mode with the
        e4m3 wire format: the chunk launches stay (a chunk's blobs have one size, the chunks of the ramp do not), but each of them gates on
        its pieces' arrival flags inside the kernel instead of the host waiting for the chunk's whole exchange
        (icv_attention_fp8_fwd_pieces_gated); e4m3 attention over bf16 rows on the wire keeps the host waits."""
        ops = self.ops
        self.attn_arrival = (bool(want) and hasattr(ops, "attention_pieces") and hasattr(self.kv_gather, "enable_arrival")
                             and (not self.attn_fp8 or self._fp8_wire_wanted()))
        self.sp_align = 64 if self.attn_arrival else 1
        self.sp_err = None
        if self.attn_arrival:
            if self._is_gpu():
                self.sp_err = ops.alloc((1,), torch.int32)
                self.sp_err.zero_()
            self.sp_timeout_us = int(float(os.environ.get("ICV_ATTN_ARRIVAL_TIMEOUT_MS", "60000")) * 1000)

    def _check_transport(self):
        """Per step, free (one host word): did a device-side wait of the K|V transport give up?  Then the rest of the loop would run on
        stale rows - stop now (the in-kernel waits fail fast after the first one, so a dead peer costs one deadline, not one per layer)."""
        kg = getattr(self, "kv_gather", None)
        if kg is not None and hasattr(kg, "check"):
            kg.check()

    def exchange_gave_up(self) -> Optional[str]:
        """None, or why the K|V exchange of this engine is no longer valid: a device-side wait of the copy-engine transport or of the
        arrival-driven attention ran into its deadline (both are bounded; what was computed since is garbage).  One host word + one
        4-byte read-back; never raises (seqpar.autotune_kv_exchange asks it after a candidate's warm-up: a transport whose data
        movement cannot make progress beside the WAITING attention work-groups - a starved RCCL channel kernel - shows up here, and
        the candidate is dropped on every rank)."""
        kg = getattr(self, "kv_gather", None)
        if kg is not None and hasattr(kg, "check"):
            try:
                kg.check()
            except RuntimeError as e:
                return str(e)
        err = getattr(self, "sp_err", None)
        if err is not None:
            e = int(err.item()) & 0xffffffff
            if e:
                return (f"sequence-parallel attention gave up waiting for K|V piece {e & 0xffff} of a layer after "
                        f"{self.sp_timeout_us / 1e6:.0f} s (the transfer made no progress beside the waiting attention, or a peer is dead / stalled)")
        return None

    def check_exchange(self):
        """Raise if the K|V exchange is no longer valid (exchange_gave_up); call per step at most."""
        why = self.exchange_gave_up()
        if why:
            raise RuntimeError(why + "; the result is invalid")

    def _sp_acquire(self):
        """Before the K|V GEMM overwrites the local rows: wait for the peers' pulls of the previous layer (mode "ipc" only)."""
        kg = self.kv_gather
        if hasattr(kg, "acquire"):
            kg.acquire()

    def set_kv_exchange(self, mode: Optional[str], sp_chunks: int):
        """Switch the transport / chunking of the per-layer K|V exchange on a prepared engine (start-up autotune,
        seqpar.autotune_kv_exchange): same workspace, new chunk bounds and a new KVGather on the same group."""
        if not self.sp_on:
            return self
        n = self.plan.n_tok
        old = self.kv_gather
        if hasattr(old, "close"):
            old.close()
        base_mode, arrival = split_kv_mode(mode)
        self.kv_gather = KVGather(self.plan, getattr(old, "group", None), base_mode)
        self._sp_set_arrival(arrival)
        self.sp_bounds = chunk_bounds(n, sp_chunks, self.sp_align)
        self._sp_local_rows()
        if self.attn_fp8:      # the e4m3 K/V side of the workspace is sized for the largest gathered chunk
            kv_rows = self.plan.world * max(b1 - b0 for b0, b1 in zip(self.sp_bounds[:-1], self.sp_bounds[1:]))
            self.attn8_ws = self.ops.attention_fp8_buffers(n, kv_rows, self.cfg.dim, self.cfg.num_heads)
        if getattr(self, "_native", None) is not None:      # the C driver's context holds the old bounds / communicator
            self.ops.lib.icv_dit_destroy(self._native)
            self._native = None
        return self

    def _native_ctx(self):
        """icv_dit context bound to this engine's weights and workspace (built lazily, once per prepare())."""
        if self._native is not None:
            return self._native
        import ctypes
        from .. import native
        lib, cfg, plan, g = self.ops.lib, self.cfg, self.plan, self.grid
        c = native.DitConfig(dim=cfg.dim, ffn_dim=cfg.ffn_dim, heads=cfg.num_heads, layers=cfg.num_layers, n_tok=plan.n_tok,
                             tok0=plan.tok0, T=g.T, Hp=g.Hp, Wp=g.Wp, k_patch=self.k_patch, out_cols=cfg.out_dim * cfg.patch_elems,
                             eps=cfg.eps)
        h = ctypes.c_void_p()
        native.check(lib.icv_dit_create(ctypes.byref(c), ctypes.byref(h)), "icv_dit_create")
        try:
            self._native_bind_all(h)
        except BaseException:
            lib.icv_dit_destroy(h)          # a failed bind / icv_dit_set_seqpar must not leak the context
            raise
        self._native = h
        return h

    def _native_bind_all(self, h):
        import ctypes
        from .. import native
        lib, plan = self.ops.lib, self.plan

        def bind(name, t, layer=-1):
            native.check(lib.icv_dit_bind(h, name.encode(), layer, t.data_ptr()), f"icv_dit_bind({name})")

        for name in ("patch_w", "patch_b", "head_w", "head_b", "x", "x_stem", "h", "qkv", "att", "ff", "patches"):
            bind(name, getattr(self, name))
        bind("rope", self.rope.table)
        for i, lw in enumerate(self.layers):
            for name in ("bqkv", "nq", "nk", "bo", "n3w", "n3b", "xq_b", "xnq", "xo_b", "f0_b", "f2_b"):
                bind(name, lw[name], i)
            for name in self.FP8_WEIGHTS:          # bf16 rows, or e4m3 rows + per-output-row scales under "<name>_s"
                w = lw[name]
                if isinstance(w, tuple):
                    bind(name, w[0], i)
                    bind(name + "_s", w[1], i)
                else:
                    bind(name, w, i)
        if self.fp8:
            for name in ("h8", "h8s", "att8", "att8s", "ff8", "ff8s"):
                bind(name, getattr(self, name))
        if self.attn8_ws is not None:
            for name, t in zip(("a8_qq", "a8_kq", "a8_vt", "a8_amax"), self.attn8_ws):
                bind(name, t)
            native.check(lib.icv_dit_set_fp8(h, 1), "icv_dit_set_fp8")
        if self.sp_on:
            from .seqpar import _NativeComm
            kg = self.kv_gather
            nc = getattr(kg, "_native", None)
            if nc is None:     # the Python driver's exchange runs in torch.distributed: the C driver brings its own communicator
                nc = _NativeComm.for_group(getattr(kg, "dist", None), kg.group, kg.peers, plan.rank, plan.world)
            self._native_comm = nc                      # keep it (and its side stream) alive as long as the context
            nc.add_dependent(self)                      # ... and let it take the context down with it (_NativeComm.close)
            for name in ("kv_loc", "kv_full", "sp_acc", "sp_ml"):
                bind(name, getattr(self, name))
            b = (ctypes.c_int64 * len(self.sp_bounds))(*self.sp_bounds)
            native.check(lib.icv_dit_set_seqpar(h, nc.handle, plan.world, len(self.sp_bounds) - 1, b, nc.stream.cuda_stream), "icv_dit_set_seqpar")

    def _drop_native_context(self):
        """The communicator the C driver's context points at is going away (seqpar._NativeComm.close): destroy the context; the
        next native forward rebuilds it against a live communicator or fails in icv_comm_create, never in a freed handle."""
        h, self._native = getattr(self, "_native", None), None
        self._native_comm = None
        if h is not None:
            try:
                if self._is_gpu():
                    torch.cuda.synchronize(self.ops.device)
                self.ops.lib.icv_dit_destroy(h)
            except Exception:  # pragma: no cover - interpreter shutdown
                pass

    def _native_eligible(self) -> bool:
        # the C driver knows the RCCL transport, bf16 rows on the wire and the chunked carried-state launches - nothing newer: with e4m3
        # on the wire, the copy-engine transport or the arrival-driven attention the per-op driver runs (it used to take
        # the C path and silently re-quantise gathered bf16 rows per chunk)
        sp_ok = not self.sp_on or (isinstance(self.kv_gather, KVGather) and self.kv_gather.mode != "ipc"
                                   and not getattr(self, "fp8_wire", False) and not getattr(self, "attn_arrival", False))
        return self.native_forward and self._is_gpu() and hasattr(self.ops, "lib") and sp_ok

    def native_profile(self, enable: bool):
        """Time every self-attention launch of the native forward with HIP events on the launch stream (bench.py)."""
        from .. import native
        native.check(self.ops.lib.icv_dit_profile(self._native_ctx(), int(bool(enable))), "icv_dit_profile")

    def native_profile_read(self):
        """(summed ms, launches) of the self-attention launches recorded since the last read; waits for them."""
        import ctypes
        from .. import native
        ms, n = ctypes.c_double(), ctypes.c_int64()
        native.check(self.ops.lib.icv_dit_profile_read(self._native_ctx(), ctypes.byref(ms), ctypes.byref(n)), "icv_dit_profile_read")
        return ms.value, n.value

    def __del__(self):
        h = getattr(self, "_native", None)
        if h is not None:
            try:
                self.ops.lib.icv_dit_destroy(h)
            except Exception:  # pragma: no cover - interpreter shutdown
                pass

    def _is_gpu(self) -> bool:
        dev = getattr(self.ops, "device", None)
        return dev is not None and torch.device(dev).type == "cuda"

    def _sp_start_gather(self, kv_loc=None, kv_full=None, kv8=None, start: bool = True):
        """K13: enqueue the exchange of every K|V row-chunk (RCCL runs them back to back on its own
        stream; chunk c = rows [r0, r1) of EVERY rank's shard, rank-major).
        e4m3 on the wire (``fp8_wire``; ``kv8`` = this branch's wire set): the per-head abs-max of the local K and V rows is
        max-reduced over the group first (2 x H floats, one tiny collective per layer), every rank quantises its own rows of
        each chunk ONCE with those scales - the scales of the unsharded launch, so the e4m3 values are the single-GPU ones -
        and the exchange moves the e4m3 blobs: half the bytes of bf16 rows, 1/world of the quantise work.
        ``start=False`` (bench.py's attention-from-memory probe): only describe the buffers of the LAST exchange, move nothing."""
        world, b, d = self.plan.world, self.sp_bounds, self.cfg.dim
        kv_loc = self.kv_loc if kv_loc is None else kv_loc
        handles, bufs = [], []
        if getattr(self, "fp8_wire", False):
            loc8, full8, amax = self.kv8 if kv8 is None else kv8
            H = self.cfg.num_heads
            if start:
                self.ops.attention_fp8_kv_amax(kv_loc[:, :d], kv_loc[:, d:], H, amax)
                self.kv_gather.allreduce_max(amax[1:3])
            for off8, rows8, r0, r1 in self._kv8_chunks:
                blob = loc8[off8: off8 + rows8]
                if start:
                    self.ops.attention_fp8_quantize_kv(kv_loc[r0:r1, :d], kv_loc[r0:r1, d:], H, amax, blob.view(-1))
                full = full8[world * off8: world * (off8 + rows8)]
                bufs.append((full, r1 - r0, amax, blob))
                handles.append(self.kv_gather.start(blob, full) if start else ())
            return handles, bufs
        kv_full = self.kv_full if kv_full is None else kv_full
        bufs = _ChunkBufs()
        bufs.own = kv_loc
        for c in range(len(b) - 1):
            r0, r1 = b[c], b[c + 1]
            full = kv_full[world * r0: world * r1]
            bufs.append((full[:, :d], full[:, d:]))            # strided views: the kernels take a row stride
            handles.append(self.kv_gather.start(kv_loc[r0:r1], full) if start else ())
        return handles, bufs

    def _sp_attention(self, q, handles, bufs, H, scale, att=None, from_memory: bool = False):
        """K6 pipelined with K13: attention consumes chunk c as soon as it has landed, carrying the
        online-softmax state in fp32 between launches, while later chunks are still in flight.
        ``from_memory`` (bench.py's probe): the same launches over rows that are already there - no wait, no flag."""
        ops = self.ops
        att = self.att if att is None else att
        C = len(bufs)
        if getattr(self, "attn_arrival", False) and not getattr(self, "fp8_wire", False):
            # ONE launch over the pieces: this rank's own rows (in place, no flag) first, then every (row chunk, peer) in the order the
            # exchange delivers them - chunk-major, peers starting with the right-hand neighbour (the order the pulls are issued in)
            d, kg, world, rank = self.cfg.dim, self.kv_gather, self.plan.world, self.plan.rank
            own = bufs.own
            pieces, flags = [(own[:, :d], own[:, d:], -1, 0)], None
            for c in range(C):
                kf, vf = bufs[c]
                m = kf.shape[0] // world
                fl, entries = (None, [(j, -1, 0) for j in range(world) if j != rank]) if from_memory else kg.arrival(handles[c])
                flags = fl if fl is not None else flags
                for j, idx, val in sorted(entries, key=lambda e: (e[0] - rank) % world):
                    pieces.append((kf[j * m:(j + 1) * m], vf[j * m:(j + 1) * m], idx if fl is not None else -1, val))
            ops.attention_pieces(q, pieces, att, H, scale, flags=flags, err=self.sp_err, timeout_us=self.sp_timeout_us)
            for c in range(C if not from_memory else 0):
                kg.consumed(handles[c])
            return
        wire = getattr(self, "fp8_wire", False)
        ws = self.attn8_ws
        if ws is not None:
            if wire:
                ws = ops.attention_fp8_with_amax(ws, bufs[0][2])     # this branch's abs-max table (queries' row written here)
            ops.attention_fp8_prepare(ws, H, q=q)                       # queries once per layer, under the first transfer
        gated = wire and getattr(self, "attn_arrival", False)
        for c in range(C):
            if not from_memory and not gated:
                self.kv_gather.wait(handles[c])
            if wire:
                full, m, amax, own_blob = bufs[c]
                gate = None
                if gated:
                    # the chunk's launch starts NOW: this rank's blob first (read where it was quantised), then the peers' in the order the
                    # pulls were issued, each behind its arrival flag inside the kernel
                    kg, world, rank = self.kv_gather, self.plan.world, self.plan.rank
                    fl, entries = (None, [(j, -1, 0) for j in range(world) if j != rank]) if from_memory else kg.arrival(handles[c])
                    seq = [(rank, -1, 0)] + [(j, idx if fl is not None else -1, val) for j, idx, val in sorted(entries, key=lambda e: (e[0] - rank) % world)]
                    gate = dict(seq=seq, flags=fl, own=(own_blob.view(-1), rank), err=self.sp_err, timeout_us=self.sp_timeout_us)
                ops.attention_fp8_pieces(ws, amax, full.view(-1), m, self.plan.world, q.shape[0], att, self.sp_acc, self.sp_ml, H,
                                         first=(c == 0), last=(c == C - 1), gate=gate)
                if gated and not from_memory:
                    self.kv_gather.consumed(handles[c])
            elif ws is not None:
                ops.attention_fp8_prepare(ws, H, k=bufs[c][0], v=bufs[c][1])
                ops.attention_fp8_chunk(ws, q.shape[0], bufs[c][0].shape[0], att, self.sp_acc, self.sp_ml, H,
                                        first=(c == 0), last=(c == C - 1))
            else:
                ops.attention_chunk(q, bufs[c][0], bufs[c][1], att, self.sp_acc, self.sp_ml, H, scale,
                                    first=(c == 0), last=(c == C - 1))

    # ------------------------------------------------------------------------------------
    def encode_context(self, context: torch.Tensor, clip_fea: Optional[torch.Tensor] = None) -> ContextKV:
        """text_embedding MLP + every layer's cross-attention K (RMS-normed) and V; once per prompt.
        i2v: ``clip_fea`` [img_len, img_dim] additionally yields the image K/V set."""
        cfg, ops = self.cfg, self.ops
        if cfg.has_image_input != (clip_fea is not None):
            raise ValueError("clip_fea must be given exactly when the DiT has the image branch (i2v)")
        d, L = cfg.dim, cfg.num_layers
        ctx = ops.to_device(context, BF16)
        n = ctx.shape[0]
        t1 = ops.alloc((n, d), BF16)
        emb = ops.alloc((n, d), BF16)
        ops.gemm(ctx, self.text0_w, self.text0_b, t1, EPI_GELU_BF16)
        ops.gemm(t1, self.text2_w, self.text2_b, emb, EPI_BF16)
        kv = ops.alloc((L, 2, n, d), BF16)
        for i, lw in enumerate(self.layers):
            ops.gemm(emb, lw["xkv_w"], lw["xkv_b"], kv[i], EPI_BF16, nsplit=d)
            ops.rmsnorm_rope(kv[i, 0], lw["xnk"], eps=cfg.eps)
        if clip_fea is None:
            return ContextKV(kv[:, 0], kv[:, 1])
        img = self._image_embed(clip_fea)
        m = img.shape[0]
        kvi = ops.alloc((L, 2, m, d), BF16)
        for i, lw in enumerate(self.layers):
            ops.gemm(img, lw["xkv_img_w"], lw["xkv_img_b"], kvi[i], EPI_BF16, nsplit=d)
            ops.rmsnorm_rope(kvi[i, 0], lw["xnk_img"], eps=cfg.eps)
        return ContextKV(kv[:, 0], kv[:, 1], kvi[:, 0], kvi[:, 1])

    def _image_embed(self, clip_fea: torch.Tensor) -> torch.Tensor:
        """img_emb ([EXT] Wan2.1 MLPProj): LayerNorm, Linear, GELU(erf), Linear, LayerNorm -> bf16 [img_len, d].
        Once per generation on 257 rows: the two GEMMs run in libicvideo, the two LayerNorms (torch eps 1e-5)
        and the erf-GELU are stock device-side torch ops like the encoders outside the loop."""
        import torch.nn.functional as F
        cfg, ops = self.cfg, self.ops
        x = ops.to_device(clip_fea, F32)
        if tuple(x.shape) != (cfg.img_len, cfg.img_dim):
            raise ValueError(f"clip_fea shape {tuple(x.shape)} != {(cfg.img_len, cfg.img_dim)}")
        h = F.layer_norm(x, (cfg.img_dim,), self.img_ln0[0], self.img_ln0[1], 1e-5).to(BF16)
        t1 = ops.alloc((cfg.img_len, cfg.img_dim), F32)
        ops.gemm(h, self.img1_w, self.img1_b, t1, EPI_F32)
        t2 = ops.alloc((cfg.img_len, cfg.dim), F32)
        ops.gemm(F.gelu(t1).to(BF16), self.img3_w, self.img3_b, t2, EPI_F32)
        return F.layer_norm(t2, (cfg.dim,), self.img_ln4[0], self.img_ln4[1], 1e-5).to(BF16)

    def embed_cond_latents(self, y: torch.Tensor, add_to: Optional[torch.Tensor] = None) -> torch.Tensor:
        """i2v: tokens of the step-invariant conditioning latent y [in_dim-16, T, H8, W8] under the y columns
        of the patch embedding (no bias — that stays in the per-step GEMM), f32 [n, d]; ``add_to`` (e.g. the
        guidance-buffer tokens) is accumulated so the result is the one cached additive term of K1."""
        if self.cond_w is None:
            raise RuntimeError("this DiT has no conditioning-latent channels (in_dim == out_dim)")
        ops, plan, cfg = self.ops, self.plan, self.cfg
        yl = ops.to_device(y, F32).contiguous()
        if yl.shape[0] != cfg.cond_channels:
            raise ValueError(f"conditioning latent has {yl.shape[0]} channels, expected {cfg.cond_channels}")
        pt = torch.zeros((plan.n_tok, self.k_cond), dtype=BF16, device=ops.device)
        ops.patchify(yl, pt, plan.tok0, plan.n_tok)
        out = ops.alloc((plan.n_tok, cfg.dim), F32)
        if add_to is None:
            ops.gemm(pt, self.cond_w, None, out, EPI_F32)
        else:
            ops.gemm(pt, self.cond_w, None, out, EPI_RESID_F32, resid=add_to)
        return out

    def embed_buffers(self, buffer_latents: torch.Tensor) -> torch.Tensor:
        """Guidance-buffer tokens f32 [n, d] for this shard (step-invariant)."""
        if self.buffer_embedder is None:
            raise RuntimeError("buffer embedder not initialised")
        ops, plan = self.ops, self.plan
        bl = ops.to_device(buffer_latents, F32)
        out = ops.alloc((plan.n_tok, self.cfg.dim), F32)
        c0 = 0
        for j, cv in enumerate(self.buffer_embedder):
            part = bl[c0: c0 + cv["cin"]].contiguous()
            c0 += cv["cin"]
            pt = torch.zeros((plan.n_tok, cv["k"]), dtype=BF16, device=ops.device)
            ops.patchify(part, pt, plan.tok0, plan.n_tok)
            if j == 0:
                ops.gemm(pt, cv["w"], cv["b"], out, EPI_F32)
            else:
                ops.gemm(pt, cv["w"], cv["b"], out, EPI_RESID_F32, resid=out)
        if c0 != bl.shape[0]:
            raise ValueError(f"buffer latents have {bl.shape[0]} channels, embedder consumes {c0}")
        return out

    # ------------------------------------------------------------------------------------
    def _time_state(self, timestep: float):
        """t = MLP(sinus(timestep)); t_mod = Linear(SiLU(t)); per-layer modulation tables (fp32)."""
        if self._t_cached == timestep:
            return
        ops = self.ops
        ops.sinusoidal(timestep, self.t_sin)
        ops.gemv(self.t_sin, self.time0_w, self.time0_b, self.t_a, 0, ACT_SILU)
        ops.gemv(self.t_a, self.time2_w, self.time2_b, self.t_e, 0, 0)
        ops.gemv(self.t_e, self.tproj_w, self.tproj_b, self.t_mod, ACT_SILU, 0)
        ops.bcast_add(self.modulation, self.t_mod, self.mod)      # [L,6d] + [6d]
        ops.bcast_add(self.head_mod, self.t_e, self.hmod)         # [2,d] + [d]  (note: t, not t_mod)
        self._t_cached = timestep

    def forward_tokens(self, latent: torch.Tensor, ctx: ContextKV, timestep: float,
                       buf_tokens: Optional[torch.Tensor], head_out: torch.Tensor,
                       num_layers: Optional[int] = None, stem: Optional[str] = None):
        """One DiT forward on this rank's token shard: latent f32 [C,T,H8,W8] (full, replicated) ->
        head_out f32 [n, out_dim*4] (velocity in token space).
        ``stem``: the patch embedding and layer 0's self-attention block (LN/modulate, QKV, RoPE, attention, O-projection
        + gated residual) see nothing of the text context, so the cond and uncond forwards of one step compute
        bit-identical values there.  "save" keeps the residual stream after that block in ``self.x_stem``; "load" starts
        from it instead of recomputing (denoise() does this for the second CFG forward: 1/80 of a step's self-attention,
        QKV and O work, same numbers)."""
        cfg, ops, plan = self.cfg, self.ops, self.plan
        d, H, n, eps = cfg.dim, cfg.num_heads, plan.n_tok, cfg.eps
        scale = self.attn_scale          # = ln 2: K already carries (1/sqrt(hd)) * log2(e)
        if (self.cond_w is not None) and buf_tokens is None:
            raise ValueError("i2v DiT: pass the cached embed_cond_latents(y) tokens as buf_tokens")
        if cfg.has_image_input != (ctx.k_img is not None):
            raise ValueError("context was encoded without/with CLIP features but the DiT is/isn't i2v")
        self._time_state(timestep)
        if stem not in (None, "save", "load"):
            raise ValueError(f"stem must be None, 'save' or 'load', got {stem!r}")
        if stem is not None and (num_layers == 0 or cfg.num_layers == 0):
            stem = None
        if self._graphs_on:
            key = (id(ctx), latent.data_ptr(), 0 if buf_tokens is None else buf_tokens.data_ptr(), head_out.data_ptr(), num_layers, stem)
            entry = self._graphs.get(key)
            if entry is None:
                # first call with these buffers: run eagerly (that IS this call's result; it also lets every kernel do
                # its one-time launch-attribute set-up), then record the same launch sequence for the later calls
                self._forward_body(latent, ctx, buf_tokens, head_out, num_layers, stem)
                torch.cuda.synchronize(ops.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._forward_body(latent, ctx, buf_tokens, head_out, num_layers, stem)
                self._graphs[key] = (g, ctx, latent, buf_tokens, head_out)   # keep the captured buffers alive
            else:
                entry[0].replay()
            return
        self._forward_body(latent, ctx, buf_tokens, head_out, num_layers, stem)

    def _forward_body(self, latent, ctx, buf_tokens, head_out, num_layers, stem=None):
        """The launch sequence of one forward after the time state: patch embed, L blocks, head."""
        cfg, ops, plan = self.cfg, self.ops, self.plan
        d, H, n, eps = cfg.dim, cfg.num_heads, plan.n_tok, cfg.eps
        scale = self.attn_scale
        if self._native_eligible():
            from .. import native
            C, _, H8, W8 = latent.shape
            ki, vi = (ctx.k_img, ctx.v_img) if ctx.k_img is not None else (None, None)
            native.check(ops.lib.icv_dit_forward(
                self._native_ctx(), latent.data_ptr(), C, H8, W8, self.mod.data_ptr(), self.hmod.data_ptr(), ctx.k.data_ptr(),
                ctx.v.data_ptr(), ctx.k.shape[1], ctx.k.stride(0), native.ptr(ki), native.ptr(vi), ki.shape[1] if ki is not None else 0,
                ki.stride(0) if ki is not None else 0, native.ptr(buf_tokens), head_out.data_ptr(),
                -1 if num_layers is None else num_layers, {None: 0, "save": 1, "load": 2}[stem], scale, ops._stream()), "icv_dit_forward")
            return
        if stem == "load":
            self.x.copy_(self.x_stem)          # the residual stream after layer 0's self-attention block (see forward_tokens)
        else:
            # K1: patch embed (+ cached guidance-buffer tokens fused into the GEMM epilogue)
            ops.patchify(latent, self.patches, plan.tok0, n)
            if buf_tokens is not None:
                ops.gemm(self.patches, self.patch_w, self.patch_b, self.x, EPI_RESID_F32, resid=buf_tokens)
            else:
                ops.gemm(self.patches, self.patch_w, self.patch_b, self.x, EPI_F32)
        q, k, v = self.qkv[0], self.qkv[1], self.qkv[2]
        L = cfg.num_layers if num_layers is None else num_layers
        for i in range(L):
            lw = self.layers[i]
            m = self.mod[i]
            sh1, sc1, g1 = m[0:d], m[d:2 * d], m[2 * d:3 * d]
            sh2, sc2, g2 = m[3 * d:4 * d], m[4 * d:5 * d], m[5 * d:6 * d]
            # --- self-attention ---
            if i == 0 and stem == "load":
                pass                                                                        # taken from the cond forward
            elif self.sp_on:
                h = self._norm(lw["wqkv"], shift=sh1, scale=sc1, eps=eps)                   # K3
                self._sp_acquire()
                # K and V first, so their all-gather (K13) is already moving while Q is projected
                self._mm(h, lw["wqkv"], lw["bqkv"], self.kv_loc, EPI_BF16, rows=slice(d, 3 * d))            # K4 (k | v rows)
                ops.rmsnorm_rope(self.kv_loc[:, :d], lw["nk"], eps=eps, rope=self.rope, tok0=plan.tok0)      # K5 (k)
                handles, bufs = self._sp_start_gather()
                self._mm(h, lw["wqkv"], lw["bqkv"], q, EPI_BF16, rows=slice(0, d))                   # K4 (q)
                ops.rmsnorm_rope(q, lw["nq"], eps=eps, rope=self.rope, tok0=plan.tok0)               # K5 (q)
                self._sp_attention(q, handles, bufs, H, scale)                                       # K6
            else:
                h = self._norm(lw["wqkv"], shift=sh1, scale=sc1, eps=eps)                   # K3
                self._mm(h, lw["wqkv"], lw["bqkv"], self.qkv, EPI_BF16, nsplit=d)           # K4
                ops.rmsnorm_rope(q, lw["nq"], k, lw["nk"], eps=eps, rope=self.rope, tok0=plan.tok0)  # K5
                if self.attn8_ws is not None:
                    ops.attention_fp8(q, k, v, self.att, H, self.attn8_ws)                  # K6 (e4m3)
                else:
                    ops.attention(q, k, v, self.att, H, scale)                              # K6
            if not (i == 0 and stem == "load"):
                a = self._operand(self.att, self.att8, self.att8s, lw["wo"])
                self._mm(a, lw["wo"], lw["bo"], self.x, EPI_RESID_F32, resid=self.x, gate=g1)   # K7
                if i == 0 and stem == "save":
                    self.x_stem.copy_(self.x)
            # --- cross-attention to text (no gate) ---
            h = self._norm(lw["xq_w"], weight=lw["n3w"], bias=lw["n3b"], eps=eps)           # K8
            self._mm(h, lw["xq_w"], lw["xq_b"], q, EPI_BF16)                                # K9
            ops.rmsnorm_rope(q, lw["xnq"], eps=eps)
            ops.attention(q, ctx.k[i], ctx.v[i], self.att, H, scale)
            if ctx.k_img is not None:                                                      # i2v: + softmax over CLIP tokens
                ops.attention_add(q, ctx.k_img[i], ctx.v_img[i], self.att, H, scale)
            a = self._operand(self.att, self.att8, self.att8s, lw["xo_w"])
            self._mm(a, lw["xo_w"], lw["xo_b"], self.x, EPI_RESID_F32, resid=self.x)
            # --- FFN ---
            h = self._norm(lw["f0_w"], shift=sh2, scale=sc2, eps=eps)                       # K3
            self._mm(h, lw["f0_w"], lw["f0_b"], self.ff, EPI_GELU_BF16)                     # K10
            a = self._operand(self.ff, self.ff8, self.ff8s, lw["f2_w"])
            self._mm(a, lw["f2_w"], lw["f2_b"], self.x, EPI_RESID_F32, resid=self.x, gate=g2)
        # K11: head
        ops.ln_modulate(self.x, self.h, shift=self.hmod[0], scale=self.hmod[1], eps=eps)
        ops.gemm(self.h, self.head_w, self.head_b, head_out, EPI_F32)

    # ------------------------------------------------------------------------------------
    # CFG-batched forward pair (single rank): the cond and uncond forwards of a step as ONE batch of 2n rows through every
    # token-local kernel (LayerNorm, the six projections, the head), two launches only where the branches differ (self-attention:
    # own K / V; RoPE offsets; cross-attention: own context).  Per row the arithmetic is unchanged - the same kernels, the same
    # K order - so the pair is BIT-IDENTICAL to two sequential forwards (tests); what changes is tile quantisation: at
    # S = 37 440 the four N = 5120 GEMMs of a layer are 2940 tiles = 11.48 rounds of 256 CUs (4.3 % of their time is an empty
    # half round), as 2S rows they are 22.97 rounds, and each weight matrix is read once per step instead of twice.
    PAIR_MAX_OPERAND_BYTES = (1 << 32) - 1     # the GEMM kernels keep 32-bit per-lane byte offsets from the operand base

    def _pair_ok(self) -> bool:
        n2 = 2 * self.plan.n_tok
        return (self.cfg_batch and not self._graphs_on and not self._native_eligible() and not self.dual_stream
                and n2 * max(self.cfg.dim, self.cfg.ffn_dim) * 2 <= self.PAIR_MAX_OPERAND_BYTES)

    def _pair_engine(self):
        """Twin of this engine whose workspace holds 2n rows (weights and caches by reference)."""
        if getattr(self, "_pair", None) is None:
            import copy
            cfg, a, n2, d = self.cfg, self.ops.alloc, 2 * self.plan.n_tok, self.cfg.dim
            t = copy.copy(self)
            t._pair, t._twin, t._native, t._graphs, t._graphs_on, t.native_forward = None, None, None, {}, False, False
            t.x, t.h, t.qkv = a((n2, d), F32), a((n2, d), BF16), a((3, n2, d), BF16)
            t.att, t.ff = a((n2, d), BF16), a((n2, cfg.ffn_dim), BF16)
            if self.fp8:
                t.h8, t.h8s = a((n2, d), FP8), a((n2,), F32)
                t.att8, t.att8s = a((n2, d), FP8), a((n2,), F32)
                t.ff8, t.ff8s = a((n2, cfg.ffn_dim), FP8), a((n2,), F32)
            if self.sp_on:      # both branches' K|V rows: local [2n, 2d] (cond rows, then uncond rows), gathered [2, world*n, 2d]
                t.kv_loc = self._kv_rows(n2)
                t.kv8 = [self._sp_wire_set(self._kv_rows, self.kv8[0].shape[0]) for _ in range(2)] if self.kv8 is not None else None
                t.kv_full = a((2, self.plan.world * self.plan.n_tok, 2 * d), BF16)
            self._pair = t
        return self._pair

    def forward_pair(self, latent: torch.Tensor, ctx_c: ContextKV, ctx_u: ContextKV, timestep: float,
                     buf_tokens: Optional[torch.Tensor], head_out2: torch.Tensor, share_stem: bool = False):
        """Both CFG forwards of a step: head_out2 f32 [2, n, out_dim*4] (slot 0 = cond, 1 = uncond).  ``share_stem``: layer 0's
        self-attention block is computed once on n rows (it sees nothing of the context) and duplicated, as
        forward_tokens(stem="save" / "load") does for sequential forwards."""
        cfg, ops, plan = self.cfg, self.ops, self.plan
        d, H, n, eps, scale = cfg.dim, cfg.num_heads, plan.n_tok, cfg.eps, self.attn_scale
        if (self.cond_w is not None) and buf_tokens is None:
            raise ValueError("i2v DiT: pass the cached embed_cond_latents(y) tokens as buf_tokens")
        for ctx in (ctx_c, ctx_u):
            if cfg.has_image_input != (ctx.k_img is not None):
                raise ValueError("context was encoded without/with CLIP features but the DiT is/isn't i2v")
        self._time_state(timestep)
        t = self._pair_engine()
        t.mod, t.hmod = self.mod, self.hmod
        halves = (slice(0, n), slice(n, 2 * n))
        # K1 once: both branches start from the same tokens
        ops.patchify(latent, self.patches, plan.tok0, n)
        if buf_tokens is not None:
            ops.gemm(self.patches, self.patch_w, self.patch_b, t.x[halves[0]], EPI_RESID_F32, resid=buf_tokens)
        else:
            ops.gemm(self.patches, self.patch_w, self.patch_b, t.x[halves[0]], EPI_F32)
        q, k, v = t.qkv[0], t.qkv[1], t.qkv[2]

        def self_attention_block(lw, sh1, sc1, g1, rows_list, rows_all):
            h = t._norm(lw["wqkv"], rows=rows_all, shift=sh1, scale=sc1, eps=eps)                       # K3
            if self.sp_on:
                # sequence parallel ("sp" layout: this rank runs BOTH forwards of a step on its token shard): the projections run
                # once over the 2n rows of the pair - at n = 4 680 (sp8, 14B) the N = 5120 GEMMs go from 2 x 380 tiles = 2 x 1.48
                # rounds of 256 CUs to 740 tiles = 2.89 rounds - while exchange and attention stay per branch: the uncond rows
                # travel under the cond branch's attention
                rs = slice(0, 2 * n) if rows_all is None else rows_all
                self._sp_acquire()
                t._mm(h, lw["wqkv"], lw["bqkv"], t.kv_loc[rs], EPI_BF16, rows=slice(d, 3 * d))            # K4 (k | v rows)
                pend = []
                for bi, r in enumerate(halves):
                    if r not in rows_list:
                        continue
                    ops.rmsnorm_rope(t.kv_loc[r][:, :d], lw["nk"], eps=eps, rope=self.rope, tok0=plan.tok0)   # K5 (k)
                    pend.append((r,) + self._sp_start_gather(t.kv_loc[r], t.kv_full[bi], t.kv8[bi] if t.kv8 is not None else None))   # K13
                t._mm(h, lw["wqkv"], lw["bqkv"], q[rs], EPI_BF16, rows=slice(0, d))                       # K4 (q)
                for r, handles, bufs in pend:
                    ops.rmsnorm_rope(q[r], lw["nq"], eps=eps, rope=self.rope, tok0=plan.tok0)             # K5 (q)
                    self._sp_attention(q[r], handles, bufs, H, scale, att=t.att[r])                       # K6
                a = t._operand(t.att, t.att8, t.att8s, lw["wo"], rows=rows_all)
                xr = t.x if rows_all is None else t.x[rows_all]
                t._mm(a, lw["wo"], lw["bo"], xr, EPI_RESID_F32, resid=xr, gate=g1)                        # K7
                return
            if rows_all is None:
                t._mm(h, lw["wqkv"], lw["bqkv"], t.qkv, EPI_BF16, nsplit=d)                             # K4, 2n rows
            else:   # n rows into the first half of each plane: three plain GEMMs keep the [3, 2n, d] plane layout
                for j in range(3):
                    t._mm(h, lw["wqkv"], lw["bqkv"], t.qkv[j][rows_all], EPI_BF16, rows=slice(j * d, (j + 1) * d))
            for r in rows_list:
                ops.rmsnorm_rope(q[r], lw["nq"], k[r], lw["nk"], eps=eps, rope=self.rope, tok0=plan.tok0)   # K5
                if self.attn8_ws is not None:
                    ops.attention_fp8(q[r], k[r], v[r], t.att[r], H, self.attn8_ws)                     # K6 (e4m3)
                else:
                    ops.attention(q[r], k[r], v[r], t.att[r], H, scale)                                 # K6
            a = t._operand(t.att, t.att8, t.att8s, lw["wo"], rows=rows_all)
            xr = t.x if rows_all is None else t.x[rows_all]
            t._mm(a, lw["wo"], lw["bo"], xr, EPI_RESID_F32, resid=xr, gate=g1)                          # K7

        for i in range(cfg.num_layers):
            lw = self.layers[i]
            m = self.mod[i]
            sh1, sc1, g1 = m[0:d], m[d:2 * d], m[2 * d:3 * d]
            sh2, sc2, g2 = m[3 * d:4 * d], m[4 * d:5 * d], m[5 * d:6 * d]
            if i == 0 and share_stem:
                self_attention_block(lw, sh1, sc1, g1, [halves[0]], halves[0])
                t.x[halves[1]].copy_(t.x[halves[0]])
            else:
                if i == 0:
                    t.x[halves[1]].copy_(t.x[halves[0]])
                self_attention_block(lw, sh1, sc1, g1, halves, None)
            # --- cross-attention: one projection over 2n rows, each half against its own context ---
            h = t._norm(lw["xq_w"], weight=lw["n3w"], bias=lw["n3b"], eps=eps)                          # K8
            t._mm(h, lw["xq_w"], lw["xq_b"], q, EPI_BF16)                                               # K9
            ops.rmsnorm_rope(q, lw["xnq"], eps=eps)
            for r, ctx in zip(halves, (ctx_c, ctx_u)):
                ops.attention(q[r], ctx.k[i], ctx.v[i], t.att[r], H, scale)
                if ctx.k_img is not None:
                    ops.attention_add(q[r], ctx.k_img[i], ctx.v_img[i], t.att[r], H, scale)
            a = t._operand(t.att, t.att8, t.att8s, lw["xo_w"])
            t._mm(a, lw["xo_w"], lw["xo_b"], t.x, EPI_RESID_F32, resid=t.x)
            # --- FFN over 2n rows ---
            h = t._norm(lw["f0_w"], shift=sh2, scale=sc2, eps=eps)                                      # K3
            t._mm(h, lw["f0_w"], lw["f0_b"], t.ff, EPI_GELU_BF16)                                       # K10
            a = t._operand(t.ff, t.ff8, t.ff8s, lw["f2_w"])
            t._mm(a, lw["f2_w"], lw["f2_b"], t.x, EPI_RESID_F32, resid=t.x, gate=g2)
        # K11: head over 2n rows
        ops.ln_modulate(t.x, t.h, shift=self.hmod[0], scale=self.hmod[1], eps=eps)
        ops.gemm(t.h, self.head_w, self.head_b, head_out2.view(2 * n, -1), EPI_F32)

    # ------------------------------------------------------------------------------------
    def _cfg_twin(self):
        """Second engine for the dual-stream CFG mode: shares every weight tensor, owns a workspace and a stream."""
        if getattr(self, "_twin", None) is None:
            import copy
            twin = copy.copy(self)                      # shallow: weights / caches by reference
            twin._twin = None
            twin._native = None                         # its own workspace -> its own native context, if any
            twin.dual_stream = False
            twin.prepare(self.grid, self.plan, graphs=False)
            self._twin = (twin, torch.cuda.Stream(device=self.ops.device))
        return self._twin

    def denoise(self, latent: torch.Tensor, ctx_cond: Optional[ContextKV], ctx_uncond: Optional[ContextKV],
                buf_tokens: Optional[torch.Tensor], scheduler: FlowMatchScheduler,
                cfg_scale: float = 5.0, steps: Optional[range] = None, on_step=None,
                branch_exchange=None, round_bf16: bool = False) -> torch.Tensor:
        """The hot loop: per step 2 DiT forwards (cond, uncond) + fused unpatchify/CFG/Euler.
        ``latent`` f32 [C,T,H8,W8] is updated IN PLACE for this rank's tokens.
        ``branch_exchange`` (seqpar.BranchExchange, cfg+sp layout): this rank runs ONE forward per step — the
        cond one if it was given ``ctx_cond`` only, the uncond one if ``ctx_uncond`` only — and swaps velocity
        tokens with the rank that runs the other branch on the same token shard.
        ``round_bf16``: "reference rounding" of the CFG combine and the Euler update (icv_unpatchify_cfg_euler)."""
        ops, plan = self.ops, self.plan
        if branch_exchange is not None:
            if (ctx_cond is None) == (ctx_uncond is None) or cfg_scale == 1.0:
                raise ValueError("cfg+sp: pass exactly one of ctx_cond / ctx_uncond and a cfg_scale != 1")
            own_ctx = ctx_cond if ctx_cond is not None else ctx_uncond
            for i in (steps if steps is not None else range(len(scheduler.sigmas))):
                self.forward_tokens(latent, own_ctx, scheduler.timesteps[i], buf_tokens, self.head_own)
                branch_exchange(self.head_own, self.head_out)           # slot 0 = cond, slot 1 = uncond
                ops.unpatchify_cfg_euler(latent, self.head_out[0], self.head_out[1], cfg_scale,
                                         scheduler.dsigma(i), plan.tok0, plan.n_tok, round_bf16=round_bf16)
                if self.sp_on:
                    self._check_transport()
                if on_step is not None:
                    on_step(i, latent)
            if self.sp_on:
                self.check_exchange()
            return latent
        use_cfg = ctx_uncond is not None and cfg_scale != 1.0
        twin, side = self._cfg_twin() if (use_cfg and self.dual_stream) else (None, None)
        for i in (steps if steps is not None else range(len(scheduler.sigmas))):
            ts = scheduler.timesteps[i]
            if twin is not None:
                # the two CFG forwards are independent: the uncond one runs on a second HIP stream in a twin engine
                # (same weights, own workspace), so the partial last wave of blocks of every kernel of one branch is
                # filled by the other branch's kernels
                main = torch.cuda.current_stream(ops.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    twin.forward_tokens(latent, ctx_uncond, ts, buf_tokens, self.head_out[1])
                self.forward_tokens(latent, ctx_cond, ts, buf_tokens, self.head_out[0])
                main.wait_stream(side)
            elif use_cfg and self._pair_ok():
                self.forward_pair(latent, ctx_cond, ctx_uncond, ts, buf_tokens, self.head_out, share_stem=self.share_stem)
            else:
                share = use_cfg and self.share_stem
                self.forward_tokens(latent, ctx_cond, ts, buf_tokens, self.head_out[0], stem="save" if share else None)
                if use_cfg:
                    self.forward_tokens(latent, ctx_uncond, ts, buf_tokens, self.head_out[1], stem="load" if share else None)
            ops.unpatchify_cfg_euler(latent, self.head_out[0], self.head_out[1] if use_cfg else None,
                                     cfg_scale, scheduler.dsigma(i), plan.tok0, plan.n_tok, round_bf16=round_bf16)
            if self.sp_on:
                self._check_transport()
            if on_step is not None:
                on_step(i, latent)
        if self.sp_on:
            self.check_exchange()
        return latent
