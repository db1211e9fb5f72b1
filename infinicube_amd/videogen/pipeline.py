"""``WanVideoPipeline`` / ``ModelConfig`` — the surface the reference reaches into diffsynth for
(SURVEY.md §8a-2 rows D1-D5), re-implemented MI355X-native.

Pinned by the reference's call sites:
  * ``ModelConfig(model_id=, origin_file_pattern=, skip_download=True)``        [R infinicube/videogen/inference.py:67-69,77-79]
  * ``WanVideoPipeline.from_pretrained(torch_dtype=, device=, model_configs=)`` [R infinicube/videogen/inference.py:63,73]
  * ``pipe.dit`` / ``pipe.buffer_embedder`` with ``load_state_dict``            [R infinicube/videogen/inference.py:106-127]
  * ``pipe.initialize_buffer_embedder(buffer_channels=16, zero_init=True)``     [R infinicube/videogen/inference.py:86-88]
  * ``pipe.enable_vram_management()``                                           [R infinicube/videogen/inference.py:97]
  * ``pipe(prompt=, negative_prompt=, semantic_buffer_video=, coordinate_buffer_video=, height=,
    width=, num_frames=, seed=, tiled=) -> List[PIL.Image]``                    [R infinicube/videogen/inference.py:216-226]
Everything the fork does *inside* those calls is [EXT]: 50 flow-match steps,
shift 5, cfg 5, CPU-generator noise, VAE-encode the two buffer videos, embed, add to the tokens.

The denoising loop runs in libicvideo's HIP kernels (dit.WanDiT + ops.HipOps).  The text encoder
and the VAE are outside the hot loop and stay on stock PyTorch-ROCm modules behind two tiny
interfaces (``encode(prompt) -> [text_len, text_dim]``; ``encode/decode`` video<->latent).
"""

from __future__ import annotations

import os
from collections import namedtuple
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from PIL import Image

from .config import TokenGrid, WanDiTConfig, infer_config_from_state_dict
from .io import load_sharded_state_dict
from .scheduler import FlowMatchScheduler
from .seqpar import BranchExchange, ParallelLayout, gather_latent

_IncompatibleKeys = namedtuple("_IncompatibleKeys", ["missing_keys", "unexpected_keys"])


class ModelConfig:
    """Where a model file lives: ``models/<model_id>/<origin_file_pattern>`` (diffsynth's layout,
    [R README.md:33]) or an explicit ``path``.  ``skip_download=True`` (what the generator passes
    [R infinicube/videogen/inference.py:67-69]) never touches the network; with ``skip_download=False`` (the
    reference's download script [R infinicube/videogen/download_checkpoint.py:19-31]) a file that is missing locally is
    fetched into that layout first (``download()``: ModelScope when importable, like diffsynth's default, else the
    Hugging Face hub)."""

    def __init__(self, path=None, model_id: Optional[str] = None, origin_file_pattern: Optional[str] = None,
                 skip_download: bool = False, offload_device=None, offload_dtype=None, local_model_path: Optional[str] = None,
                 **unused):
        self.path = path
        self.model_id = model_id
        self.origin_file_pattern = origin_file_pattern
        self.skip_download = skip_download
        self.offload_device = offload_device
        self.offload_dtype = offload_dtype
        self.local_model_path = local_model_path

    def resolve(self) -> str:
        if self.path is not None:
            return self.path if isinstance(self.path, str) else self.path[0]
        root = self.local_model_path or os.environ.get("ICV_MODEL_ROOT", "models")
        return os.path.join(root, self.model_id or "", self.origin_file_pattern or "")

    def present(self) -> bool:
        """True when the file(s) the pattern stands for are all there.  A sharded checkpoint
        (``diffusion_pytorch_model*.safetensors``) is complete only when every shard its ``*.index.json`` weight map names
        exists — one shard of an interrupted download must not count as "present" (it would skip the download and load
        half a model)."""
        import glob
        import json
        files = glob.glob(self.resolve())
        if not files:
            return False
        folder = os.path.dirname(files[0])
        for idx in glob.glob(os.path.join(folder, "*.index.json")):
            try:
                with open(idx) as f:
                    shards = set(json.load(f).get("weight_map", {}).values())
            except (OSError, ValueError):
                continue
            import fnmatch
            mine = {s for s in shards if fnmatch.fnmatch(s, os.path.basename(self.resolve()))}
            if mine and any(not os.path.isfile(os.path.join(folder, s)) for s in mine):
                return False
        # no index file at hand (the reference's pattern does not fetch it): the shard names carry the count themselves
        import re
        for f in files:
            m = re.search(r"-(\d+)-of-(\d+)(\.[A-Za-z0-9]+)$", os.path.basename(f))
            if m:
                stem, total, ext = os.path.basename(f)[: m.start()], int(m.group(2)), m.group(3)
                width = len(m.group(1))
                if any(not os.path.isfile(os.path.join(folder, f"{stem}-{i:0{width}d}-of-{m.group(2)}{ext}")) for i in range(1, total + 1)):
                    return False
        return True

    def download(self) -> str:
        """Fetch ``origin_file_pattern`` of ``model_id`` into ``<root>/<model_id>/`` (no-op when present or when the
        config is an explicit path).  Raises with the hub's own error when the network or the hub package is missing."""
        if self.path is not None or self.present():
            return self.resolve()
        root = self.local_model_path or os.environ.get("ICV_MODEL_ROOT", "models")
        target = os.path.join(root, self.model_id)
        os.makedirs(target, exist_ok=True)
        source = os.environ.get("ICV_DOWNLOAD_SOURCE", "auto").lower()
        fetch = None
        if source in ("auto", "modelscope"):
            try:
                from modelscope import snapshot_download as ms_fetch
                fetch = lambda: ms_fetch(self.model_id, allow_file_pattern=self.origin_file_pattern, local_dir=target)   # noqa: E731
            except ImportError:
                if source == "modelscope":
                    raise
        if fetch is None:
            try:
                from huggingface_hub import snapshot_download as hf_fetch
            except ImportError as e:
                raise FileNotFoundError(
                    f"{self.resolve()!r} is missing (or incomplete) and cannot be downloaded: neither `modelscope` nor "
                    f"`huggingface_hub` is importable ({e}); place the file(s) there or pass skip_download=True to get the "
                    f"plain 'file missing' error from the loader") from e
            fetch = lambda: hf_fetch(repo_id=self.model_id, allow_patterns=[self.origin_file_pattern], local_dir=target)   # noqa: E731
        print(f"Downloading {self.model_id}/{self.origin_file_pattern} -> {target}")
        fetch()
        if not self.present():
            raise FileNotFoundError(f"download of {self.model_id} finished but nothing matches {self.resolve()!r}")
        return self.resolve()

    def __repr__(self):
        return f"ModelConfig(model_id={self.model_id!r}, origin_file_pattern={self.origin_file_pattern!r})"


class DiTHolder:
    """``pipe.dit``: owns the DiT state dict until the first generation packs it into HBM
    (``dit.WanDiT``); supports the reference's partial fine-tune overlay
    ``dit.load_state_dict(state, strict=False)`` [R infinicube/videogen/inference.py:127]."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: Optional[WanDiTConfig] = None):
        self._sd = dict(state_dict)
        self.cfg = cfg or infer_config_from_state_dict(self._sd)
        self.version = 0

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return self._sd

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        missing = [k for k in self._sd if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._sd]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for WanModel: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        for k, v in state_dict.items():
            if k in self._sd:
                if tuple(v.shape) != tuple(self._sd[k].shape):
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(self._sd[k].shape)}")
                self._sd[k] = v
        self.version += 1
        return _IncompatibleKeys(missing, unexpected)


class BufferEmbedder:
    """``pipe.buffer_embedder``: guidance buffers -> tokens added to the noisy tokens [R README.md:65].
    Default layout = hypothesis H1 of SURVEY.md §8a K1: the two VAE-encoded buffers concatenated on
    channels (2 x buffer_channels) -> ``Conv3d(2C -> dim, kernel = stride = (1,2,2))``, zero-initialised.
    ``load_state_dict`` is STRICT, as the reference's call is [R infinicube/videogen/inference.py:113]."""

    def __init__(self, dim: int, buffer_channels: int = 16, zero_init: bool = True, patch=(1, 2, 2), variant: str = "concat"):
        self.dim, self.buffer_channels, self.patch, self.variant = dim, buffer_channels, tuple(patch), variant
        shapes = self._shapes()
        g = torch.Generator().manual_seed(1234)
        self._sd = {k: (torch.zeros(s) if (zero_init or k.endswith("bias")) else torch.randn(s, generator=g) * 0.02)
                    for k, s in shapes.items()}
        self.version = 0

    def _shapes(self):
        d, c, p = self.dim, self.buffer_channels, self.patch
        if self.variant == "concat":
            return {"proj.weight": (d, 2 * c) + p, "proj.bias": (d,)}
        return {"semantic_proj.weight": (d, c) + p, "semantic_proj.bias": (d,),
                "coordinate_proj.weight": (d, c) + p, "coordinate_proj.bias": (d,)}

    def state_dict(self):
        return self._sd

    def parameters(self):
        return list(self._sd.values())

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        keys = set(state_dict)
        if keys != set(self._sd):
            # the fork's real layout is unknown ([EXT]); accept the other known variant if it fits exactly
            other = BufferEmbedder(self.dim, self.buffer_channels, True, self.patch,
                                   "dual" if self.variant == "concat" else "concat")
            if keys == set(other._sd):
                self.variant, self._sd = other.variant, other._sd
            elif strict:
                raise RuntimeError(
                    f"Error(s) in loading state_dict for BufferEmbedder: expected keys {sorted(self._sd)} "
                    f"(or {sorted(other._sd)}), got {sorted(keys)}")
        for k, v in state_dict.items():
            if k in self._sd:
                if tuple(v.shape) != tuple(self._sd[k].shape):
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(self._sd[k].shape)}")
                self._sd[k] = v
        self.version += 1
        return _IncompatibleKeys([], [])


def _video_to_tensor(frames: Sequence[Image.Image], height: int, width: int) -> torch.Tensor:
    """List[PIL RGB] -> f32 [3, F, H, W] in [-1, 1] (diffsynth preprocess_video)."""
    arrs = []
    for im in frames:
        if im.size != (width, height):
            im = im.resize((width, height), Image.BILINEAR)
        arrs.append(np.asarray(im.convert("RGB"), dtype=np.uint8))
    v = torch.from_numpy(np.stack(arrs, 0)).to(torch.float32)
    return (v * (2.0 / 255.0) - 1.0).permute(3, 0, 1, 2).contiguous()


def _video_to_uint8(frames: Sequence[Image.Image], height: int, width: int) -> torch.Tensor:
    """List[PIL RGB] -> uint8 [F, H, W, 3] (host): what a VAE with ``accepts_uint8`` normalises on the device — the same
    numbers as ``_video_to_tensor`` without building a 4x larger float clip on the host."""
    arrs = []
    for im in frames:
        if im.size != (width, height):
            im = im.resize((width, height), Image.BILINEAR)
        arrs.append(np.asarray(im.convert("RGB"), dtype=np.uint8))
    return torch.from_numpy(np.stack(arrs, 0))


def _tensor_to_video(video: torch.Tensor) -> List[Image.Image]:
    """f32 [3, F, H, W] in [-1, 1] -> List[PIL RGB] (diffsynth vae_output_to_video)."""
    v = ((video.float().clamp(-1, 1) + 1.0) * 127.5).round().to(torch.uint8).permute(1, 2, 3, 0).cpu().numpy()
    return [Image.fromarray(f, mode="RGB") for f in v]


class WanVideoPipeline:
    def __init__(self, device="cuda:0", torch_dtype=torch.bfloat16, dit: Optional[DiTHolder] = None,
                 text_encoder=None, vae=None, ops=None, image_encoder=None):
        self.device = device
        self.image_encoder = image_encoder   # i2v only (CLIP ViT-H/14 tokens of the conditioning image)
        # diffsynth takes torch_dtype=float8_e4m3fn to mean "fp8 DiT weights"; here it selects the fp8 MFMA
        # projections (BASELINE.json config #5).  Anything else = bf16, the reference's setting
        # [R infinicube/inference/guidance_buffer_generation.py:762].
        self.gemm_dtype = "fp8" if torch_dtype == torch.float8_e4m3fn else "bf16"
        # the fp8 mode also runs self-attention in e4m3 (60.1 dB alone at 1.3B depth, 47.1 dB at 14B depth) and quantises
        # dit.WanDiT.FP8_DEFAULT = the QKV projection: the largest set that stays >= 40 dB at the real 14B depth (see there)
        self.attn_dtype = self.gemm_dtype
        # multi-GPU layout when torch.distributed is initialised (seqpar.ParallelLayout): "auto" | "sp" | "cfg+sp"
        self.parallelism = os.environ.get("ICV_PARALLELISM", "auto")
        self.sp_chunks = int(os.environ.get("ICV_SP_CHUNKS", "4"))       # K/V exchange chunks per layer (overlap depth)
        # K|V transport of the sequence-parallel path (seqpar.KVGather): "allgather" | "p2p" | "native" | "ipc", or "auto" = a
        # start-up autotune on the first call (seqpar.autotune_kv_exchange: two real layers per candidate, the ranks agree on
        # the fastest; cached per layout).  Unset: "auto" on RCCL ranks (what the transports cost on a given node is not
        # known before first contact), "allgather" elsewhere (gloo: CPU tests, ranks sharing one GPU).
        self.kv_exchange = os.environ.get("ICV_KV_EXCHANGE") or None
        self._kv_tuned = {}
        self._layouts = {}
        self.torch_dtype = torch_dtype
        self.dit = dit
        self.text_encoder = text_encoder
        self.vae = vae
        self.buffer_embedder: Optional[BufferEmbedder] = None
        # sampling defaults of the fork's __call__ ([EXT]); the reference never
        # overrides them [R infinicube/videogen/inference.py:216-226], so they are attributes here
        self.num_inference_steps, self.cfg_scale, self.sigma_shift = 50, 5.0, 5.0
        # ICV_REFERENCE_ROUNDING=1 / pipe.reference_rounding = True: reproduce the rounding points of a pipeline that keeps
        # timestep, noise, noise_pred and latents in torch_dtype=bf16 ([EXT] upstream DiffSynth; ORACLE_RISKS.md R1-R3).
        # Default False: exact float timestep, fp32 noise / latents / CFG / Euler (more accurate; not what a bf16
        # checkpoint was sampled with during fine-tuning validation).
        self.reference_rounding = os.environ.get("ICV_REFERENCE_ROUNDING", "0") == "1"
        self.scheduler = FlowMatchScheduler(self.num_inference_steps, self.sigma_shift, self.reference_rounding)
        self._ops = ops
        self._engine = None
        self._engine_key = None
        self.vram_management_enabled = False

    # ---- D2 --------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, torch_dtype=torch.bfloat16, device="cuda", model_configs: Sequence[ModelConfig] = (),
                        tokenizer_config: Optional[ModelConfig] = None, **unused) -> "WanVideoPipeline":
        """Load DiT / UMT5 / Wan-VAE from local files; a config built WITHOUT skip_download=True whose file is missing is
        downloaded first (ModelConfig.download), as diffsynth does for the reference's download script.  Files are
        recognised by name, like the reference's three patterns [R infinicube/videogen/inference.py:67-69]."""
        for mc in model_configs:
            if not getattr(mc, "skip_download", True) and mc.model_id and not mc.present():
                mc.download()
        pipe = cls(device=device, torch_dtype=torch_dtype)
        # torch_dtype=float8_e4m3fn selects the DiT's fp8 MFMA mode only; the encoders outside the loop are stock
        # torch modules that cannot run on unscaled e4m3 weights, so they load in bf16
        aux_dtype = torch.bfloat16 if torch_dtype == torch.float8_e4m3fn else torch_dtype
        for mc in model_configs:
            pattern = mc.resolve()
            base = os.path.basename(pattern)
            if "clip" in base.lower():
                from .clip_vision import load_clip_vision
                pipe.image_encoder = load_clip_vision(pattern, device, aux_dtype)
            elif "t5" in base.lower():
                from .text_encoder import load_umt5_encoder
                pipe.text_encoder = load_umt5_encoder(pattern, device, aux_dtype, tokenizer_config)
            elif "vae" in base.lower():
                from .vae import load_wan_vae
                pipe.vae = load_wan_vae(pattern, device)
            else:
                pipe.dit = DiTHolder(load_sharded_state_dict(pattern))
        if pipe.dit is None:
            raise FileNotFoundError("from_pretrained: no DiT checkpoint among model_configs")
        return pipe

    # ---- D3 / D4 -----------------------------------------------------------------------------
    def initialize_buffer_embedder(self, buffer_channels: int = 16, zero_init: bool = True):
        self.buffer_embedder = BufferEmbedder(self.dit.cfg.dim, buffer_channels, zero_init, self.dit.cfg.patch)
        return self.buffer_embedder

    def enable_vram_management(self, **unused):
        """diffsynth offloads layers to host RAM to fit an 80 GB A100 [R infinicube/videogen/inference.py:95-97].
        With 288 GB of HBM3E the whole 14B DiT (28 GB bf16) + UMT5 + VAE stay resident: a no-op
        with no numeric effect."""
        self.vram_management_enabled = True

    # ---- engine ------------------------------------------------------------------------------
    def _get_ops(self):
        if self._ops is None:
            from .ops import HipOps
            self._ops = HipOps(self.resolve_device(self.device))   # raises loudly without GPU / native library: no fallback
        return self._ops

    @staticmethod
    def resolve_device(device) -> str:
        """One process drives one GPU.  The reference's caller passes the literal ``"cuda:0"``
        [R infinicube/inference/guidance_buffer_generation.py:761]; when this process is one rank of a multi-GPU job
        (LOCAL_RANK set by torch.distributed.run or by multigpu.WorkerPool) that literal means "my GPU", i.e.
        ``cuda:LOCAL_RANK`` - otherwise every rank would land on GPU 0."""
        dev = str(device)
        if not dev.startswith("cuda"):
            return dev
        lr = os.environ.get("LOCAL_RANK")
        if lr is not None and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            return f"cuda:{int(lr)}"
        return "cuda:0" if dev == "cuda" else dev

    def _get_engine(self):
        from .dit import WanDiT
        key = (self.dit.version, self.buffer_embedder.version if self.buffer_embedder else -1, self.gemm_dtype, self.attn_dtype)
        if self._engine is None or self._engine_key != key:
            cfg = self.dit.cfg
            if self.buffer_embedder is not None and cfg.buffer_channels != self.buffer_embedder.buffer_channels:
                import dataclasses
                cfg = dataclasses.replace(cfg, buffer_channels=self.buffer_embedder.buffer_channels)
            self._engine = WanDiT(cfg, self.dit.state_dict(), self._get_ops(),
                                  self.buffer_embedder.state_dict() if self.buffer_embedder else None,
                                  gemm_dtype=self.gemm_dtype, attn_dtype=self.attn_dtype)
            self._engine_key = key
        return self._engine

    def _autotune_kv(self, engine, latent, ctx, buf_tokens, ops):
        """First sequence-parallel call of this layout: time every K|V transport x {sp_chunks, 2} chunks on two real layers of
        THIS generation's shard and keep the fastest (every rank runs this at the same point and ends with the same choice)."""
        import sys
        import time
        import torch.distributed as dist
        from .seqpar import autotune_kv_exchange
        on_dev = dist.get_backend() == "nccl"

        def sync():
            if torch.device(ops.device).type == "cuda":
                torch.cuda.synchronize(ops.device)
            dist.barrier()

        def reduce_max(vals):
            t = torch.tensor(vals, dtype=torch.float64, device=ops.device if on_dev else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return t.tolist()

        def two_layers():
            engine.forward_tokens(latent, ctx, 500.0, buf_tokens, engine.head_own, num_layers=min(2, engine.cfg.num_layers))

        modes = ("allgather", "p2p", "native", "ipc") if on_dev else ("allgather", "p2p")
        # per transport: ONE arrival-gated attention launch per layer (dit.ARRIVAL_SUFFIX; in the e4m3 mode with e4m3 blobs on the wire: the chunk launches gate on their blobs in the kernel) and the chunked
        # carried-state launches with sp_chunks and with 2 chunks
        arrival = [] if ((engine.attn_fp8 and not getattr(engine, "fp8_wire", False)) or os.environ.get("ICV_ATTN_ARRIVAL") == "0") else ["+arrival"]
        cands = [(m + sfx, c) for m in modes for sfx, c in [(a, self.sp_chunks) for a in arrival] + [("", c) for c in sorted({self.sp_chunks, 2}, reverse=True)]]
        t0 = time.perf_counter()
        best, table = autotune_kv_exchange(engine, two_layers, sync, cands, reps=2, reduce_max=reduce_max)
        self.kv_autotune = dict(chosen=best, table=table, seconds=time.perf_counter() - t0)
        if dist.get_rank() == 0 and os.environ.get("ICV_QUIET", "0") != "1":
            print(f"[icv] K|V exchange autotune: {best[0]} x {best[1]} chunks ({self.kv_autotune['seconds']:.1f} s; "
                  + ", ".join(f"{r['kv_exchange']}/{r['sp_chunks']}: " + (f"{r['ms']:.2f} ms" if r['ms'] else 'n/a') for r in table) + ")", file=sys.stderr, flush=True)
        return best

    def _image_cond_latents(self, image, grid: TokenGrid, tiled, tile_size, tile_stride) -> torch.Tensor:
        """i2v conditioning latent y [4 + 16, T, H/8, W/8] ([EXT] Wan2.1 / diffsynth ``encode_image``): the
        VAE encoding of [image, 0, 0, ...] under a 4-channel mask that marks the first latent frame."""
        h, w, f = grid.height, grid.width, grid.num_frames
        img = _video_to_tensor([image], h, w)                                   # [3, 1, H, W] in [-1, 1]
        video = torch.cat([img, torch.zeros((3, f - 1, h, w), dtype=img.dtype)], dim=1)
        lat = self.vae.encode(video, tiled=tiled, tile_size=tile_size, tile_stride=tile_stride).to(torch.float32).cpu()
        msk = torch.zeros((1, f, h // 8, w // 8))
        msk[:, 0] = 1.0
        msk = torch.cat([msk[:, :1].repeat_interleave(4, dim=1), msk[:, 1:]], dim=1)   # [1, 4T, h8, w8]
        msk = msk.reshape(grid.T, 4, h // 8, w // 8).transpose(0, 1)                    # [4, T, h8, w8]
        return torch.cat([msk, lat], dim=0)

    # ---- D5: the generation call -----------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt: str, negative_prompt: str = "", semantic_buffer_video=None,
                 coordinate_buffer_video=None, height: int = 480, width: int = 832, num_frames: int = 81,
                 seed: Optional[int] = None, tiled: bool = True, num_inference_steps: Optional[int] = None,
                 cfg_scale: Optional[float] = None, sigma_shift: Optional[float] = None, rand_device: str = "cpu",
                 tile_size=(30, 52), tile_stride=(15, 26), progress_bar_cmd=None, return_latents: bool = False,
                 input_image=None, join_decode: bool = False, **unused):
        if self.text_encoder is None or self.vae is None:
            raise RuntimeError("WanVideoPipeline: text encoder / VAE not loaded")
        num_inference_steps = self.num_inference_steps if num_inference_steps is None else num_inference_steps
        cfg_scale = self.cfg_scale if cfg_scale is None else cfg_scale
        sigma_shift = self.sigma_shift if sigma_shift is None else sigma_shift
        grid = TokenGrid(num_frames, height, width)
        engine = self._get_engine()
        ops = engine.ops
        world, rank = 1, 0
        from . import multigpu
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and not multigpu.degraded():   # degraded: a failed multi-GPU start
                world, rank = dist.get_world_size(), dist.get_rank()                        # left a dead group behind
        except Exception:  # pragma: no cover
            pass
        lkey = (world, rank, self.parallelism, cfg_scale != 1.0)
        # process groups are created once per (world, mode).  Behind a worker pool the cache belongs to the POOL, not to this
        # pipeline object: a second generator on the same pool gives rank 0 a new pipeline while the workers keep theirs, and
        # creating groups is a collective they would never join
        layouts = multigpu.layout_cache()
        if layouts is None:
            layouts = self._layouts
        if lkey not in layouts:
            layouts[lkey] = ParallelLayout.make(world, rank, self.parallelism, use_cfg=cfg_scale != 1.0)
        layout = layouts[lkey]
        plan = layout.shard_plan(grid.S)
        kv_exchange, sp_chunks = self.kv_exchange, self.sp_chunks
        tune_key = None
        if layout.sp_world > 1:
            if kv_exchange is None:
                kv_exchange = "auto" if dist.get_backend() == "nccl" else "allgather"
            if kv_exchange == "auto":
                tune_key = (lkey, grid.S)
                kv_exchange, sp_chunks = self._kv_tuned.get(tune_key, ("allgather", sp_chunks))
        engine.prepare(grid, plan, group=layout.sp_group, sp_chunks=sp_chunks, kv_exchange=kv_exchange if layout.sp_world > 1 else None)
        self.scheduler = FlowMatchScheduler(num_inference_steps, sigma_shift, self.reference_rounding)
        # i2v (BASELINE.json config #5): CLIP tokens + conditioning latent of the first frame, once per call
        i2v = engine.cfg.has_image_input
        clip_fea = None
        if i2v:
            if input_image is None or self.image_encoder is None:
                raise ValueError("this DiT is image-to-video: pass input_image= and load a CLIP image encoder")
            clip_fea = self.image_encoder.encode_image(input_image)
        elif input_image is not None:
            raise ValueError("input_image given but the loaded DiT is text-to-video")
        # text (cond / uncond), once per prompt
        ctx_c = ctx_u = None
        if layout.branch in (None, 0):
            ctx_c = engine.encode_context(self.text_encoder.encode(prompt), clip_fea)
        if cfg_scale != 1.0 and layout.branch in (None, 1):
            ctx_u = engine.encode_context(self.text_encoder.encode(negative_prompt), clip_fea)
        # noise: CPU generator, fp32 (rand_device='cpu' upstream) -> identical across devices/ranks
        g = torch.Generator(device="cpu")
        if seed is None:
            # upstream passes generator=None: a different noise on every unseeded call.  Under a multi-rank launch every
            # rank must still draw the SAME noise (the ranks hold shards / CFG branches of one latent), so the seed is
            # drawn once on rank 0 and broadcast.
            seed = int.from_bytes(os.urandom(7), "little")
            if world > 1:
                box = [seed]
                dist.broadcast_object_list(box, src=0)
                seed = int(box[0])
        g.manual_seed(int(seed))
        latent = torch.randn((1, 16) + grid.latent_shape()[1:], generator=g, dtype=torch.float32)[0]
        if self.reference_rounding:
            latent = latent.to(torch.bfloat16).to(torch.float32)
        latent = ops.to_device(latent, torch.float32)
        # guidance buffers -> VAE latents -> tokens (step-invariant)
        buf_tokens = None
        # multi-rank run: the VAE's tiles are dealt to ALL ranks of the job (vae.TileShard; bit-identical result on every rank)
        from .vae import TileShard
        vshard = dict(shard=TileShard.current()) if (world > 1 and getattr(self.vae, "supports_tile_shard", False)) else {}
        if self.buffer_embedder is not None and semantic_buffer_video is not None and coordinate_buffer_video is not None:
            vids = (semantic_buffer_video, coordinate_buffer_video)
            for vid in vids:
                if len(vid) != num_frames:
                    raise ValueError(f"buffer video has {len(vid)} frames, num_frames={num_frames}")
            if hasattr(self.vae, "encode_many"):      # both clips in one pass over their tiles, bytes normalised on the device
                to_clip = _video_to_uint8 if getattr(self.vae, "accepts_uint8", False) else _video_to_tensor
                lats = self.vae.encode_many([to_clip(v, height, width) for v in vids], tiled=tiled, tile_size=tile_size,
                                            tile_stride=tile_stride, **vshard)
            else:
                lats = [self.vae.encode(_video_to_tensor(v, height, width), tiled=tiled, tile_size=tile_size, tile_stride=tile_stride)
                        for v in vids]
            buf_tokens = engine.embed_buffers(torch.cat([x.to(torch.float32) for x in lats], dim=0))
        if i2v:
            y = self._image_cond_latents(input_image, grid, tiled, tile_size, tile_stride)
            buf_tokens = engine.embed_cond_latents(y, add_to=buf_tokens)
        if tune_key is not None and tune_key not in self._kv_tuned:
            self._kv_tuned[tune_key] = self._autotune_kv(engine, latent, ctx_c if ctx_c is not None else ctx_u, buf_tokens, ops)
        # the hot loop (HIP)
        it = range(num_inference_steps)
        if progress_bar_cmd is not None:
            it = progress_bar_cmd(it)
        engine.denoise(latent, ctx_c, ctx_u, buf_tokens, self.scheduler, cfg_scale, steps=it,
                       branch_exchange=BranchExchange(layout) if layout.mode == "cfg+sp" else None,
                       round_bf16=self.reference_rounding)
        latent = gather_latent(latent, plan, grid, group=layout.sp_group)
        # The DECODE is sharded (a collective of every rank: vae.TileShard broadcasts the tiles) only where every rank is known to
        # take part: behind a multigpu.WorkerPool (its workers pass join_decode=True) or when the caller says so
        # (ICV_VAE_SHARD_DECODE=1: every rank calls the pipeline with return_latents=False, or with join_decode=True).  A
        # torchrun-style script that decodes on rank 0 only and asks the other ranks for latents gets the unsharded decode.
        if multigpu.layout_cache() is None and os.environ.get("ICV_VAE_SHARD_DECODE", "0") != "1":
            vshard = {}
        if return_latents:
            if join_decode and vshard.get("shard") is not None:
                # a worker rank of multigpu.WorkerPool: it needs no frames, but its share of the decode tiles is part of
                # the collective rank 0 is in (vae.WanVAE._run_tiles)
                self.vae.decode(latent, tiled=tiled, tile_size=tile_size, tile_stride=tile_stride, blend=False, **vshard)
            return latent
        video = self.vae.decode(latent, tiled=tiled, tile_size=tile_size, tile_stride=tile_stride, **vshard)
        return _tensor_to_video(video)
