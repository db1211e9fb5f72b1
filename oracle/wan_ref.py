"""ORACLE — CPU restatement of the buffer-conditioned Wan2.1 DiT denoising loop.

THIS IS TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  Nothing under ``infinicube_amd/`` imports it, and the product
path has no CPU fallback.

** PARITY UNPINNED ** — the reference's arithmetic for this path lives in the third-party package
``diffsynth @ git+https://github.com/yifanlu0227/DiffSynth-Studio-InfiniCube`` (no tag, no commit)
[R pyproject.toml:71], which is absent from /root/reference and from this container, and the
reference's only harness asserts nothing numeric [R infinicube/videogen/test_api.py:88-90].  What
is restated here is the *published* Wan2.1 / DiffSynth ``wan_video_dit`` algorithm
(every line tagged [EXT] there), anchored on the reference's own call sites:

  * the nine kwargs of ``pipe(...)``                 [R infinicube/videogen/inference.py:216-226]
  * ``initialize_buffer_embedder(16, zero_init)``    [R infinicube/videogen/inference.py:86-88]
  * "embeds guidance buffers to tokens and adds to noisy tokens"  [R README.md:65]
  * model list (DiT / UMT5 / Wan-VAE)                [R infinicube/videogen/inference.py:67-79]
  * 480p = 480x832, 93-frame cap                     [R infinicube/inference/guidance_buffer_generation.py:79-82,744-745]

Everything is plain PyTorch on CPU in ``dtype`` (fp32 by default, fp64 on request); RoPE angles
and the sinusoidal embedding are computed in fp64 as upstream does.
"""

from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# A.2 embeddings
# --------------------------------------------------------------------------------------------
def sinusoidal_embedding_1d(dim: int, position: Tensor) -> Tensor:
    """cat[cos(t f_i), sin(t f_i)], f_i = 10000^(-i/(dim/2)), fp64 (Appendix A.2)."""
    half = dim // 2
    freqs = torch.pow(10000.0, -torch.arange(half, dtype=torch.float64) / half).to(position.device)
    ang = torch.outer(position.to(torch.float64).reshape(-1), freqs)
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=1)


def time_embed(sd: Dict[str, Tensor], cfg, timestep: float, dtype=torch.float32) -> Tuple[Tensor, Tensor]:
    """returns (t [d], t_mod [6, d])."""
    dev = sd["time_embedding.0.weight"].device   # the oracle runs wherever its weights live (CPU by default)
    emb = sinusoidal_embedding_1d(cfg.freq_dim, torch.tensor([timestep], dtype=torch.float64)).to(device=dev, dtype=dtype)
    t = F.linear(F.silu(F.linear(emb, sd["time_embedding.0.weight"].to(dtype), sd["time_embedding.0.bias"].to(dtype))),
                 sd["time_embedding.2.weight"].to(dtype), sd["time_embedding.2.bias"].to(dtype))
    t_mod = F.linear(F.silu(t), sd["time_projection.1.weight"].to(dtype), sd["time_projection.1.bias"].to(dtype))
    return t[0], t_mod[0].reshape(6, cfg.dim)


def text_embed(sd: Dict[str, Tensor], context: Tensor, dtype=torch.float32) -> Tensor:
    h = F.linear(context.to(dtype), sd["text_embedding.0.weight"].to(dtype), sd["text_embedding.0.bias"].to(dtype))
    h = F.gelu(h, approximate="tanh")
    return F.linear(h, sd["text_embedding.2.weight"].to(dtype), sd["text_embedding.2.bias"].to(dtype))


def image_embed(sd: Dict[str, Tensor], clip_fea: Tensor, dtype=torch.float32) -> Tensor:
    """i2v ([EXT] Wan2.1 MLPProj): LayerNorm -> Linear -> GELU (erf) -> Linear -> LayerNorm on the CLIP
    tokens [img_len, img_dim] -> [img_len, d]; LayerNorm eps is torch's default 1e-5."""
    g = lambda n: sd[f"img_emb.proj.{n}"].to(dtype)   # noqa: E731
    h = F.layer_norm(clip_fea.to(dtype), (clip_fea.shape[-1],), g("0.weight"), g("0.bias"), 1e-5)
    h = F.gelu(F.linear(h, g("1.weight"), g("1.bias")))
    h = F.linear(h, g("3.weight"), g("3.bias"))
    return F.layer_norm(h, (h.shape[-1],), g("4.weight"), g("4.bias"), 1e-5)


# --------------------------------------------------------------------------------------------
# A.3 RoPE-3D
# --------------------------------------------------------------------------------------------
def rope_axis_dims(head_dim: int) -> Tuple[int, int, int]:
    """(44, 42, 42) real dims for head_dim 128: f gets the remainder."""
    hw = head_dim // 3
    return head_dim - 2 * hw, hw, hw


def rope_axis_angles(axis_dim: int, n_pos: int, theta: float = 10000.0) -> Tensor:
    """[n_pos, axis_dim/2] fp64 angles pos * theta^(-2j/axis_dim)."""
    inv = 1.0 / (theta ** (torch.arange(0, axis_dim, 2, dtype=torch.float64)[: axis_dim // 2] / axis_dim))
    return torch.outer(torch.arange(n_pos, dtype=torch.float64), inv)


def rope_freqs_3d(head_dim: int, T: int, Hp: int, Wp: int) -> Tensor:
    """complex128 [T*Hp*Wp, head_dim/2]: first 22 pairs <- frame, next 21 <- row, last 21 <- col."""
    df, dh, dw = rope_axis_dims(head_dim)
    af, ah, aw = rope_axis_angles(df, T), rope_axis_angles(dh, Hp), rope_axis_angles(dw, Wp)
    ang = torch.cat([
        af[:, None, None, :].expand(T, Hp, Wp, -1),
        ah[None, :, None, :].expand(T, Hp, Wp, -1),
        aw[None, None, :, :].expand(T, Hp, Wp, -1)], dim=-1).reshape(T * Hp * Wp, -1)
    return torch.polar(torch.ones_like(ang), ang)


def rope_apply(x: Tensor, freqs: Tensor, num_heads: int) -> Tensor:
    """x [S, heads*hd]; rotate adjacent pairs (2i, 2i+1) in complex128, cast back."""
    S = x.shape[0]
    xc = torch.view_as_complex(x.to(torch.float64).reshape(S, num_heads, -1, 2))
    out = torch.view_as_real(xc * freqs[:, None, :]).reshape(S, -1)
    return out.to(x.dtype)


# --------------------------------------------------------------------------------------------
# A.4 block
# --------------------------------------------------------------------------------------------
def rms_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    return x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps) * w


def layer_norm(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def modulate(x: Tensor, shift: Tensor, scale: Tensor) -> Tensor:
    return x * (1.0 + scale) + shift


def attention(q: Tensor, k: Tensor, v: Tensor, num_heads: int, scale: Optional[float] = None) -> Tensor:
    """q [Sq, H*hd], k/v [Sk, H*hd] -> [Sq, H*hd]; non-causal softmax(q k^T * scale) v, scale = 1/sqrt(hd) by default.
    On CPU: torch's SDPA.  On a GPU (the full-size checker of tests/test_fullsize_gpu.py): the explicit
    definition matmul -> softmax -> matmul over query chunks, so the checker depends on no fused attention
    backend (and on nothing in libicvideo)."""
    Sq, Sk = q.shape[0], k.shape[0]
    qh = q.reshape(Sq, num_heads, -1).transpose(0, 1)
    kh = k.reshape(Sk, num_heads, -1).transpose(0, 1)
    vh = v.reshape(Sk, num_heads, -1).transpose(0, 1)
    if q.device.type == "cpu":
        o = F.scaled_dot_product_attention(qh[None], kh[None], vh[None], scale=scale)[0]
        return o.transpose(0, 1).reshape(Sq, -1)
    sc = (1.0 / math.sqrt(qh.shape[-1])) if scale is None else scale
    return attention_explicit(qh, kh, vh, sc).transpose(0, 1).reshape(Sq, -1)


def attention_explicit(qh: Tensor, kh: Tensor, vh: Tensor, sc: float) -> Tensor:
    """[H, Sq, hd] x [H, Sk, hd] x [H, Sk, hd] -> [H, Sq, hd]: softmax(sc * q k^T) v over query chunks of <= 2 GiB of fp32 scores.
    The scale rides in the GEMM's alpha (baddbmm, beta = 0): the same fp32 product-then-scale as a separate multiply, one pass
    over the scores fewer - at S = 37 440 the checker is bound by those passes, not by the matmuls.  Pinned against torch's SDPA
    on CPU by tests/test_oracle.py."""
    H, Sq, Sk = qh.shape[0], qh.shape[1], kh.shape[1]
    step = max(1, (1 << 31) // (4 * H * Sk))
    kt = kh.transpose(1, 2)
    outs = []
    for r0 in range(0, Sq, step):
        qc = qh[:, r0: r0 + step]
        s = torch.empty((H, qc.shape[1], Sk), dtype=qh.dtype, device=qh.device)
        torch.baddbmm(s, qc, kt, beta=0.0, alpha=sc, out=s)
        outs.append(torch.matmul(torch.softmax(s, dim=-1), vh))
        del s
    return torch.cat(outs, dim=1)


def attention_fp8(q: Tensor, k: Tensor, v: Tensor, num_heads: int) -> Tensor:
    """The build's fp8 attention mode (not the reference's): per head, Q / K / V are scaled by a power of two
    2^-ceil(log2(amax/448)) and rounded to e4m3; S = Qq Kq^T in log2 units (K already carries scale * log2 e), P = 2^(S - max)
    rounded to e4m3, O = P Vq / sum(P) with the row sum taken before P is rounded.  The kernel rounds P against a LAZY
    reference (P up to 2^8 larger than here), so the two agree statistically, not bit for bit: e4m3 keeps the same
    relative precision at either scale."""
    Sq, Sk = q.shape[0], k.shape[0]
    hd = q.shape[1] // num_heads
    out = torch.empty((Sq, q.shape[1]), dtype=torch.float32)

    def qd(x):   # [S, hd] -> dequantised e4m3 values
        amax = float(x.abs().max())
        e = math.ceil(math.log2(amax / FP8_MAX)) if amax > 0 else 0
        return (x * 2.0 ** -e).to(torch.float8_e4m3fn).to(torch.float32) * 2.0 ** e

    for h in range(num_heads):
        sl = slice(h * hd, (h + 1) * hd)
        qh, kh, vh = qd(q[:, sl].float()), qd(k[:, sl].float()), qd(v[:, sl].float())
        s = qh @ kh.t()
        p = torch.exp2(s - s.max(dim=-1, keepdim=True).values)
        l = p.sum(-1, keepdim=True)
        out[:, sl] = (p.to(torch.float8_e4m3fn).to(torch.float32) @ vh) / l
    return out


def _lin(sd, name, x, dtype):
    return F.linear(x, sd[f"{name}.weight"].to(dtype), sd[f"{name}.bias"].to(dtype))


# ---- fp8 GEMM mode of the build (BASELINE.json config #5 "fp8 MFMA weights"; not a behaviour of the reference) ----
FP8_MAX = 448.0   # largest finite OCP e4m3 value
FP8_ALL = ("wqkv", "wo", "xq_w", "xo_w", "f0_w", "f2_w")
FP8_DEFAULT = ("wqkv",)   # the build's default e4m3 set (dit.WanDiT.FP8_DEFAULT: the QKV projection; the other five stay bf16)


def quantize_rows_fp8(x: Tensor) -> Tuple[Tensor, Tensor]:
    """x [rows, K] -> (q: the e4m3 values as f32, scale f32 [rows]); scale = row abs-max / 448 (1 for a zero
    row), q = RNE_e4m3(x * (1 / scale)).  The exact f32 operation order of csrc/fp8.hip."""
    x = x.to(torch.float32)
    amax = x.abs().amax(dim=-1)
    scale = torch.where(amax > 0, amax / FP8_MAX, torch.ones_like(amax))
    inv = 1.0 / scale
    q = (x * inv[:, None]).to(torch.float8_e4m3fn).to(torch.float32)
    return q, scale


def fake_quant_rows(x: Tensor) -> Tensor:
    q, s = quantize_rows_fp8(x)
    return q * s[:, None]


def _lin8(sd, name, x, dtype):
    """Linear with both operands row-quantised to e4m3 (per token / per output channel), f32 accumulate."""
    w = fake_quant_rows(sd[f"{name}.weight"]).to(dtype)
    return F.linear(fake_quant_rows(x).to(dtype), w, sd[f"{name}.bias"].to(dtype))


def dit_block(sd: Dict[str, Tensor], cfg, i: int, x: Tensor, ctx: Tensor, t_mod: Tensor,
              freqs: Tensor, dtype=torch.float32, kv_override=None, ctx_img: Optional[Tensor] = None,
              fp8: bool = False) -> Tensor:
    """``fp8``: which of the six projections applied to token rows take e4m3 row-quantised operands — True = all six,
    or a tuple of the build's names ("wqkv" = q/k/v, "wo", "xq_w" = cross q, "xo_w" = cross o, "f0_w" = ffn.0,
    "f2_w" = ffn.2); the cross-attention K/V projections of the context always stay unquantised."""
    p = f"blocks.{i}"
    fp8_set = FP8_ALL if fp8 is True else (tuple(fp8) if fp8 else ())

    def lin(sd_, name, x_, dtype_):
        key = {"self_attn.q": "wqkv", "self_attn.k": "wqkv", "self_attn.v": "wqkv", "self_attn.o": "wo",
               "cross_attn.q": "xq_w", "cross_attn.o": "xo_w", "ffn.0": "f0_w", "ffn.2": "f2_w"}[name.split(".", 2)[2]]
        return (_lin8 if key in fp8_set else _lin)(sd_, name, x_, dtype_)
    mod = sd[f"{p}.modulation"].to(dtype).reshape(6, cfg.dim) + t_mod
    sh1, sc1, g1, sh2, sc2, g2 = mod.unbind(0)
    H, eps = cfg.num_heads, cfg.eps
    # self-attention
    h = modulate(layer_norm(x, None, None, eps), sh1, sc1)
    q = rope_apply(rms_norm(lin(sd, f"{p}.self_attn.q", h, dtype), sd[f"{p}.self_attn.norm_q.weight"].to(dtype), eps), freqs, H)
    k = rope_apply(rms_norm(lin(sd, f"{p}.self_attn.k", h, dtype), sd[f"{p}.self_attn.norm_k.weight"].to(dtype), eps), freqs, H)
    v = lin(sd, f"{p}.self_attn.v", h, dtype)
    if kv_override is not None:  # sequence-parallel tests: attend over gathered K/V
        k, v = kv_override(k, v)
    x = x + g1 * lin(sd, f"{p}.self_attn.o", attention(q, k, v, H), dtype)
    # cross-attention to text (no gate)
    h = layer_norm(x, sd[f"{p}.norm3.weight"].to(dtype), sd[f"{p}.norm3.bias"].to(dtype), eps)
    q = rms_norm(lin(sd, f"{p}.cross_attn.q", h, dtype), sd[f"{p}.cross_attn.norm_q.weight"].to(dtype), eps)
    k = rms_norm(_lin(sd, f"{p}.cross_attn.k", ctx, dtype), sd[f"{p}.cross_attn.norm_k.weight"].to(dtype), eps)
    v = _lin(sd, f"{p}.cross_attn.v", ctx, dtype)
    a = attention(q, k, v, H)
    if ctx_img is not None:   # i2v: a second softmax over the CLIP tokens, outputs summed before o
        k_img = rms_norm(_lin(sd, f"{p}.cross_attn.k_img", ctx_img, dtype), sd[f"{p}.cross_attn.norm_k_img.weight"].to(dtype), eps)
        a = a + attention(q, k_img, _lin(sd, f"{p}.cross_attn.v_img", ctx_img, dtype), H)
    x = x + lin(sd, f"{p}.cross_attn.o", a, dtype)
    # FFN
    h = modulate(layer_norm(x, None, None, eps), sh2, sc2)
    h = F.gelu(lin(sd, f"{p}.ffn.0", h, dtype), approximate="tanh")
    return x + g2 * lin(sd, f"{p}.ffn.2", h, dtype)


# --------------------------------------------------------------------------------------------
# patchify / unpatchify / buffer embedder / head
# --------------------------------------------------------------------------------------------
def patchify_tokens(latent: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """Conv3d(C->d, k=s=(1,2,2)) then 'c f h w -> (f h w) c'."""
    y = F.conv3d(latent[None], w, b, stride=tuple(w.shape[2:]))[0]
    return y.reshape(y.shape[0], -1).transpose(0, 1).contiguous()


def buffer_embed(bsd: Dict[str, Tensor], buffer_latents: Tensor, dtype=torch.float32) -> Tensor:
    """Guidance-buffer tokens [S, d] (step-invariant).  H1 'concat' or H2 'dual' (SURVEY §8a K1)."""
    bl = buffer_latents.to(dtype)
    if "proj.weight" in bsd:
        return patchify_tokens(bl, bsd["proj.weight"].to(dtype), bsd["proj.bias"].to(dtype))
    c = bl.shape[0] // 2
    return (patchify_tokens(bl[:c], bsd["semantic_proj.weight"].to(dtype), bsd["semantic_proj.bias"].to(dtype))
            + patchify_tokens(bl[c:], bsd["coordinate_proj.weight"].to(dtype), bsd["coordinate_proj.bias"].to(dtype)))


def unpatchify(x: Tensor, grid: Tuple[int, int, int], out_dim: int, patch=(1, 2, 2)) -> Tensor:
    """'(f h w) (x y z c) -> c (f x) (h y) (w z)'."""
    f, h, w = grid
    px, py, pz = patch
    y = x.reshape(f, h, w, px, py, pz, out_dim).permute(6, 0, 3, 1, 4, 2, 5)
    return y.reshape(out_dim, f * px, h * py, w * pz)


def head(sd: Dict[str, Tensor], cfg, x: Tensor, t: Tensor, dtype=torch.float32) -> Tensor:
    mod = sd["head.modulation"].to(dtype).reshape(2, cfg.dim) + t[None, :]
    shift, scale = mod.unbind(0)
    return _lin(sd, "head.head", modulate(layer_norm(x, None, None, cfg.eps), shift, scale), dtype)


def dit_forward(sd: Dict[str, Tensor], cfg, latent: Tensor, context: Tensor, timestep: float,
                buf_tokens: Optional[Tensor] = None, dtype=torch.float32,
                num_layers: Optional[int] = None, return_tokens: bool = False,
                clip_fea: Optional[Tensor] = None, y: Optional[Tensor] = None, fp8: bool = False) -> Tensor:
    """One DiT forward: latent [C,T,H8,W8], raw text context [text_len, text_dim] -> velocity
    [out_dim, T, H8, W8].  i2v: ``y`` [in_dim-C,T,H8,W8] is concatenated under the noise channels and
    ``clip_fea`` [img_len, img_dim] feeds the image cross-attention."""
    if y is not None:
        latent = torch.cat([latent.to(dtype), y.to(dtype)], dim=0)
    ctx_img = image_embed(sd, clip_fea, dtype) if clip_fea is not None else None
    C, T, H8, W8 = latent.shape
    grid = (T // cfg.patch[0], H8 // cfg.patch[1], W8 // cfg.patch[2])
    t, t_mod = time_embed(sd, cfg, timestep, dtype)
    ctx = text_embed(sd, context, dtype)
    x = patchify_tokens(latent.to(dtype), sd["patch_embedding.weight"].to(dtype), sd["patch_embedding.bias"].to(dtype))
    if buf_tokens is not None:
        x = x + buf_tokens.to(dtype)
    freqs = rope_freqs_3d(cfg.head_dim, *grid).to(x.device)
    for i in range(cfg.num_layers if num_layers is None else num_layers):
        x = dit_block(sd, cfg, i, x, ctx, t_mod, freqs, dtype, ctx_img=ctx_img, fp8=fp8)
    if return_tokens:
        return x
    return unpatchify(head(sd, cfg, x, t, dtype), grid, cfg.out_dim, cfg.patch)


# --------------------------------------------------------------------------------------------
# A.5 / A.6 sampler + CFG loop
# --------------------------------------------------------------------------------------------
def flow_match_sigmas(num_steps: int, shift: float = 5.0) -> Tensor:
    s = torch.linspace(1.0, 0.0, num_steps + 1, dtype=torch.float64)[:-1]
    return shift * s / (1.0 + (shift - 1.0) * s)


def denoise_loop(sd, bsd, cfg, noise: Tensor, ctx_cond: Tensor, ctx_uncond: Tensor,
                 buffer_latents: Optional[Tensor], num_steps: int = 50, cfg_scale: float = 5.0,
                 shift: float = 5.0, dtype=torch.float32, trace: Optional[list] = None,
                 clip_fea: Optional[Tensor] = None, y: Optional[Tensor] = None, fp8: bool = False,
                 reference_rounding: bool = False) -> Tensor:
    """for sigma in sigmas: v = v_u + s (v_c - v_u); x += v (sigma_next - sigma).
    ``reference_rounding`` ([EXT], ORACLE_RISKS.md R1-R3): the rounding points of a pipeline that keeps timestep,
    noise, noise_pred and latents in torch_dtype = bf16: timestep -> bf16 before the sinusoidal embedding, noise ->
    bf16, every tensor op of the CFG combine and the Euler update rounded to bf16."""
    sig = flow_match_sigmas(num_steps, shift)
    buf = buffer_embed(bsd, buffer_latents, dtype) if buffer_latents is not None else None
    rb = (lambda t: t.to(torch.bfloat16).to(dtype)) if reference_rounding else (lambda t: t)   # noqa: E731
    x = rb(noise.to(dtype).clone())
    for i in range(num_steps):
        ts = float(sig[i]) * 1000.0
        if reference_rounding:
            ts = float(torch.tensor(ts, dtype=torch.float32).to(torch.bfloat16))
        v_c = rb(dit_forward(sd, cfg, x, ctx_cond, ts, buf, dtype, clip_fea=clip_fea, y=y, fp8=fp8))
        if cfg_scale != 1.0:
            v_u = rb(dit_forward(sd, cfg, x, ctx_uncond, ts, buf, dtype, clip_fea=clip_fea, y=y, fp8=fp8))
            v = rb(v_u + rb(cfg_scale * rb(v_c - v_u)))
        else:
            v = v_c
        nxt = float(sig[i + 1]) if i + 1 < num_steps else 0.0
        x = rb(x + rb(v * (nxt - float(sig[i]))))
        if trace is not None:
            trace.append(x.clone())
    return x


def round_state_dict_to_bf16(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Matrices are stored in bf16 by the product; give the oracle the same rounded values so a
    parity gap measures kernel arithmetic, not weight quantisation (SURVEY.md §8d tolerance)."""
    return {k: v.to(torch.bfloat16).to(torch.float32) for k, v in sd.items()}


def psnr(a: Tensor, b: Tensor) -> float:
    """LATENT PSNR: peak = max |reference latent| (a latent has no fixed dynamic range; this is the convention of every
    "latent PSNR" figure in this repo).  The north star's ">= 40 dB" is stated on FRAMES: tests/psnr_util.frame_psnr (uint8,
    peak 255, same decoder on both arms) is reported next to it everywhere; snr_db below is the range-free companion."""
    a, b = a.double(), b.double()
    mse = float(((a - b) ** 2).mean())
    peak = float(b.abs().max())
    return float("inf") if mse == 0 else 10.0 * math.log10(peak * peak / mse)


def snr_db(a: Tensor, b: Tensor) -> float:
    """Signal-to-noise ratio 10 log10(mean ref^2 / mean err^2): no peak convention involved."""
    a, b = a.double(), b.double()
    mse = float(((a - b) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * math.log10(float((b ** 2).mean()) / mse)
