"""UMT5-XXL text encoder on stock PyTorch-ROCm (outside the hot loop; north star: "the T5 text encoder
... run on stock PyTorch-ROCm").

The reference names the checkpoint ``models_t5_umt5-xxl-enc-bf16.pth``
[R infinicube/videogen/inference.py:68,78]; the module that consumes it lives in the absent diffsynth
fork.  What follows restates the PUBLIC Wan2.1 text encoder (an encoder-only T5 v1.1 / UMT5 variant:
pre-RMSNorm, un-scaled dot-product attention with a per-layer bidirectional relative-position bias,
gated tanh-GELU FFN, no biases) — [EXT], unverifiable offline: parameter names are laid out to match
the public checkpoint keys (``token_embedding.weight``, ``blocks.N.attn.{q,k,v,o}.weight``,
``blocks.N.ffn.{gate.0,fc1,fc2}.weight``, ``blocks.N.pos_embedding.embedding.weight``, ``norm.weight``)
and ``load_state_dict(strict=True)`` will say so loudly if the restatement is off.
"""

from __future__ import annotations

import glob
import html
import math
import os
import re
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


class T5LayerNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        y = x.float() * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + self.eps)
        return self.weight * y.to(self.weight.dtype)


class T5RelativeEmbedding(nn.Module):
    def __init__(self, num_buckets: int, num_heads: int, max_dist: int = 128):
        super().__init__()
        self.num_buckets, self.max_dist = num_buckets, max_dist
        self.embedding = nn.Embedding(num_buckets, num_heads)

    def forward(self, lq: int, lk: int):
        dev = self.embedding.weight.device
        rel = torch.arange(lk, device=dev)[None, :] - torch.arange(lq, device=dev)[:, None]
        nb = self.num_buckets // 2                      # bidirectional
        buckets = (rel > 0).long() * nb
        rel = rel.abs()
        max_exact = nb // 2
        large = max_exact + (torch.log(rel.float().clamp(min=1) / max_exact)
                             / math.log(self.max_dist / max_exact) * (nb - max_exact)).long()
        large = torch.min(large, torch.full_like(large, nb - 1))
        buckets = buckets + torch.where(rel < max_exact, rel, large)
        return self.embedding(buckets).permute(2, 0, 1)[None]      # [1, heads, lq, lk]


class T5Attention(nn.Module):
    def __init__(self, dim: int, dim_attn: int, num_heads: int):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim_attn // num_heads
        self.q = nn.Linear(dim, dim_attn, bias=False)
        self.k = nn.Linear(dim, dim_attn, bias=False)
        self.v = nn.Linear(dim, dim_attn, bias=False)
        self.o = nn.Linear(dim_attn, dim, bias=False)

    def forward(self, x, mask, pos_bias):
        b, n, c = x.shape[0], self.num_heads, self.head_dim
        q = self.q(x).view(b, -1, n, c).transpose(1, 2)
        k = self.k(x).view(b, -1, n, c).transpose(1, 2)
        v = self.v(x).view(b, -1, n, c).transpose(1, 2)
        bias = pos_bias.to(q.dtype).expand(b, -1, -1, -1).clone()
        if mask is not None:
            bias = bias.masked_fill(mask[:, None, None, :] == 0, torch.finfo(q.dtype).min)
        # T5 does NOT scale by 1/sqrt(d): scale=1.0
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=bias, scale=1.0)
        return self.o(y.transpose(1, 2).reshape(b, -1, n * c))


class T5FeedForward(nn.Module):
    def __init__(self, dim: int, dim_ffn: int):
        super().__init__()
        self.gate = nn.Sequential(nn.Linear(dim, dim_ffn, bias=False), nn.GELU(approximate="tanh"))
        self.fc1 = nn.Linear(dim, dim_ffn, bias=False)
        self.fc2 = nn.Linear(dim_ffn, dim, bias=False)

    def forward(self, x):
        return self.fc2(self.fc1(x) * self.gate(x))


class T5SelfAttention(nn.Module):
    def __init__(self, dim, dim_attn, dim_ffn, num_heads, num_buckets):
        super().__init__()
        self.norm1 = T5LayerNorm(dim)
        self.attn = T5Attention(dim, dim_attn, num_heads)
        self.norm2 = T5LayerNorm(dim)
        self.ffn = T5FeedForward(dim, dim_ffn)
        self.pos_embedding = T5RelativeEmbedding(num_buckets, num_heads)   # per layer (shared_pos=False)

    def forward(self, x, mask):
        e = self.pos_embedding(x.size(1), x.size(1))
        x = x + self.attn(self.norm1(x), mask, e)
        return x + self.ffn(self.norm2(x))


class UMT5Encoder(nn.Module):
    """umt5-xxl encoder: vocab 256384, dim 4096, ffn 10240, 64 heads, 24 layers, 32 buckets."""

    def __init__(self, vocab_size=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64,
                 num_layers=24, num_buckets=32):
        super().__init__()
        self.dim = dim
        self.token_embedding = nn.Embedding(vocab_size, dim)
        self.blocks = nn.ModuleList([T5SelfAttention(dim, dim_attn, dim_ffn, num_heads, num_buckets)
                                     for _ in range(num_layers)])
        self.norm = T5LayerNorm(dim)

    @torch.no_grad()
    def forward(self, ids, mask=None):
        x = self.token_embedding(ids)
        for blk in self.blocks:
            x = blk(x, mask)
        return self.norm(x)


def _clean(text: str) -> str:
    """'whitespace' cleaning of the Wan tokenizer wrapper (ftfy is not required for ASCII/CJK prompts)."""
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


class UMT5TextEncoder:
    """prompt -> f32 [text_len, dim] with rows past the token count zeroed (Wan prompter semantics)."""

    def __init__(self, model: UMT5Encoder, tokenizer, device, text_len: int = 512):
        self.model, self.tokenizer, self.device, self.text_len = model, tokenizer, device, text_len

    @torch.no_grad()
    def encode(self, prompt: str) -> torch.Tensor:
        tok = self.tokenizer([_clean(prompt)], return_tensors="pt", padding="max_length", truncation=True,
                             max_length=self.text_len, add_special_tokens=True)
        ids, mask = tok["input_ids"].to(self.device), tok["attention_mask"].to(self.device)
        emb = self.model(ids, mask)[0].float()
        n = int(mask[0].gt(0).sum())
        emb[n:] = 0
        return emb.cpu()


def load_umt5_encoder(pattern, device, torch_dtype=torch.bfloat16, tokenizer_config=None) -> UMT5TextEncoder:
    files = sorted(glob.glob(pattern))
    if not files:
        raise FileNotFoundError(f"UMT5 encoder checkpoint not found: {pattern!r} (skip_download=True: nothing is fetched)")
    sd = torch.load(files[0], map_location="cpu", weights_only=True)
    # architecture read off the tensors (umt5-xxl for the real file); a naming mismatch still fails in the strict load
    vocab, dim = (int(x) for x in sd["token_embedding.weight"].shape)
    buckets, heads = (int(x) for x in sd["blocks.0.pos_embedding.embedding.weight"].shape)
    layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    model = UMT5Encoder(vocab_size=vocab, dim=dim, dim_attn=int(sd["blocks.0.attn.q.weight"].shape[0]),
                        dim_ffn=int(sd["blocks.0.ffn.fc1.weight"].shape[0]), num_heads=heads, num_layers=layers, num_buckets=buckets)
    model.load_state_dict(sd, strict=True)
    model = model.to(device=device, dtype=torch_dtype).eval()
    tok_dir = tokenizer_config.resolve() if tokenizer_config is not None else os.path.join(
        os.path.dirname(files[0]), "google", "umt5-xxl")
    if not os.path.isdir(tok_dir):
        raise FileNotFoundError(f"UMT5 tokenizer directory not found: {tok_dir!r}")
    from transformers import AutoTokenizer
    return UMT5TextEncoder(model, AutoTokenizer.from_pretrained(tok_dir), device)
