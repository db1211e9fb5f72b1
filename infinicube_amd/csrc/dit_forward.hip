// Context-style entry points of the C ABI (SURVEY.md §8b "B-native": icv_create / icv_bind_weight / icv_dit_forward):
// ONE call enqueues the whole launch sequence of a DiT forward on a token shard — patch embed, L blocks, head — from C++,
// for a host that does not want to drive the per-op entry points itself.  It computes nothing of its own: every step is
// one of the per-op launchers of this library, called in exactly the order infinicube_amd/videogen/dit.py issues them, so
// the two drivers are bit-identical by construction (tests/test_dit_gpu.py::test_native_forward_matches_python_driver).
// Modes (all of dit.py's): bf16 and the e4m3 projection / e4m3 self-attention modes (icv_dit_set_fp8), t2v and i2v,
// single rank and the sequence-parallel schedule (icv_dit_set_seqpar: K|V rows of every layer travel by icv_allgather_kv
// on a side stream fenced with events, attention consumes the row chunks in arrival order with carried softmax state).
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "icv_common.h"

namespace {

// one projection weight: bf16 [N, K] (scale == nullptr) or e4m3 [N, K] + f32 scale per output row
struct Wt {
  const void* w = nullptr;
  const float* s = nullptr;
};

struct Layer {
  Wt wqkv, wo, xq_w, xo_w, f0_w, f2_w;
  const float *bqkv = nullptr, *nq = nullptr, *nk = nullptr, *bo = nullptr, *n3w = nullptr, *n3b = nullptr, *xq_b = nullptr, *xnq = nullptr,
              *xo_b = nullptr, *f0_b = nullptr, *f2_b = nullptr;
};

// GEMM A operand: bf16 rows (s == nullptr) or e4m3 rows + one f32 scale per row
struct Act {
  const void* p = nullptr;
  const float* s = nullptr;
};

}  // namespace

struct icv_dit {
  icv_dit_config cfg;
  std::vector<Layer> layers;
  // globals: weights
  const void *patch_w = nullptr, *head_w = nullptr;
  const float *patch_b = nullptr, *head_b = nullptr, *rope = nullptr;
  // workspace (borrowed, like every tensor of this ABI)
  float *x = nullptr, *x_stem = nullptr;
  void *h = nullptr, *qkv = nullptr, *att = nullptr, *ff = nullptr, *patches = nullptr;
  // fp8 modes: e4m3 activations + row scales, e4m3 attention workspace
  void *h8 = nullptr, *att8 = nullptr, *ff8 = nullptr;
  float *h8s = nullptr, *att8s = nullptr, *ff8s = nullptr;
  void *a8_qq = nullptr, *a8_kq = nullptr, *a8_vt = nullptr;
  float* a8_amax = nullptr;
  bool attn_fp8 = false;
  // sequence-parallel schedule
  icv_comm* comm = nullptr;
  int64_t world = 1;
  std::vector<int64_t> bounds;       // row bounds of the K|V exchange chunks inside the local shard
  hipStream_t side = nullptr;
  hipEvent_t ev_ready = nullptr;
  std::vector<hipEvent_t> ev_done;
  void *kv_loc = nullptr, *kv_full = nullptr;
  float *sp_acc = nullptr, *sp_ml = nullptr;
  // optional per-launch timing of the self-attention kernel (icv_dit_profile): event pairs recorded on the launch stream
  bool profile = false;
  std::vector<hipEvent_t> events;   // 2 per timed launch; reused by the next profiled run after icv_dit_profile_read
  size_t n_events = 0;
  ~icv_dit() {
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_done) (void)hipEventDestroy(e);
    if (ev_ready) (void)hipEventDestroy(ev_ready);
  }
  hipEvent_t next_event() {
    if (n_events == events.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      events.push_back(e);
    }
    return events[n_events++];
  }
};

extern "C" int icv_dit_create(const icv_dit_config* cfg, icv_dit** out) {
  ICV_REQUIRE(cfg && out, "icv_dit_create: null argument");
  ICV_REQUIRE(cfg->dim > 0 && cfg->dim % 128 == 0 && cfg->heads * 128 == cfg->dim, "icv_dit_create: dim must be heads * 128");
  ICV_REQUIRE(cfg->ffn_dim > 0 && cfg->ffn_dim % 64 == 0 && cfg->layers >= 0 && cfg->n_tok > 0, "icv_dit_create: bad sizes");
  ICV_REQUIRE(cfg->T > 0 && cfg->Hp > 0 && cfg->Wp > 0 && cfg->tok0 >= 0 && cfg->tok0 + cfg->n_tok <= (int64_t)cfg->T * cfg->Hp * cfg->Wp,
              "icv_dit_create: token shard outside the (T, Hp, Wp) grid");
  ICV_REQUIRE(cfg->k_patch > 0 && cfg->k_patch % 64 == 0 && cfg->out_cols > 0 && cfg->out_cols % 4 == 0, "icv_dit_create: k_patch %% 64, out_cols %% 4");
  icv_dit* d = new icv_dit();
  d->cfg = *cfg;
  d->layers.resize((size_t)cfg->layers);
  *out = d;
  return 0;
}

extern "C" void icv_dit_destroy(icv_dit* d) { delete d; }

extern "C" int icv_dit_bind(icv_dit* d, const char* name, int64_t layer, const void* ptr) {
  ICV_REQUIRE(d && name && ptr, "icv_dit_bind: null argument");
  const std::string n(name);
  if (layer >= 0) {
    ICV_REQUIRE(layer < (int64_t)d->layers.size(), "icv_dit_bind: layer %lld out of range", (long long)layer);
    Layer& L = d->layers[(size_t)layer];
#define BINDW(F) if (n == #F) { L.F.w = ptr; return 0; } if (n == #F "_s") { L.F.s = (const float*)ptr; return 0; }
#define BINDF(F) if (n == #F) { L.F = (const float*)ptr; return 0; }
    BINDW(wqkv) BINDW(wo) BINDW(xq_w) BINDW(xo_w) BINDW(f0_w) BINDW(f2_w)
    BINDF(bqkv) BINDF(nq) BINDF(nk) BINDF(bo) BINDF(n3w) BINDF(n3b) BINDF(xq_b) BINDF(xnq) BINDF(xo_b) BINDF(f0_b) BINDF(f2_b)
#undef BINDW
#undef BINDF
    icv_set_error("icv_dit_bind: unknown per-layer tensor '%s'", name);
    return 1;
  }
#define BINDG(F, T) if (n == #F) { d->F = (T)ptr; return 0; }
  BINDG(patch_w, const void*) BINDG(head_w, const void*) BINDG(patch_b, const float*) BINDG(head_b, const float*) BINDG(rope, const float*)
  BINDG(x, float*) BINDG(x_stem, float*) BINDG(h, void*) BINDG(qkv, void*) BINDG(att, void*) BINDG(ff, void*) BINDG(patches, void*)
  BINDG(h8, void*) BINDG(att8, void*) BINDG(ff8, void*) BINDG(h8s, float*) BINDG(att8s, float*) BINDG(ff8s, float*)
  BINDG(a8_qq, void*) BINDG(a8_kq, void*) BINDG(a8_vt, void*) BINDG(a8_amax, float*)
  BINDG(kv_loc, void*) BINDG(kv_full, void*) BINDG(sp_acc, float*) BINDG(sp_ml, float*)
#undef BINDG
  icv_set_error("icv_dit_bind: unknown tensor '%s'", name);
  return 1;
}

extern "C" int icv_dit_set_fp8(icv_dit* d, int attn_fp8) {
  ICV_REQUIRE(d, "icv_dit_set_fp8: null context");
  d->attn_fp8 = attn_fp8 != 0;
  return 0;
}

extern "C" int icv_dit_set_seqpar(icv_dit* d, icv_comm* comm, int64_t world, int64_t n_chunks, const int64_t* bounds, void* side_stream) {
  ICV_REQUIRE(d, "icv_dit_set_seqpar: null context");
  if (!comm) {                       // back to the single-rank schedule
    d->comm = nullptr; d->world = 1; d->bounds.clear(); d->side = nullptr;
    return 0;
  }
  ICV_REQUIRE(world >= 1 && n_chunks >= 1 && bounds && side_stream, "icv_dit_set_seqpar: world >= 1, n_chunks >= 1, bounds and a side stream are required");
  ICV_REQUIRE(bounds[0] == 0 && bounds[n_chunks] == d->cfg.n_tok, "icv_dit_set_seqpar: bounds must run from 0 to n_tok");
  for (int64_t c = 0; c < n_chunks; ++c) ICV_REQUIRE(bounds[c + 1] > bounds[c], "icv_dit_set_seqpar: empty chunk %lld", (long long)c);
  d->comm = comm; d->world = world; d->side = (hipStream_t)side_stream;
  d->bounds.assign(bounds, bounds + n_chunks + 1);
  if (!d->ev_ready && hipEventCreateWithFlags(&d->ev_ready, hipEventDisableTiming) != hipSuccess) {
    icv_set_error("icv_dit_set_seqpar: hipEventCreate failed");
    return 2;
  }
  while ((int64_t)d->ev_done.size() < n_chunks) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      icv_set_error("icv_dit_set_seqpar: hipEventCreate failed");
      return 2;
    }
    d->ev_done.push_back(e);
  }
  return 0;
}

#define ICV_HIP_CHECK(expr)                                                \
  do {                                                                     \
    const hipError_t e_ = (expr);                                          \
    if (e_ != hipSuccess) {                                                \
      icv_set_error("%s: %s", #expr, hipGetErrorString(e_));               \
      return 1;                                                            \
    }                                                                      \
  } while (0)

#define DIT_CALL(expr)          \
  do {                          \
    const int rc_ = (expr);     \
    if (rc_) return rc_;        \
  } while (0)

extern "C" int icv_dit_profile(icv_dit* d, int enable) {
  ICV_REQUIRE(d, "icv_dit_profile: null context");
  d->profile = enable != 0;
  d->n_events = 0;
  return 0;
}

extern "C" int icv_dit_profile_read(icv_dit* d, double* total_ms, int64_t* launches) {
  ICV_REQUIRE(d && total_ms && launches, "icv_dit_profile_read: null argument");
  double sum = 0.0;
  for (size_t i = 0; i + 1 < d->n_events; i += 2) {
    ICV_HIP_CHECK(hipEventSynchronize(d->events[i + 1]));
    float ms = 0.f;
    ICV_HIP_CHECK(hipEventElapsedTime(&ms, d->events[i], d->events[i + 1]));
    sum += ms;
  }
  *total_ms = sum;
  *launches = (int64_t)(d->n_events / 2);
  d->n_events = 0;
  return 0;
}

namespace {

// K3 / K8: LayerNorm(+affine)(+modulate) of the residual stream into the A operand of the GEMM with weight W
// (e4m3 rows + scales when that weight is quantised, bf16 otherwise) — dit.py WanDiT._norm
int norm_into(icv_dit& D, const Wt& W, const float* weight, const float* bias, const float* shift, const float* scale, Act& out, void* stream) {
  const icv_dit_config& c = D.cfg;
  if (W.s) {
    ICV_REQUIRE(D.h8 && D.h8s, "icv_dit_forward: bind h8 / h8s for the e4m3 projections");
    DIT_CALL(icv_ln_modulate_fp8(D.x, c.dim, weight, bias, shift, scale, D.h8, c.dim, D.h8s, c.n_tok, c.dim, c.eps, stream));
    out.p = D.h8; out.s = D.h8s;
  } else {
    DIT_CALL(icv_ln_modulate(D.x, c.dim, weight, bias, shift, scale, D.h, c.dim, c.n_tok, c.dim, c.eps, stream));
    out.p = D.h; out.s = nullptr;
  }
  return 0;
}

// bf16 activation [n, K] produced by attention / the GELU epilogue -> A operand of the GEMM with weight W — WanDiT._operand
int operand_of(icv_dit& D, const void* t, int64_t K, void* q8, float* s8, const Wt& W, Act& out, void* stream) {
  if (W.s) {
    ICV_REQUIRE(q8 && s8, "icv_dit_forward: bind att8 / att8s / ff8 / ff8s for the e4m3 projections");
    DIT_CALL(icv_quantize_rows_fp8(t, 0, K, D.cfg.n_tok, K, q8, K, s8, stream));
    out.p = q8; out.s = s8;
  } else {
    out.p = t; out.s = nullptr;
  }
  return 0;
}

// out = epilogue(a @ W[r0 : r0 + N].T + bias[r0 : r0 + N]) — WanDiT._mm
int mm(icv_dit& D, const Act& a, const Wt& W, int64_t r0, int64_t N, int64_t K, const float* bias, void* out, int64_t ldo, int epi,
       int64_t nsplit, int64_t sstride, const float* resid, const float* gate, void* stream) {
  const int64_t n = D.cfg.n_tok;
  ICV_REQUIRE((a.s != nullptr) == (W.s != nullptr), "icv_dit_forward: operand / weight dtype mismatch");
  if (W.s)
    return icv_gemm_fp8(a.p, K, a.s, (const char*)W.w + r0 * K, K, W.s + r0, bias ? bias + r0 : nullptr, n, N, K, epi, out, ldo, nsplit, sstride, resid,
                        D.cfg.dim, gate, stream);
  return icv_gemm_bf16(a.p, K, (const bf16_t*)W.w + r0 * K, K, bias ? bias + r0 : nullptr, n, N, K, epi, out, ldo, nsplit, sstride, resid, D.cfg.dim, gate, stream);
}

}  // namespace

extern "C" int icv_dit_forward(icv_dit* dp, const float* latent, int64_t C, int64_t H8, int64_t W8, const float* mod,
                               const float* hmod, const void* ctx_k, const void* ctx_v, int64_t ctx_len,
                               int64_t ctx_layer_stride, const void* img_k, const void* img_v, int64_t img_len,
                               int64_t img_layer_stride, const float* buf_tokens,
                               float* head_out, int64_t num_layers, int stem, float attn_scale, void* stream) {
  ICV_REQUIRE(dp && latent && mod && hmod && ctx_k && ctx_v && head_out, "icv_dit_forward: null argument");
  icv_dit& D = *dp;
  const icv_dit_config& c = D.cfg;
  const int64_t d = c.dim, n = c.n_tok, L = num_layers < 0 ? c.layers : num_layers;
  ICV_REQUIRE(L <= c.layers, "icv_dit_forward: num_layers > layers");
  ICV_REQUIRE(D.patch_w && D.patch_b && D.head_w && D.head_b && D.rope && D.x && D.h && D.qkv && D.att && D.ff && D.patches,
              "icv_dit_forward: bind patch_w, patch_b, head_w, head_b, rope, x, h, qkv, att, ff, patches first");
  ICV_REQUIRE(stem >= 0 && stem <= 2 && (stem == 0 || D.x_stem), "icv_dit_forward: stem = 0 | 1 (save) | 2 (load); bind x_stem to use it");
  ICV_REQUIRE((img_k == nullptr) == (img_v == nullptr), "icv_dit_forward: image K and V go together");
  const bool sp = D.comm != nullptr;
  if (sp) ICV_REQUIRE(D.kv_loc && D.kv_full && D.sp_acc && D.sp_ml, "icv_dit_forward: bind kv_loc, kv_full, sp_acc, sp_ml for the sequence-parallel schedule");
  if (D.attn_fp8) ICV_REQUIRE(D.a8_qq && D.a8_kq && D.a8_vt && D.a8_amax, "icv_dit_forward: bind a8_qq, a8_kq, a8_vt, a8_amax for e4m3 attention");
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* qkv = (const bf16_t*)D.qkv;
  void* q = (void*)qkv;
  void* k = (void*)(qkv + n * d);
  void* v = (void*)(qkv + 2 * n * d);
  const bool use_stem = stem != 0 && L > 0;
  if (use_stem && stem == 2) {
    if (hipMemcpyAsync(D.x, D.x_stem, (size_t)n * d * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
      icv_set_error("icv_dit_forward: stem copy failed");
      return 2;
    }
  } else {
    // K1: patch embed (+ cached guidance-buffer / conditioning tokens in the GEMM epilogue)
    DIT_CALL(icv_patchify(latent, C, c.T, H8, W8, D.patches, c.k_patch, c.tok0, n, stream));
    DIT_CALL(icv_gemm_bf16(D.patches, c.k_patch, D.patch_w, c.k_patch, D.patch_b, n, d, c.k_patch,
                           buf_tokens ? ICV_EPI_RESID_F32 : ICV_EPI_F32, D.x, d, d, 0, buf_tokens, d, nullptr, stream));
  }
  for (int64_t i = 0; i < L; ++i) {
    const Layer& W = D.layers[(size_t)i];
    ICV_REQUIRE(W.wqkv.w && W.bqkv && W.nq && W.nk && W.wo.w && W.bo && W.n3w && W.n3b && W.xq_w.w && W.xq_b && W.xnq && W.xo_w.w && W.xo_b &&
                W.f0_w.w && W.f0_b && W.f2_w.w && W.f2_b, "icv_dit_forward: layer %lld has unbound tensors", (long long)i);
    const float* m = mod + i * 6 * d;
    const float *sh1 = m, *sc1 = m + d, *g1 = m + 2 * d, *sh2 = m + 3 * d, *sc2 = m + 4 * d, *g2 = m + 5 * d;
    Act a;
    if (!(i == 0 && use_stem && stem == 2)) {
      // icv_dit_profile: one event pair per self-attention LAUNCH on the launch stream - the single launch of a one-rank
      // forward, or each key-chunk launch of the sequence-parallel schedule (recorded after the stream's wait on that chunk's
      // transfer, so exposed transfer time is not part of it; icv_dit_profile_read then counts chunk launches)
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (D.profile && !sp) {
        e0 = D.next_event();
        e1 = D.next_event();
        ICV_REQUIRE(e0 && e1, "icv_dit_forward: hipEventCreate failed");
      }
      DIT_CALL(norm_into(D, W.wqkv, nullptr, nullptr, sh1, sc1, a, stream));                                           // K3
      if (sp) {
        // K and V first, into ONE [n, 2d] matrix (row = k | v), so their exchange (K13) is moving while Q is projected
        const int64_t nc = (int64_t)D.bounds.size() - 1;
        bf16_t* kv_loc = (bf16_t*)D.kv_loc;
        bf16_t* kv_full = (bf16_t*)D.kv_full;
        DIT_CALL(mm(D, a, W.wqkv, d, 2 * d, d, W.bqkv, kv_loc, 2 * d, ICV_EPI_BF16, 2 * d, 0, nullptr, nullptr, stream));   // K4 (k | v rows)
        DIT_CALL(icv_rmsnorm_rope(kv_loc, W.nk, nullptr, nullptr, 2 * d, n, d, c.eps, D.rope, c.T, c.Hp, c.Wp, c.tok0, stream));   // K5 (k)
        ICV_HIP_CHECK(hipEventRecord(D.ev_ready, st));
        ICV_HIP_CHECK(hipStreamWaitEvent(D.side, D.ev_ready, 0));
        for (int64_t ch = 0; ch < nc; ++ch) {      // chunk = rows [r0, r1) of EVERY rank's shard, rank-major, on the side stream
          const int64_t r0 = D.bounds[(size_t)ch], r1 = D.bounds[(size_t)ch + 1];
          DIT_CALL(icv_allgather_kv(D.comm, kv_loc + r0 * 2 * d, kv_full + D.world * r0 * 2 * d, r1 - r0, 2 * d * (int64_t)sizeof(bf16_t), D.side));
          ICV_HIP_CHECK(hipEventRecord(D.ev_done[(size_t)ch], D.side));
        }
        DIT_CALL(mm(D, a, W.wqkv, 0, d, d, W.bqkv, q, d, ICV_EPI_BF16, d, 0, nullptr, nullptr, stream));                // K4 (q)
        DIT_CALL(icv_rmsnorm_rope(q, W.nq, nullptr, nullptr, d, n, d, c.eps, D.rope, c.T, c.Hp, c.Wp, c.tok0, stream));   // K5 (q)
        if (D.attn_fp8)   // queries once per layer, under the first transfer
          DIT_CALL(icv_attention_fp8_prepare(q, d, nullptr, 0, nullptr, 0, n, 0, c.heads, D.a8_qq, d, nullptr, d, nullptr, D.a8_amax, stream));
        for (int64_t ch = 0; ch < nc; ++ch) {      // K6 pipelined with K13: consume chunk ch as soon as it has landed
          const int64_t r0 = D.bounds[(size_t)ch], r1 = D.bounds[(size_t)ch + 1], rows = D.world * (r1 - r0);
          const bf16_t* kc = kv_full + D.world * r0 * 2 * d;
          ICV_HIP_CHECK(hipStreamWaitEvent(st, D.ev_done[(size_t)ch], 0));
          hipEvent_t c0 = nullptr, c1 = nullptr;
          if (D.profile) {
            c0 = D.next_event();
            c1 = D.next_event();
            ICV_REQUIRE(c0 && c1, "icv_dit_forward: hipEventCreate failed");
            ICV_HIP_CHECK(hipEventRecord(c0, st));
          }
          if (D.attn_fp8) {
            DIT_CALL(icv_attention_fp8_prepare(nullptr, 0, kc, 2 * d, kc + d, 2 * d, 0, rows, c.heads, nullptr, d, D.a8_kq, d, D.a8_vt, D.a8_amax, stream));
            DIT_CALL(icv_attention_fp8_fwd_chunk(D.a8_qq, d, D.a8_kq, d, D.a8_vt, D.a8_amax, D.att, d, D.sp_acc, d, D.sp_ml, n, rows, c.heads,
                                                 ch == 0, ch == nc - 1, stream));
          } else {
            DIT_CALL(icv_attention_fwd_chunk(q, d, kc, 2 * d, kc + d, 2 * d, D.att, d, D.sp_acc, d, D.sp_ml, n, rows, c.heads, attn_scale,
                                             ch == 0, ch == nc - 1, stream));
          }
          if (c1) ICV_HIP_CHECK(hipEventRecord(c1, st));
        }
      } else {
        // single rank: K4 (one fused QKV GEMM, split planes), K5, K6
        DIT_CALL(mm(D, a, W.wqkv, 0, 3 * d, d, W.bqkv, D.qkv, d, ICV_EPI_BF16, d, n * d, nullptr, nullptr, stream));
        DIT_CALL(icv_rmsnorm_rope(q, W.nq, k, W.nk, d, n, d, c.eps, D.rope, c.T, c.Hp, c.Wp, c.tok0, stream));
        if (e0) ICV_HIP_CHECK(hipEventRecord(e0, st));
        if (D.attn_fp8) {
          DIT_CALL(icv_attention_fp8_prepare(q, d, k, d, v, d, n, n, c.heads, D.a8_qq, d, D.a8_kq, d, D.a8_vt, D.a8_amax, stream));
          DIT_CALL(icv_attention_fp8_fwd(D.a8_qq, d, D.a8_kq, d, D.a8_vt, D.a8_amax, D.att, d, n, n, c.heads, stream));
        } else {
          DIT_CALL(icv_attention_fwd(q, d, k, d, v, d, D.att, d, n, n, c.heads, attn_scale, stream));
        }
        if (e1) ICV_HIP_CHECK(hipEventRecord(e1, st));
      }
      DIT_CALL(operand_of(D, D.att, d, D.att8, D.att8s, W.wo, a, stream));
      DIT_CALL(mm(D, a, W.wo, 0, d, d, W.bo, D.x, d, ICV_EPI_RESID_F32, d, 0, D.x, g1, stream));                        // K7
      if (i == 0 && use_stem && stem == 1) {
        if (hipMemcpyAsync(D.x_stem, D.x, (size_t)n * d * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
          icv_set_error("icv_dit_forward: stem copy failed");
          return 2;
        }
      }
    }
    // cross-attention to the text (and, i2v, the image) tokens: K8, K9
    DIT_CALL(norm_into(D, W.xq_w, W.n3w, W.n3b, nullptr, nullptr, a, stream));
    DIT_CALL(mm(D, a, W.xq_w, 0, d, d, W.xq_b, q, d, ICV_EPI_BF16, d, 0, nullptr, nullptr, stream));
    DIT_CALL(icv_rmsnorm_rope(q, W.xnq, nullptr, nullptr, d, n, d, c.eps, nullptr, 0, 0, 0, 0, stream));
    const bf16_t* ck = (const bf16_t*)ctx_k + i * ctx_layer_stride;
    const bf16_t* cv = (const bf16_t*)ctx_v + i * ctx_layer_stride;
    DIT_CALL(icv_attention_fwd(q, d, ck, d, cv, d, D.att, d, n, ctx_len, c.heads, attn_scale, stream));
    if (img_k) {
      const bf16_t* ik = (const bf16_t*)img_k + i * img_layer_stride;
      const bf16_t* iv = (const bf16_t*)img_v + i * img_layer_stride;
      DIT_CALL(icv_attention_fwd_add(q, d, ik, d, iv, d, D.att, d, n, img_len, c.heads, attn_scale, stream));
    }
    DIT_CALL(operand_of(D, D.att, d, D.att8, D.att8s, W.xo_w, a, stream));
    DIT_CALL(mm(D, a, W.xo_w, 0, d, d, W.xo_b, D.x, d, ICV_EPI_RESID_F32, d, 0, D.x, nullptr, stream));
    // FFN: K3, K10
    DIT_CALL(norm_into(D, W.f0_w, nullptr, nullptr, sh2, sc2, a, stream));
    DIT_CALL(mm(D, a, W.f0_w, 0, c.ffn_dim, d, W.f0_b, D.ff, c.ffn_dim, ICV_EPI_GELU_BF16, c.ffn_dim, 0, nullptr, nullptr, stream));
    DIT_CALL(operand_of(D, D.ff, c.ffn_dim, D.ff8, D.ff8s, W.f2_w, a, stream));
    DIT_CALL(mm(D, a, W.f2_w, 0, d, c.ffn_dim, W.f2_b, D.x, d, ICV_EPI_RESID_F32, d, 0, D.x, g2, stream));
  }
  // K11: head
  DIT_CALL(icv_ln_modulate(D.x, d, nullptr, nullptr, hmod, hmod + d, D.h, d, n, d, c.eps, stream));
  DIT_CALL(icv_gemm_bf16(D.h, d, D.head_w, d, D.head_b, n, c.out_cols, d, ICV_EPI_F32, head_out, c.out_cols, c.out_cols, 0, nullptr, 0, nullptr, stream));
  return 0;
}
