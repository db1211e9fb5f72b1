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
    ICV_REQUIRE(W.wqkv && W.bqkv && W.nq && W.nk && W.wo && W.bo && W.n3w && W.n3b && W.xq_w && W.xq_b && W.xnq && W.xo_w && W.xo_b &&
                W.f0_w && W.f0_b && W.f2_w && W.f2_b, "icv_dit_forward: layer %lld has unbound tensors", (long long)i);
    const float* m = mod + i * 6 * d;
    const float *sh1 = m, *sc1 = m + d, *g1 = m + 2 * d, *sh2 = m + 3 * d, *sc2 = m + 4 * d, *g2 = m + 5 * d;
    if (!(i == 0 && use_stem && stem == 2)) {
      // self-attention: K3, K4 (one fused QKV GEMM, split planes), K5, K6, K7
      DIT_CALL(icv_ln_modulate(D.x, d, nullptr, nullptr, sh1, sc1, D.h, d, n, d, c.eps, stream));
      DIT_CALL(icv_gemm_bf16(D.h, d, W.wqkv, d, W.bqkv, n, 3 * d, d, ICV_EPI_BF16, D.qkv, d, d, n * d, nullptr, 0, nullptr, stream));
      DIT_CALL(icv_rmsnorm_rope(q, W.nq, k, W.nk, d, n, d, c.eps, D.rope, c.T, c.Hp, c.Wp, c.tok0, stream));
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (D.profile) {
        e0 = D.next_event();
        e1 = D.next_event();
        ICV_REQUIRE(e0 && e1, "icv_dit_forward: hipEventCreate failed");
        ICV_HIP_CHECK(hipEventRecord(e0, st));
      }
      DIT_CALL(icv_attention_fwd(q, d, k, d, v, d, D.att, d, n, n, c.heads, attn_scale, stream));
      if (D.profile) ICV_HIP_CHECK(hipEventRecord(e1, st));
      DIT_CALL(icv_gemm_bf16(D.att, d, W.wo, d, W.bo, n, d, d, ICV_EPI_RESID_F32, D.x, d, d, 0, D.x, d, g1, stream));
      if (i == 0 && use_stem && stem == 1) {
        if (hipMemcpyAsync(D.x_stem, D.x, (size_t)n * d * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
          icv_set_error("icv_dit_forward: stem copy failed");
          return 2;
        }
      }
    }
    // cross-attention to the text (and, i2v, the image) tokens: K8, K9
    DIT_CALL(icv_ln_modulate(D.x, d, W.n3w, W.n3b, nullptr, nullptr, D.h, d, n, d, c.eps, stream));
    DIT_CALL(icv_gemm_bf16(D.h, d, W.xq_w, d, W.xq_b, n, d, d, ICV_EPI_BF16, q, d, d, 0, nullptr, 0, nullptr, stream));
    DIT_CALL(icv_rmsnorm_rope(q, W.xnq, nullptr, nullptr, d, n, d, c.eps, nullptr, 0, 0, 0, 0, stream));
    const bf16_t* ck = (const bf16_t*)ctx_k + i * ctx_layer_stride;
    const bf16_t* cv = (const bf16_t*)ctx_v + i * ctx_layer_stride;
    DIT_CALL(icv_attention_fwd(q, d, ck, d, cv, d, D.att, d, n, ctx_len, c.heads, attn_scale, stream));
    if (img_k) {
      const bf16_t* ik = (const bf16_t*)img_k + i * img_layer_stride;
      const bf16_t* iv = (const bf16_t*)img_v + i * img_layer_stride;
      DIT_CALL(icv_attention_fwd_add(q, d, ik, d, iv, d, D.att, d, n, img_len, c.heads, attn_scale, stream));
    }
    DIT_CALL(icv_gemm_bf16(D.att, d, W.xo_w, d, W.xo_b, n, d, d, ICV_EPI_RESID_F32, D.x, d, d, 0, D.x, d, nullptr, stream));
    // FFN: K3, K10
    DIT_CALL(icv_ln_modulate(D.x, d, nullptr, nullptr, sh2, sc2, D.h, d, n, d, c.eps, stream));
    DIT_CALL(icv_gemm_bf16(D.h, d, W.f0_w, d, W.f0_b, n, c.ffn_dim, d, ICV_EPI_GELU_BF16, D.ff, c.ffn_dim, c.ffn_dim, 0, nullptr, 0, nullptr, stream));
    DIT_CALL(icv_gemm_bf16(D.ff, c.ffn_dim, W.f2_w, c.ffn_dim, W.f2_b, n, d, c.ffn_dim, ICV_EPI_RESID_F32, D.x, d, d, 0, D.x, d, g2, stream));
  }
  // K11: head
  DIT_CALL(icv_ln_modulate(D.x, d, nullptr, nullptr, hmod, hmod + d, D.h, d, n, d, c.eps, stream));
  DIT_CALL(icv_gemm_bf16(D.h, d, D.head_w, d, D.head_b, n, c.out_cols, d, ICV_EPI_F32, head_out, c.out_cols, c.out_cols, 0, nullptr, 0, nullptr, stream));
  return 0;
}
