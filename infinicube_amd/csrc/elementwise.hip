// HBM-bound kernels of the DiT forward.
// Design: one 64-lane wavefront owns one token row, keeps the whole row in VGPRs (d <= 8192),
// reduces with wave shuffles (no LDS, no block barrier), 16-byte accesses per lane, coalesced.
#include "icv_common.h"

// ---------------------------------------------------------------------------------------------
// K3 / K8  LayerNorm (+affine) (+modulate): x f32 [rows,d] -> out bf16 [rows,d]
// A row is owned by W waves of a 256-thread block (4/W rows per block); each lane keeps NV float4 of
// its row in VGPRs (two-pass fp32 statistics), so d = 256 * W * NV.  W > 1 keeps the per-lane
// footprint small for wide rows (d = 5120: W = 4, NV = 5 instead of 20 vectors per lane, which halved
// the achieved bandwidth); the W partial sums meet through 16 bytes of LDS.
// ---------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ float row_sum(float v, float* red, int wave) {
  v = wave_sum(v);
  if (W == 1) return v;
  __syncthreads();                       // previous use of red[] is over
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  const int base = (wave / W) * W;
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < W; ++i) t += red[base + i];
  return t;
}

template <int W>
__device__ __forceinline__ float row_max(float v, float* red, int wave) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if (W == 1) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  const int base = (wave / W) * W;
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < W; ++i) t = fmaxf(t, red[base + i]);
  return t;
}

// FP8 = true (fp8 GEMM path, BASELINE config #5): the modulated row is written as OCP e4m3 bytes with one
// f32 scale per row (= row abs-max / 448) instead of bf16 — the abs-max is one more row reduction on values
// that are already in registers, so the activation is quantised for free in the kernel that produces it.
template <int NV, int W, bool FP8>
__global__ __launch_bounds__(256) void ln_modulate_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ weight,
    const float* __restrict__ bias, const float* __restrict__ shift,
    const float* __restrict__ scale, void* __restrict__ out_, int64_t ldo, float* __restrict__ out_scale,
    int64_t rows, int d, float eps) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  constexpr int RPB = 4 / W;             // rows per block
  const int64_t row_raw = (int64_t)blockIdx.x * RPB + wave / W;
  const bool live = row_raw < rows;      // dead rows still take part in the barriers of row_sum
  const int64_t row = live ? row_raw : rows - 1;
  const int sub = wave % W;              // which 1/W of the row this wave owns
  const float4* xr = reinterpret_cast<const float4*>(x + row * ldx);
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = xr[(i * W + sub) * 64 + lane];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = row_sum<W>(s, red, wave) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  const float rstd = rsqrtf(row_sum<W>(q, red, wave) / (float)d + eps);
  if (!FP8 && !live) return;
  bf16_t* out = reinterpret_cast<bf16_t*>(out_);
  uint2* orow = reinterpret_cast<uint2*>(out + row * ldo);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = (i * W + sub) * 64 + lane;  // float4 index within the row
    float4 y = make_float4(v[i].x * rstd, v[i].y * rstd, v[i].z * rstd, v[i].w * rstd);
    if (weight) {
      const float4 w = reinterpret_cast<const float4*>(weight)[c4];
      y.x *= w.x; y.y *= w.y; y.z *= w.z; y.w *= w.w;
    }
    if (bias) {
      const float4 b = reinterpret_cast<const float4*>(bias)[c4];
      y.x += b.x; y.y += b.y; y.z += b.z; y.w += b.w;
    }
    if (scale) {
      const float4 sc = reinterpret_cast<const float4*>(scale)[c4];
      y.x *= 1.f + sc.x; y.y *= 1.f + sc.y; y.z *= 1.f + sc.z; y.w *= 1.f + sc.w;
    }
    if (shift) {
      const float4 sh = reinterpret_cast<const float4*>(shift)[c4];
      y.x += sh.x; y.y += sh.y; y.z += sh.z; y.w += sh.w;
    }
    if (FP8) {
      v[i] = y;
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w))));
    } else {
      orow[c4] = make_uint2(pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
    }
  }
  if (FP8) {
    amax = row_max<W>(amax, red, wave);
    if (!live) return;
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    const float inv = 1.0f / sc;
    if (sub == 0 && lane == 0) out_scale[row] = sc;
    unsigned* qrow = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(out_) + row * ldo);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].x * inv, v[i].y * inv, 0, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].z * inv, v[i].w * inv, w, true);
      qrow[(i * W + sub) * 64 + lane] = (unsigned)w;
    }
  }
}

template <bool FP8>
static int ln_modulate_launch(const char* who, const float* x, int64_t ldx, const float* weight, const float* bias,
                              const float* shift, const float* scale, void* out, int64_t ldo, float* out_scale,
                              int64_t rows, int64_t d, float eps, void* stream) {
  ICV_REQUIRE(rows > 0 && d > 0 && d % 256 == 0 && d <= 8192, "%s: d=%lld must be a multiple of 256 and <= 8192", who, (long long)d);
  ICV_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0, "%s: ldx/ldo must be multiples of 4", who);
  hipStream_t st = (hipStream_t)stream;
  const int nvec = (int)(d / 256);                       // float4 per lane at one wave per row
  int forced = icv_get_option_int("ln_waves_per_row", 0);
  const int W = (forced == 1 || forced == 2 || forced == 4) && nvec % forced == 0 ? forced
              : (nvec >= 16 && nvec % 4 == 0) ? 4 : (nvec >= 8 && nvec % 2 == 0) ? 2 : 1;
  const int NV = nvec / W;
  dim3 grid((unsigned)((rows * W + 3) / 4)), block(256);
#define LAUNCH(NV_, W_)                                                                           \
  if (NV == NV_ && W == W_) {                                                                     \
    hipLaunchKernelGGL((ln_modulate_kernel<NV_, W_, FP8>), grid, block, 0, st, x, ldx, weight, bias, shift, \
                       scale, out, ldo, out_scale, rows, (int)d, eps);                            \
    return icv_check_launch(who);                                                                 \
  }
  LAUNCH(1, 1) LAUNCH(2, 1) LAUNCH(3, 1) LAUNCH(4, 1) LAUNCH(5, 1) LAUNCH(6, 1) LAUNCH(7, 1)
  LAUNCH(3, 2) LAUNCH(4, 2) LAUNCH(5, 2) LAUNCH(6, 2) LAUNCH(7, 2)
  LAUNCH(4, 4) LAUNCH(5, 4) LAUNCH(6, 4) LAUNCH(7, 4) LAUNCH(8, 4)
  LAUNCH(8, 1) LAUNCH(10, 1) LAUNCH(12, 1) LAUNCH(20, 1) LAUNCH(10, 2) LAUNCH(2, 4) LAUNCH(3, 4)
#undef LAUNCH
  icv_set_error("%s: unsupported d=%lld (no instantiation for %d vectors x %d waves)", who, (long long)d, NV, W);
  return 1;
}

extern "C" int icv_ln_modulate(const float* x, int64_t ldx, const float* weight, const float* bias,
                               const float* shift, const float* scale, void* out, int64_t ldo,
                               int64_t rows, int64_t d, float eps, void* stream) {
  return ln_modulate_launch<false>("icv_ln_modulate", x, ldx, weight, bias, shift, scale, out, ldo, nullptr, rows, d, eps, stream);
}

extern "C" int icv_ln_modulate_fp8(const float* x, int64_t ldx, const float* weight, const float* bias,
                                   const float* shift, const float* scale, void* out_fp8, int64_t ldo,
                                   float* out_scale, int64_t rows, int64_t d, float eps, void* stream) {
  ICV_REQUIRE(out_scale, "icv_ln_modulate_fp8: null out_scale");
  return ln_modulate_launch<true>("icv_ln_modulate_fp8", x, ldx, weight, bias, shift, scale, out_fp8, ldo, out_scale, rows, d, eps, stream);
}

// ---------------------------------------------------------------------------------------------
// K5  RMSNorm over full d (+ 3-D RoPE) in place on bf16.  One wave per row, 8 bf16 per lane per
// vector; NV = ceil(d / 512).  Each lane always owns in-head pair positions p0 = (lane&15)*4..+3
// (because 64 lanes * 8 elems = 512 = 4 heads), so its 4 (cos,sin) pairs are loaded once per row.
// ---------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(
    bf16_t* __restrict__ x0, const float* __restrict__ w0, bf16_t* __restrict__ x1,
    const float* __restrict__ w1, int64_t ld, int64_t rows, int d, float eps,
    const float2* __restrict__ rope_tab, int T, int Hp, int Wp, int64_t tok0) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  bf16_t* x = blockIdx.y ? x1 : x0;
  const float* w = blockIdx.y ? w1 : w0;
  uint4* xr = reinterpret_cast<uint4*>(x + row * ld);
  const int nvec = d >> 3;
  uint4 raw[NV];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c8 = i * 64 + lane;
    raw[i] = (c8 < nvec) ? xr[c8] : make_uint4(0, 0, 0, 0);
    const unsigned u[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = bf16lo_to_f32(u[j]), b = bf16hi_to_f32(u[j]);
      ss += a * a + b * b;
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)d + eps);
  float2 cs[4];
  if (rope_tab) {
    const int64_t tok = tok0 + row;
    const int wp = (int)(tok % Wp);
    const int hp = (int)((tok / Wp) % Hp);
    const int fr = (int)(tok / ((int64_t)Wp * Hp));
    const float2* tf = rope_tab + (int64_t)fr * 22;
    const float2* th = rope_tab + (int64_t)T * 22 + (int64_t)hp * 21;
    const float2* tw = rope_tab + (int64_t)T * 22 + (int64_t)Hp * 21 + (int64_t)wp * 21;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = (lane & 15) * 4 + j;  // pair index within the 128-wide head: 0..63
      cs[j] = (p < 22) ? tf[p] : (p < 43 ? th[p - 22] : tw[p - 43]);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c8 = i * 64 + lane;
    if (c8 >= nvec) continue;
    const float4 wa = reinterpret_cast<const float4*>(w)[c8 * 2];
    const float4 wb = reinterpret_cast<const float4*>(w)[c8 * 2 + 1];
    const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
    const unsigned u[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = bf16lo_to_f32(u[j]) * rstd * wv[2 * j];
      float b = bf16hi_to_f32(u[j]) * rstd * wv[2 * j + 1];
      if (rope_tab) {
        const float c = cs[j].x, s = cs[j].y;
        const float ra = a * c - b * s, rb = a * s + b * c;
        a = ra; b = rb;
      }
      o[j] = pack_bf16x2(a, b);
    }
    xr[c8] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

extern "C" int icv_rmsnorm_rope(void* x0, const float* w0, void* x1, const float* w1, int64_t ld,
                                int64_t rows, int64_t d, float eps, const float* rope_tab,
                                int64_t T, int64_t Hp, int64_t Wp, int64_t tok0, void* stream) {
  ICV_REQUIRE(rows > 0 && d > 0 && d % 128 == 0 && d <= 8192, "icv_rmsnorm_rope: d=%lld must be a multiple of 128 and <= 8192", (long long)d);
  ICV_REQUIRE(ld % 8 == 0, "icv_rmsnorm_rope: ld must be a multiple of 8");
  ICV_REQUIRE(x0 && w0 && (!x1 || w1), "icv_rmsnorm_rope: null tensor");
  if (rope_tab) ICV_REQUIRE(T > 0 && Hp > 0 && Wp > 0 && tok0 >= 0 && tok0 + rows <= T * Hp * Wp, "icv_rmsnorm_rope: token range outside the (T,Hp,Wp) grid");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4), x1 ? 2 : 1), block(256);
  const int nv = (int)((d + 511) / 512);
#define LAUNCH(NV)                                                                              \
  case NV:                                                                                      \
    hipLaunchKernelGGL(rmsnorm_rope_kernel<NV>, grid, block, 0, st, (bf16_t*)x0, w0, (bf16_t*)x1, \
                       w1, ld, rows, (int)d, eps, (const float2*)rope_tab, (int)T, (int)Hp,     \
                       (int)Wp, tok0);                                                          \
    break;
  switch (nv) {
    LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(8) LAUNCH(10) LAUNCH(12) LAUNCH(16)
    default:
      icv_set_error("icv_rmsnorm_rope: unsupported d=%lld", (long long)d);
      return 1;
  }
#undef LAUNCH
  return icv_check_launch("icv_rmsnorm_rope");
}

// ---------------------------------------------------------------------------------------------
// K1 im2col (patchify): latent f32 [C,T,H8,W8] -> bf16 [n_tok, C*4]; one thread per (token, c)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ lat, int C, int T,
                                                       int H8, int W8, bf16_t* __restrict__ out,
                                                       int64_t ldo, int64_t tok0, int64_t n_tok) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_tok * C) return;
  const int c = (int)(idx % C);
  const int64_t r = idx / C;
  const int64_t tok = tok0 + r;
  const int Wp = W8 >> 1, Hp = H8 >> 1;
  const int wp = (int)(tok % Wp);
  const int hp = (int)((tok / Wp) % Hp);
  const int f = (int)(tok / ((int64_t)Wp * Hp));
  const float* p = lat + (((int64_t)c * T + f) * H8 + 2 * hp) * W8 + 2 * wp;
  const float2 a = *reinterpret_cast<const float2*>(p);
  const float2 b = *reinterpret_cast<const float2*>(p + W8);
  *reinterpret_cast<uint2*>(out + r * ldo + c * 4) =
      make_uint2(pack_bf16x2(a.x, a.y), pack_bf16x2(b.x, b.y));
}

extern "C" int icv_patchify(const float* latent, int64_t C, int64_t T, int64_t H8, int64_t W8,
                            void* out, int64_t ldo, int64_t tok0, int64_t n_tok, void* stream) {
  ICV_REQUIRE(H8 % 2 == 0 && W8 % 2 == 0 && ldo % 4 == 0 && n_tok > 0, "icv_patchify: bad shape");
  ICV_REQUIRE(tok0 >= 0 && tok0 + n_tok <= T * (H8 / 2) * (W8 / 2), "icv_patchify: token range");
  const int64_t total = n_tok * C;
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, latent, (int)C, (int)T, (int)H8, (int)W8, (bf16_t*)out,
                     ldo, tok0, n_tok);
  return icv_check_launch("icv_patchify");
}

// ---------------------------------------------------------------------------------------------
// K11 tail + K12: unpatchify + CFG + Euler.  One thread per (token, y, c): handles z = 0,1.
// head-out column = (y*2+z)*C + c   ['(f h w) (x y z c) -> c (f x) (h y) (w z)']
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unpatchify_cfg_euler_kernel(
    float* __restrict__ lat, float* __restrict__ vel, const float* __restrict__ hc,
    const float* __restrict__ hu, int64_t ldh, float cfg, float dsigma, int C, int T, int H8,
    int W8, int64_t tok0, int64_t n_tok, int round_bf16) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_tok * 2 * C) return;
  const int c = (int)(idx % C);
  const int y = (int)((idx / C) & 1);
  const int64_t r = idx / (2 * C);
  const int64_t tok = tok0 + r;
  const int Wp = W8 >> 1, Hp = H8 >> 1;
  const int wp = (int)(tok % Wp);
  const int hp = (int)((tok / Wp) % Hp);
  const int f = (int)(tok / ((int64_t)Wp * Hp));
  const int64_t h0 = r * ldh + (int64_t)(y * 2) * C + c;
  float v0 = hc[h0], v1 = hc[h0 + C];
  const int64_t li = (((int64_t)c * T + f) * H8 + 2 * hp + y) * W8 + 2 * wp;
  float2* lp = reinterpret_cast<float2*>(lat + li);
  float2 l = *lp;
  if (round_bf16) {
    // "reference rounding": a pipeline that keeps noise_pred and the latents in torch_dtype = bf16 rounds after every
    // tensor op — the model outputs, (c - u), cfg * (.), u + (.), v * dsigma and the updated latent ([EXT], ORACLE_RISKS.md)
    auto rb = [](float x) { return bf16_to_f32((bf16_t)f32_to_bf16_bits(x)); };
    v0 = rb(v0); v1 = rb(v1);
    if (hu) {
      const float u0 = rb(hu[h0]), u1 = rb(hu[h0 + C]);
      v0 = rb(u0 + rb(cfg * rb(v0 - u0)));
      v1 = rb(u1 + rb(cfg * rb(v1 - u1)));
    }
    l.x = rb(rb(l.x) + rb(v0 * dsigma));
    l.y = rb(rb(l.y) + rb(v1 * dsigma));
  } else {
    if (hu) {
      const float u0 = hu[h0], u1 = hu[h0 + C];
      v0 = u0 + cfg * (v0 - u0);
      v1 = u1 + cfg * (v1 - u1);
    }
    l.x += v0 * dsigma;
    l.y += v1 * dsigma;
  }
  *lp = l;
  if (vel) *reinterpret_cast<float2*>(vel + li) = make_float2(v0, v1);
}

extern "C" int icv_unpatchify_cfg_euler(float* latent, float* vel_out, const float* hc,
                                        const float* hu, int64_t ldh, float cfg_scale,
                                        float dsigma, int64_t C, int64_t T, int64_t H8, int64_t W8,
                                        int64_t tok0, int64_t n_tok, int round_bf16, void* stream) {
  ICV_REQUIRE(H8 % 2 == 0 && W8 % 2 == 0 && n_tok > 0 && ldh >= 4 * C, "icv_unpatchify_cfg_euler: bad shape");
  ICV_REQUIRE(tok0 >= 0 && tok0 + n_tok <= T * (H8 / 2) * (W8 / 2), "icv_unpatchify_cfg_euler: token range");
  const int64_t total = n_tok * 2 * C;
  hipLaunchKernelGGL(unpatchify_cfg_euler_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256),
                     0, (hipStream_t)stream, latent, vel_out, hc, hu, ldh, cfg_scale, dsigma,
                     (int)C, (int)T, (int)H8, (int)W8, tok0, n_tok, round_bf16);
  return icv_check_launch("icv_unpatchify_cfg_euler");
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bcast_add_kernel(const float* __restrict__ a,
                                                        const float* __restrict__ b,
                                                        float* __restrict__ out, int64_t total,
                                                        int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total) out[i] = a[i] + b[i % n];
}
extern "C" int icv_bcast_add_f32(const float* a, const float* b, float* out, int64_t rows,
                                 int64_t n, void* stream) {
  ICV_REQUIRE(rows > 0 && n > 0, "icv_bcast_add_f32: bad shape");
  const int64_t total = rows * n;
  hipLaunchKernelGGL(bcast_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, a, b, out, total, n);
  return icv_check_launch("icv_bcast_add_f32");
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in,
                                                        bf16_t* __restrict__ out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(in + i);
    *reinterpret_cast<uint2*>(out + i) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  } else {
    for (int64_t j = i; j < n; ++j) out[j] = (bf16_t)f32_to_bf16_bits(in[j]);
  }
}
extern "C" int icv_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
  ICV_REQUIRE(n > 0, "icv_cast_f32_to_bf16: n <= 0");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0,
                     (hipStream_t)stream, in, (bf16_t*)out, n);
  return icv_check_launch("icv_cast_f32_to_bf16");
}

// K2: sinusoidal embedding in fp64 (as upstream): out = cat[cos(t f_i), sin(t f_i)]
__global__ void sinus_kernel(double t, int half, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half) return;
  const double f = pow(10000.0, -(double)i / (double)half);
  out[i] = (float)cos(t * f);
  out[half + i] = (float)sin(t * f);
}
extern "C" int icv_sinusoidal_embedding(double timestep, int64_t dim, float* out, void* stream) {
  ICV_REQUIRE(dim > 0 && dim % 2 == 0, "icv_sinusoidal_embedding: dim must be even");
  const int half = (int)(dim / 2);
  hipLaunchKernelGGL(sinus_kernel, dim3((half + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                     timestep, half, out);
  return icv_check_launch("icv_sinusoidal_embedding");
}

// K2: small-M GEMV, fp32 activations x bf16 weights.  One wave per output column n; lanes stride
// over K with 16-byte weight loads; M <= 8 rows accumulated together (weights read once).
template <int MM>
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ x,
                                                   const bf16_t* __restrict__ W,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ out, int64_t N, int64_t K,
                                                   int in_act, int out_act) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) acc[m] = 0.f;
  const uint4* wr = reinterpret_cast<const uint4*>(W + n * K);
  for (int64_t k8 = lane; k8 < (K >> 3); k8 += 64) {
    const uint4 wv = wr[k8];
    const unsigned u[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
    for (int m = 0; m < MM; ++m) {
      const float4 xa = *reinterpret_cast<const float4*>(x + m * K + k8 * 8);
      const float4 xb = *reinterpret_cast<const float4*>(x + m * K + k8 * 8 + 4);
      float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
      if (in_act == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = silu(xv[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[m] += xv[2 * j] * bf16lo_to_f32(u[j]) + xv[2 * j + 1] * bf16hi_to_f32(u[j]);
    }
  }
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    float v = wave_sum(acc[m]);
    if (lane == 0) {
      if (bias) v += bias[n];
      if (out_act == 1) v = silu(v);
      out[m * N + n] = v;
    }
  }
}
extern "C" int icv_gemv_f32(const float* x, const void* W, const float* bias, float* out,
                            int64_t M, int64_t N, int64_t K, int in_act, int out_act,
                            void* stream) {
  ICV_REQUIRE(M >= 1 && M <= 8 && N > 0 && K > 0 && K % 8 == 0, "icv_gemv_f32: need 1<=M<=8, K%%8==0");
  dim3 grid((unsigned)((N + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(MM)                                                                              \
  case MM:                                                                                      \
    hipLaunchKernelGGL(gemv_kernel<MM>, grid, block, 0, st, x, (const bf16_t*)W, bias, out, N, K, \
                       in_act, out_act);                                                        \
    break;
  switch (M) { LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8) }
#undef LAUNCH
  return icv_check_launch("icv_gemv_f32");
}
