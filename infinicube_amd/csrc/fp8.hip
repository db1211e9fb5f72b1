// Row-scaled fp8 (OCP e4m3) quantisation for the fp8 GEMM path (BASELINE.json config #5).
//   scale[r] = max_k |x[r,k]| / 448  (1.0 for an all-zero row);   q[r,k] = e4m3_rne( x[r,k] * (1 / scale[r]) )
// One scale per token row (activations, recomputed every call) or per output channel (weights, once at load).
// e4m3's floating exponent keeps ~2^-4 relative precision over 2^15 of range below the row maximum, so a
// row scale is enough — no per-block scales are needed for LayerNorm'd activations with outlier channels.
#include "icv_common.h"

namespace {

constexpr float FP8_MAX = 448.0f;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// 4 floats -> 4 e4m3 bytes (v_cvt_pk_fp8_f32: round-to-nearest-even; inputs are pre-scaled into [-448, 448])
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}

// One wave per row; the row is read twice (abs-max pass, convert pass): the second read hits L2.
template <bool SRC_F32>
__global__ __launch_bounds__(256) void quantize_rows_kernel(const void* __restrict__ src, int64_t ld, int64_t rows,
                                                            int K, unsigned char* __restrict__ out, int64_t ldo,
                                                            float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = K >> 3;   // 8 elements per lane per step
  float amax = 0.f;
  if (SRC_F32) {
    const float4* xr = reinterpret_cast<const float4*>((const float*)src + row * ld);
    for (int c = lane; c < nvec; c += 64) {
      const float4 a = xr[2 * c], b = xr[2 * c + 1];
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
    }
  } else {
    const uint4* xr = reinterpret_cast<const uint4*>((const bf16_t*)src + row * ld);
    for (int c = lane; c < nvec; c += 64) {
      const uint4 u = xr[c];
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
        amax = fmaxf(amax, fmaxf(fabsf(__uint_as_float(w[i] << 16)), fabsf(__uint_as_float(w[i] & 0xFFFF0000u))));
    }
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax / FP8_MAX : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  uint2* orow = reinterpret_cast<uint2*>(out + row * ldo);
  if (SRC_F32) {
    const float4* xr = reinterpret_cast<const float4*>((const float*)src + row * ld);
    for (int c = lane; c < nvec; c += 64) {
      const float4 a = xr[2 * c], b = xr[2 * c + 1];
      orow[c] = make_uint2(pack_fp8x4(a.x * inv, a.y * inv, a.z * inv, a.w * inv),
                           pack_fp8x4(b.x * inv, b.y * inv, b.z * inv, b.w * inv));
    }
  } else {
    const uint4* xr = reinterpret_cast<const uint4*>((const bf16_t*)src + row * ld);
    for (int c = lane; c < nvec; c += 64) {
      const uint4 u = xr[c];
      const float f0 = __uint_as_float(u.x << 16), f1 = __uint_as_float(u.x & 0xFFFF0000u);
      const float f2 = __uint_as_float(u.y << 16), f3 = __uint_as_float(u.y & 0xFFFF0000u);
      const float f4 = __uint_as_float(u.z << 16), f5 = __uint_as_float(u.z & 0xFFFF0000u);
      const float f6 = __uint_as_float(u.w << 16), f7 = __uint_as_float(u.w & 0xFFFF0000u);
      orow[c] = make_uint2(pack_fp8x4(f0 * inv, f1 * inv, f2 * inv, f3 * inv),
                           pack_fp8x4(f4 * inv, f5 * inv, f6 * inv, f7 * inv));
    }
  }
}

}  // namespace

extern "C" int icv_quantize_rows_fp8(const void* src, int src_is_f32, int64_t ld, int64_t rows, int64_t K, void* out,
                                     int64_t ldo, float* scale, void* stream) {
  ICV_REQUIRE(src && out && scale, "icv_quantize_rows_fp8: null pointer");
  ICV_REQUIRE(rows > 0 && K > 0 && K % 8 == 0 && K <= (1 << 30), "icv_quantize_rows_fp8: K=%lld must be a positive multiple of 8", (long long)K);
  ICV_REQUIRE(ld % 8 == 0 && ldo % 8 == 0, "icv_quantize_rows_fp8: ld / ldo must keep 16-byte (src) and 8-byte (out) row alignment");
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (src_is_f32)
    hipLaunchKernelGGL(quantize_rows_kernel<true>, grid, block, 0, st, src, ld, rows, (int)K, (unsigned char*)out, ldo, scale);
  else
    hipLaunchKernelGGL(quantize_rows_kernel<false>, grid, block, 0, st, src, ld, rows, (int)K, (unsigned char*)out, ldo, scale);
  return icv_check_launch("icv_quantize_rows_fp8");
}
