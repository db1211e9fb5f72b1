// Shared device/host helpers for libicvideo (gfx950 only; wavefront = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "icvideo.h"

typedef unsigned short bf16_t;  // raw storage
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define ICV_WAVE 64

void icv_set_error(const char* fmt, ...);
int icv_check_launch(const char* what);
int icv_get_option_int(const char* name, int dflt);  // runtime A/B switches (icv_set_option)

// Kernels that need more than 64 KiB of dynamic LDS raise hipFuncAttributeMaxDynamicSharedMemorySize once.  The attribute
// belongs to the (function, DEVICE) pair, so the "already done" flag is per device: a process that drives several GPUs
// (one host thread per device, hipSetDevice before each call) gets it set on each of them.
#define ICV_MAX_DEVICES 64
struct icv_dev_flags { bool set[ICV_MAX_DEVICES]; };
int icv_ensure_dynamic_lds(const void* func, int bytes, icv_dev_flags* flags, const char* what);

#define ICV_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      icv_set_error(__VA_ARGS__);         \
      return 1;                           \
    }                                     \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ float bf16lo_to_f32(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16hi_to_f32(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

// round-to-nearest-even f32 -> bf16 through the hardware converter (v_cvt_pk_bf16_f32 on gfx950: one instruction per
// PAIR, where the integer add/shift idiom costs ~11)
typedef __bf16 icv_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const icv_bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned f32_to_bf16_bits(float f) { return pack_bf16x2(f, 0.f) & 0xffffu; }

// Element offset of output (m, n) for the GEMM epilogues: plain [M, N] (nsplit == N) or split planes
// [N / nsplit][M][nsplit].  A 64-bit n / nsplit per fragment cost more VALU than the GELU; the plain case needs no
// division at all and the split case only a 32-bit one (N < 2^31 is checked by the dispatchers).
__device__ __forceinline__ int64_t icv_out_offset(int64_t m, int64_t n, int64_t ldo, int64_t N, int64_t nsplit, int64_t split_stride) {
  if (nsplit == N) return m * ldo + n;
  const unsigned sub = (unsigned)n / (unsigned)nsplit;
  return (int64_t)sub * split_stride + m * ldo + (int64_t)((unsigned)n - sub * (unsigned)nsplit);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float gelu_tanh(float x) {
  // 0.5 x (1 + tanh(u)) == x * sigmoid(2u) == x / (1 + 2^(-2u log2 e)),  u = sqrt(2/pi) (x + 0.044715 x^3).
  // 7 VALU ops (2 transcendental): the constants are folded and the division is v_rcp_f32 (1 ulp; the result is
  // rounded to bf16).  x -> -inf: 2^(+big) = inf, rcp = 0, result -0 like the exact function.
  constexpr float C1 = 2.0f * 0.7978845608028654f * 1.4426950408889634f;   // 2 sqrt(2/pi) log2(e)
  constexpr float C3 = C1 * 0.044715f;
  const float w = x * fmaf(x * x, C3, C1);                                  // 2u log2(e)
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-w));
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }
