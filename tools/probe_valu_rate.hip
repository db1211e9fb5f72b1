// Hardware probe: issue cost (shader clocks per wave64 instruction) of the VALU instructions on the attention
// kernel's softmax fast path, measured with s_memtime around long runs of INDEPENDENT instructions from one wave per
// SIMD, and with two waves per SIMD (do two waves' VALU streams overlap?), plus the same VALU stream issued next to
// a stream of MFMAs from the other wave of the SIMD (does the matrix pipe hide VALU issue or do they add up?).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_valu_rate.hip -o tools/probe_valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP16(X) X X X X X X X X X X X X X X X X

// OP: 0 v_exp_f32, 1 v_add_f32, 2 v_pk_add_f32, 3 v_cvt_pk_bf16_f32, 4 v_pk_fma_f32, 5 v_fma_f32, 6 v_max_f32,
//     7 v_pk_mul_f32, 8 MFMA 32x32x16 bf16, 9 v_exp_f32 + MFMA interleaved 1:1 in ONE wave
template <int OP>
__device__ __forceinline__ void body(float (&r)[16], f32x2 (&q)[8], f32x16& acc, bf16x8 a, bf16x8 b) {
  if (OP == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
  } else if (OP == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[i]));
  } else if (OP == 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %0\n\tv_pk_add_f32 %0, %0, %0" : "+v"(q[i]));
  } else if (OP == 3) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(r[i]));
  } else if (OP == 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %0, %0, %0, %0" : "+v"(q[i]));
  } else if (OP == 5) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
  } else if (OP == 6) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %0" : "+v"(r[i]));
  } else if (OP == 7) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %0\n\tv_pk_mul_f32 %0, %0, %0" : "+v"(q[i]));
  } else if (OP == 8) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  } else if (OP == 9) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
    }
  } else if (OP == 11) {   // 16 INDEPENDENT v_mfma_f32_16x16x32_bf16 (4 passes each), accumulators aliased onto acc / r / q
    f32x4* a4 = reinterpret_cast<f32x4*>(&acc);
    f32x4* r4 = reinterpret_cast<f32x4*>(&r[0]);
    f32x4* q4 = reinterpret_cast<f32x4*>(&q[0]);
#pragma unroll
    for (int i = 0; i < 4; ++i) a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a4[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) r4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, r4[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) q4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q4[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, a4[i], 0, 0, 0);
  } else if (OP == 12) {   // 16 v_mfma_f32_32x32x16_bf16 on 3 rotating accumulators (independent neighbours)
    f32x16* r16 = reinterpret_cast<f32x16*>(&r[0]);
    f32x16* q16 = reinterpret_cast<f32x16*>(&q[0]);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i % 3 == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      else if (i % 3 == 1) *r16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, *r16, 0, 0, 0);
      else *q16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, *q16, 0, 0, 0);
    }
  } else if (OP == 10) {   // the fast path's VALU mix per MFMA: 1 exp + 1 add + 1/2 cvt, next to each MFMA
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %0" : "+v"(r[i]), "+v"(q[i & 7][0]));
      if (i & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(q[i & 7][1]) : "v"(r[i]), "v"(r[i - 1]));
    }
  }
}

// MODE 0: every wave runs OP.  MODE 1: waves 0-3 run OP, waves 4-7 (the second wave of each SIMD) run MFMAs.
template <int OP, int MODE>
__global__ void probe(long long* out, float* sink, int iters) {
  float r[16];
  f32x2 q[8];
  f32x16 acc;
  bf16x8 a, b;
  for (int i = 0; i < 16; ++i) { r[i] = -1.0f - 0.01f * (threadIdx.x + i); acc[i] = 0.f; }
  for (int i = 0; i < 8; ++i) { q[i][0] = 0.5f; q[i][1] = 0.25f; a[i] = (__bf16)0.001f; b[i] = (__bf16)0.002f; }
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (MODE == 1 && wave >= 4) {
    for (int it = 0; it < iters; ++it) body<8>(r, q, acc, a, b);
  } else {
    for (int it = 0; it < iters; ++it) body<OP>(r, q, acc, a, b);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += r[i] + acc[i];
  for (int i = 0; i < 8; ++i) s += q[i][0] + q[i][1];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int OP, int MODE>
static void run(const char* name, int threads, int per_iter) {
  const int blocks = 256, iters = 2000;
  long long* d_out; float* d_sink;
  hipMalloc(&d_out, blocks * 8 * sizeof(long long));
  hipMalloc(&d_sink, blocks * 512 * sizeof(float));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<OP, MODE>), dim3(blocks), dim3(threads), 0, 0, d_out, d_sink, iters);
  hipDeviceSynchronize();
  long long h[256 * 8];
  hipMemcpy(h, d_out, sizeof(long long) * blocks * (threads / 64), hipMemcpyDeviceToHost);
  const int wpb = threads / 64;
  double s_lo = 0, s_hi = 0; int n_lo = 0, n_hi = 0;
  for (int bl = 0; bl < blocks; ++bl)
    for (int w = 0; w < wpb; ++w) {
      if (w < 4) { s_lo += (double)h[bl * wpb + w]; ++n_lo; } else { s_hi += (double)h[bl * wpb + w]; ++n_hi; }
    }
  // s_memtime / readcyclecounter ticks at the constant 100 MHz reference on this part; the ratio to v_add_f32 is what matters
  printf("%-44s waves/SIMD %d  ticks per wave-instruction: waves0-3 %.4f", name, wpb / 4, s_lo / n_lo / ((double)iters * per_iter));
  if (n_hi) printf("   waves4-7 %.4f", s_hi / n_hi / ((double)iters * (MODE == 1 ? 16 : per_iter)));
  printf("\n");
  hipFree(d_out); hipFree(d_sink);
}

int main() {
  printf("ticks are readcyclecounter units; compare rows (v_add_f32 = 4 shader clocks per wave64 instruction)\n");
  run<1, 0>("v_add_f32", 256, 16);
  run<0, 0>("v_exp_f32", 256, 16);
  run<2, 0>("v_pk_add_f32", 256, 16);
  run<7, 0>("v_pk_mul_f32", 256, 16);
  run<4, 0>("v_pk_fma_f32", 256, 16);
  run<5, 0>("v_fma_f32", 256, 16);
  run<6, 0>("v_max_f32", 256, 16);
  run<3, 0>("v_cvt_pk_bf16_f32", 256, 16);
  run<8, 0>("v_mfma_f32_32x32x16_bf16 (dependent chain)", 256, 16);
  run<11, 0>("v_mfma_f32_16x16x32_bf16 x16 independent", 256, 16);
  run<12, 0>("v_mfma_f32_32x32x16_bf16 3 rotating accumulators", 256, 16);
  run<11, 0>("v_mfma_f32_16x16x32_bf16 x16 independent", 512, 16);
  run<1, 0>("v_add_f32", 512, 16);
  run<0, 0>("v_exp_f32", 512, 16);
  run<8, 0>("v_mfma_f32_32x32x16_bf16", 512, 16);
  run<9, 0>("1 wave: MFMA + v_exp interleaved (per pair)", 256, 16);
  run<10, 0>("1 wave: MFMA + exp + add + cvt/2 (per MFMA)", 256, 16);
  run<10, 0>("2 waves: MFMA + exp + add + cvt/2 (per MFMA)", 512, 16);
  run<0, 1>("v_exp_f32 beside an MFMA wave", 512, 16);
  run<1, 1>("v_add_f32 beside an MFMA wave", 512, 16);
  run<8, 1>("MFMA beside an MFMA wave", 512, 16);
  return 0;
}
