// Hardware probe: what MFMA-pipe occupancy can the attention inner loop's INSTRUCTION MIX reach on gfx950, as a function
// of how the instructions are ordered inside a wave?  No attention is computed: the kernel issues, per 64-key tile and per
// wave, exactly the fast path's 32 MFMA 32x32x16 + 32 v_exp_f32 + 16 v_pk_add_f32 + 16 v_cvt_pk_bf16_f32 +
// 16 ds_read_b128 + 32 ds_read_b64_tr_b16 (+ one s_barrier), 8 waves per block, one block per CU, with the LDS results
// feeding the MFMAs, in three orders:
//   PHASED   (attn7's order)  16 x [ds_read_b128, MFMA] ; 64 VALU ; 16 x [2 x ds_read_tr, MFMA]
//   EVEN     every MFMA followed by its share: 1 exp + (pk_add | cvt) + the LDS reads of the NEXT MFMA
//   MFMA_ONLY / NO_LDS / NO_VALU ablations of EVEN
// Output: clocks per tile per SIMD (2 waves) and MFMA occupancy = 2 x 32 x 32 clk / that.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_attn_mix.hip -o tools/probe_attn_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x4 lds_tr(const char* p) {
  return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
}
__device__ __forceinline__ bf16x8 cat(bf16x4 a, bf16x4 b) {
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// MODE 0 PHASED, 1 EVEN, 2 EVEN without VALU, 3 EVEN without LDS reads, 4 MFMA only, 5 EVEN with 2 exps per MFMA on
// half of the MFMAs (attn9's shape: all VALU next to the QK^T MFMAs, PV bare), 6 = MFMA only on RANDOM operands (hashed
// bf16 bit patterns, zero-mean): the same instruction stream as 4, only the datapath toggling differs
template <int MODE>
__global__ __launch_bounds__(512) void mix(long long* out, float* sink, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32768; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x16 st[2], ot[4];
  bf16x8 qf[8], pf[2];
  float ex[32];
  f32x2 ps = {0.f, 0.f};
  for (int i = 0; i < 16; ++i) { st[0][i] = st[1][i] = -1.f; ot[0][i] = ot[1][i] = ot[2][i] = ot[3][i] = 0.f; }
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) qf[i][e] = (__bf16)(0.01f * (lane + i));
  for (int e = 0; e < 8; ++e) pf[0][e] = pf[1][e] = (__bf16)0.5f;
  for (int i = 0; i < 32; ++i) ex[i] = -0.5f - 0.001f * i;
  if (MODE >= 6) {
    unsigned h = 0x9E3779B9u * (unsigned)(blockIdx.x * 512 + tid + 1);
    for (int i = 0; i < 8; ++i)
      for (int e = 0; e < 8; ++e) {
        h = h * 1664525u + 1013904223u;
        // sign + 7 mantissa bits random, exponent in [2^-2, 2^1): a zero-mean operand with every mantissa bit toggling
        const unsigned short bits = (unsigned short)(((h >> 16) & 0x807Fu) | ((125u + ((h >> 8) & 3u)) << 7));
        qf[i][e] = __builtin_bit_cast(__bf16, bits);
      }
    for (int e = 0; e < 8; ++e) pf[0][e] = qf[3][e], pf[1][e] = qf[5][e];
  }
  const char* kbase = smem + (lane & 31) * 256;
  int koff[8];                                   // the kernel's XOR swizzle: conflict-free ds_read_b128 over 256-byte rows
  for (int ds = 0; ds < 8; ++ds) koff[ds] = ((ds * 2 + (lane >> 5)) ^ (lane & 15)) << 4;
  const char* vbase = smem + 16384 + (lane >> 4) * 512 + (lane & 15) * 8;
  const long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < tiles; ++t) {
    const char* ks = kbase + (t & 3) * 32768;
    const char* vs = vbase + (t & 3) * 32768;
    if (MODE == 0) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ds = 0; ds < 8; ++ds) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + kb * 8192 + koff[ds]);
          st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], st[kb], 0, 0, 0);
        }
      FENCE();
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        f32x2 pv;
        pv[0] = __builtin_amdgcn_exp2f(ex[i] + st[i >> 4][i & 15] * 1e-30f);
        pv[1] = __builtin_amdgcn_exp2f(ex[i + 1] + st[i >> 4][(i + 1) & 15] * 1e-30f);
        ps += pv;
        pf[(i >> 3) & 1][i & 7] = (__bf16)pv[0];
        pf[(i >> 3) & 1][(i & 7) + 1] = (__bf16)pv[1];
      }
      FENCE();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int d0 = 0; d0 < 4; ++d0) {
          const bf16x4 va = lds_tr(vs + kk * 4096 + d0 * 64);
          const bf16x4 vb = lds_tr(vs + kk * 4096 + d0 * 64 + 2048);
          ot[d0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat(va, vb), pf[kk & 1], ot[d0], 0, 0, 0);
        }
    } else {
      // 32 slots; slot s: MFMA s (even slots' 16 = "QK^T" with a b128 K fragment, odd = "PV" with two tr reads), then this
      // slot's VALU share, then the LDS reads of slot s+1
      bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + koff[0]);
      bf16x4 va = lds_tr(vs), vb = lds_tr(vs + 2048);
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const int h = s >> 1;
        if ((s & 1) == 0) {
          st[h >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MODE == 7 ? qf[0] : MODE == 8 ? qf[h & 7] : MODE == 3 || MODE == 4 || MODE == 6 ? qf[(h + 1) & 7] : kf, MODE >= 7 ? qf[1] : qf[h & 7], st[h >> 3], 0, 0, 0);
          if (MODE != 3 && MODE != 4 && MODE < 6 && s + 2 < 32) kf = *reinterpret_cast<const bf16x8*>(ks + ((h + 1) >> 3) * 8192 + koff[(h + 1) & 7]);
        } else {
          ot[h & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MODE == 7 ? qf[0] : MODE == 3 || MODE == 4 || MODE == 6 || MODE == 8 ? qf[h & 7] : cat(va, vb), MODE >= 7 ? qf[1] : pf[(h >> 2) & 1], ot[h & 3], 0, 0, 0);
          if (MODE != 3 && MODE != 4 && MODE < 6 && s + 2 < 32) {
            va = lds_tr(vs + ((h + 1) >> 2) * 4096 + ((h + 1) & 3) * 64);
            vb = lds_tr(vs + ((h + 1) >> 2) * 4096 + ((h + 1) & 3) * 64 + 2048);
          }
        }
        if (MODE == 1 || MODE == 3) {
          const float pv = __builtin_amdgcn_exp2f(ex[s]);
          ex[s] = pv * 1e-30f - 0.5f;                       // keeps the exp live without a long dependency chain
          if (s & 1) {
            f32x2 two = {pv, ex[s - 1]};
            ps += two;
          } else {
            pf[(s >> 3) & 1][(s >> 1) & 7] = (__bf16)pv;     // one v_cvt_pk_bf16_f32 per two exps
            pf[(s >> 3) & 1][((s >> 1) & 7) ^ 1] = (__bf16)ex[(s + 31) & 31];
          }
        }
        if (MODE == 5 && (s & 1) == 0) {
          const float p0 = __builtin_amdgcn_exp2f(ex[s]), p1 = __builtin_amdgcn_exp2f(ex[s + 1]);
          ex[s] = p0 * 1e-30f - 0.5f; ex[s + 1] = p1 * 1e-30f - 0.5f;
          f32x2 two = {p0, p1};
          ps += two;
          pf[(s >> 3) & 1][(s >> 1) & 7] = (__bf16)p0;
          pf[(s >> 3) & 1][((s >> 1) & 7) ^ 1] = (__bf16)p1;
        }
        FENCE();
      }
    }
    __builtin_amdgcn_s_barrier();
    FENCE();
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = ps[0] + ps[1];
  for (int i = 0; i < 16; ++i) s += st[0][i] + st[1][i] + ot[0][i] + ot[1][i] + ot[2][i] + ot[3][i];
  for (int i = 0; i < 32; ++i) s += ex[i];
  sink[blockIdx.x * 512 + tid] = s + (float)pf[0][0] + (float)pf[1][3];
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

// LDS port alone: the attention tile's 16 ds_read_b128 + 32 ds_read_b64_tr_b16 per wave (32 KiB per wave, 256 KiB per
// block and tile), or 32 ds_read_b128 per wave (the same bytes), nothing else; results folded into a sink with v_xor
template <int TR>
__global__ __launch_bounds__(512) void ldsonly(long long* out, float* sink, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32768; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  const char* kbase = smem + (lane & 31) * 256;
  int koff[8];
  for (int ds = 0; ds < 8; ++ds) koff[ds] = ((ds * 2 + (lane >> 5)) ^ (lane & 15)) << 4;
  const char* vbase = smem + 16384 + (lane >> 4) * 512 + (lane & 15) * 8;
  unsigned acc0 = 0, acc1 = 0;
  // inline asm so that the compiler can neither narrow nor drop the loads; destination registers are scratch
  unsigned kaddr[8];
  for (int ds = 0; ds < 8; ++ds) kaddr[ds] = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)(kbase + koff[ds]));
  const unsigned vaddr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)vbase);
  const long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < tiles; ++t) {
    const unsigned so = (unsigned)((t & 3) * 32768);
    u32x4 d;
    u32x2 e;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(kaddr[i] + so));
      asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(d) : "v"(kaddr[i] + so));
    }
    if (TR) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(e) : "v"(vaddr + so + (unsigned)((i >> 1) * 4096 + (i & 1) * 64)));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(e) : "v"(vaddr + so + (unsigned)((i >> 1) * 4096 + (i & 1) * 64)));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:128" : "=v"(e) : "v"(vaddr + so + (unsigned)((i >> 1) * 4096 + (i & 1) * 64)));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2176" : "=v"(e) : "v"(vaddr + so + (unsigned)((i >> 1) * 4096 + (i & 1) * 64)));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(d) : "v"(kaddr[i] + so));
        asm volatile("ds_read_b128 %0, %1 offset:24576" : "=v"(d) : "v"(kaddr[i] + so));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    acc0 ^= d[0];
    acc1 ^= e[0];
  }
  const long long t1 = __builtin_readcyclecounter();
  sink[blockIdx.x * 512 + tid] = (float)(acc0 ^ acc1);
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int TR>
static void run_lds(const char* name, int tiles) {
  const int blocks = 256;
  long long* d_out; float* d_sink;
  (void)hipMalloc(&d_out, blocks * 8 * sizeof(long long));
  (void)hipMalloc(&d_sink, blocks * 512 * sizeof(float));
  (void)hipFuncSetAttribute((const void*)ldsonly<TR>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((ldsonly<TR>), dim3(blocks), dim3(512), 131072, 0, d_out, d_sink, tiles);
  (void)hipDeviceSynchronize();
  long long* hh = (long long*)malloc(blocks * 8 * sizeof(long long));
  (void)hipMemcpy(hh, d_out, blocks * 8 * sizeof(long long), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < blocks * 8; ++i) s += (double)hh[i];
  const double ticks = s / (blocks * 8) / tiles;
  printf("%-52s %8.1f counter ticks / tile   = %.1f LDS bytes per clock per CU (256 KiB per tile)\n", name, ticks, 262144.0 / ticks);
  free(hh); (void)hipFree(d_out); (void)hipFree(d_sink);
}

typedef int i32x8 __attribute__((ext_vector_type(8)));

// MFMA only, e4m3: 16 x v_mfma_scale_f32_32x32x64_f8f6f4 (unit scales) per wave per "tile" = the same 32768 x 32 flops as the
// 32 bf16 MFMAs above; RANDOM = hashed e4m3 bit patterns (sign + 3 mantissa bits random, exponent in a 4-value window)
template <int RANDOM>
__global__ __launch_bounds__(512) void mix8(long long* out, float* sink, int tiles) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x16 st[2], ot[2];
  i32x8 qf[8];
  for (int i = 0; i < 16; ++i) st[0][i] = st[1][i] = ot[0][i] = ot[1][i] = 0.f;
  unsigned h = 0x9E3779B9u * (unsigned)(blockIdx.x * 512 + tid + 1);
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 8; ++e) {
      unsigned w = 0;
      for (int b = 0; b < 4; ++b) {
        h = h * 1664525u + 1013904223u;
        const unsigned byte = RANDOM ? (((h >> 16) & 0x87u) | ((6u + ((h >> 8) & 3u)) << 3)) : 0x30u;   // e4m3: s eeee mmm
        w |= byte << (8 * b);
      }
      qf[i][e] = (int)w;
    }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < tiles; ++t) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s & 1) ot[(s >> 1) & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qf[s & 7], qf[(s + 3) & 7], ot[(s >> 1) & 1], 0, 0, 0, 127, 0, 127);
      else st[(s >> 1) & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qf[(s + 1) & 7], qf[s & 7], st[(s >> 1) & 1], 0, 0, 0, 127, 0, 127);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
  for (int i = 0; i < 16; ++i) sum += st[0][i] + st[1][i] + ot[0][i] + ot[1][i];
  sink[blockIdx.x * 512 + tid] = sum;
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int RANDOM>
static void run8(const char* name, int tiles) {
  const int blocks = 256;
  long long* d_out; float* d_sink;
  (void)hipMalloc(&d_out, blocks * 8 * sizeof(long long));
  (void)hipMalloc(&d_sink, blocks * 512 * sizeof(float));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((mix8<RANDOM>), dim3(blocks), dim3(512), 0, 0, d_out, d_sink, tiles);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((mix8<RANDOM>), dim3(blocks), dim3(512), 0, 0, d_out, d_sink, tiles);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long* hh = (long long*)malloc(blocks * 8 * sizeof(long long));
  (void)hipMemcpy(hh, d_out, blocks * 8 * sizeof(long long), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < blocks * 8; ++i) s += (double)hh[i];
  const double tf = (double)blocks * 8 * tiles * 16 * (32.0 * 32 * 64 * 2) / (ms * 1e-3) / 1e12;
  printf("%-52s %8.1f counter ticks / tile   %7.1f TF/s of MFMA work   (%.3f ms)\n", name, s / (blocks * 8) / tiles, tf, ms);
  free(hh); (void)hipFree(d_out); (void)hipFree(d_sink);
}

template <int MODE>
static void run(const char* name, int tiles = 2000) {
  const int blocks = 256;
  long long* d_out; float* d_sink;
  (void)hipMalloc(&d_out, blocks * 8 * sizeof(long long));
  (void)hipMalloc(&d_sink, blocks * 512 * sizeof(float));
  (void)hipFuncSetAttribute((const void*)mix<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((mix<MODE>), dim3(blocks), dim3(512), 131072, 0, d_out, d_sink, tiles);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((mix<MODE>), dim3(blocks), dim3(512), 131072, 0, d_out, d_sink, tiles);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long* h = (long long*)malloc(blocks * 8 * sizeof(long long));
  (void)hipMemcpy(h, d_out, blocks * 8 * sizeof(long long), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < blocks * 8; ++i) s += (double)h[i];
  const double ticks = s / (blocks * 8) / tiles;
  // wall-clock view: MFMA flops per second if this were the attention kernel (32 MFMA x 32768 flop per wave per tile)
  const double tf = (double)blocks * 8 * tiles * 32 * 32768.0 / (ms * 1e-3) / 1e12;
  printf("%-52s %8.1f counter ticks / tile   %7.1f TF/s of MFMA work   (%.3f ms)\n", name, ticks, tf, ms);
  free(h); (void)hipFree(d_out); (void)hipFree(d_sink);
}

int main() {
  run<4>("MFMA only (32 per wave per tile), benign data");
  run<6>("MFMA only, random bf16 operands, A and B change every MFMA");
  run<8>("MFMA only, random operands, A changes, B fixed");
  run<7>("MFMA only, random operands, A and B fixed");
  run<2>("MFMA + LDS fragment reads, even");
  run<3>("MFMA + softmax VALU, even");
  run<1>("MFMA + VALU + LDS, EVEN interleave");
  run<5>("MFMA + VALU + LDS, VALU on the QK^T half only");
  run<0>("MFMA + VALU + LDS, PHASED (attn7 order)");
  // steady state: ~10x longer launches (the power manager reacts within milliseconds)
  run<4>("long: MFMA only, benign data", 20000);
  run<6>("long: MFMA only, random, A and B change", 20000);
  run<7>("long: MFMA only, random, A and B fixed", 20000);
  run<0>("long: PHASED mix, benign data", 20000);
  run_lds<1>("LDS only: 16 ds_read_b128 + 32 ds_read_b64_tr_b16 / wave", 2000);
  run_lds<0>("LDS only: 32 ds_read_b128 / wave", 2000);
  run8<0>("long: e4m3 MFMA only (32x32x64), benign data", 20000);
  run8<1>("long: e4m3 MFMA only (32x32x64), random operands", 20000);
  return 0;
}
