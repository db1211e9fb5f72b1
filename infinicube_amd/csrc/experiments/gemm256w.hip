// 256x256x64 bf16 MFMA GEMM, FOUR waves (2 x 2), wave tile 128 x 128, one wave per SIMD (icv_set_option("gemm256", 3)).
// Same LDS units, source-side swizzle, 2-stage x 4-unit ring and DMA issue order as gemm256.hip; what changes:
//   * a wave owns 128 x 128 of C (256 accumulator registers - hipcc places them in AGPRs), so a block reads
//     4 x 32 KB = 128 KB of LDS fragments per K-tile instead of 8 x 24 KB = 192 KB: 512 LDS cycles + the DMA writes
//     against 1024 MFMA cycles (gemm256.hip: 768 + writes against 1024 - it is LDS-bound);
//   * with one wave per SIMD nothing else covers a wave's LDS latency, so the fragments of phase p+1 are read into a
//     second register set while phase p's 32 MFMAs run (operands already in registers), one barrier per phase:
//       ph1(t): a0 x b0 | read b1(t)      | DMA B1(t+1)
//       ph2(t): a0 x b1 | read a1(t)      | DMA A1(t+1)
//       ph3(t): a1 x b1 | read a0(t+1)    | DMA A0(t+2)      (A0(t+1) was retired by the wait that ended ph2)
//       ph4(t): a1 x b0 | read b0(t+1), each k-step half right after its last MFMA | DMA B0(t+2)
//     every phase ends with vmcnt(12) = the three youngest units (4 DMA instructions per wave each) may still fly,
//     which retires exactly the unit(s) the NEXT phase's reads need; the barrier publishes them.
#include "icv_common.h"

namespace g256w {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int UNIT_BYTES = 128 * 128;         // 16 KiB
constexpr int STAGE_BYTES = 4 * UNIT_BYTES;   // A0 A1 B0 B1
constexpr int LDS_BYTES = 2 * STAGE_BYTES;    // 128 KiB
constexpr int U_A0 = 0, U_A1 = 1, U_B0 = 2, U_B1 = 3;

struct Params {
  const bf16_t* A; int64_t lda;
  const bf16_t* W; int64_t ldw;
  const float* bias;
  int64_t M, N, K;
  void* out; int64_t ldo; int64_t nsplit; int64_t split_stride;
  const float* resid; int64_t ldr;
  const float* gate;
  int tiles_m, tiles_n;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

#define G256_BARRIER()                      \
  do {                                      \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_s_barrier();           \
    asm volatile("" ::: "memory");          \
  } while (0)
#define GW_VMCNT12() asm volatile("s_waitcnt vmcnt(12)" ::: "memory")

__device__ __forceinline__ void dma_unit(const char* __restrict__ base, const unsigned (&off)[4],
                                         int64_t kbyte, char* lds_unit, int wave) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const char* src = base + (int64_t)off[q] + kbyte;
    char* dst = lds_unit + q * 4096 + wave * 1024;  // wave-uniform; HW adds lane*16
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm256w_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // ---- block -> tile (bijective XCD remap + grouped order), as gemm256.hip ----
  const int nwg = p.tiles_m * p.tiles_n;
  int wg;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  constexpr int GM = 4;
  const int group_size = GM * p.tiles_n;
  const int g = wg / group_size;
  const int first_m = g * GM;
  const int gm = min(p.tiles_m - first_m, GM);
  const int tm = first_m + (wg % group_size) % gm;
  const int tn = (wg % group_size) / gm;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  // ---- per-thread DMA source offsets (bytes, k = 0): 4 passes of 32 unit rows per unit ----
  unsigned offA[2][4], offB[2][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int u = q * 32 + (tid >> 3);            // unit row 0..127
    const int pc = tid & 7;
    const int c = pc ^ ((u >> 1) & 7);            // logical 16-B chunk held by physical chunk pc
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int64_t ra = m0 + (u >> 6) * 128 + h * 64 + (u & 63);   // unit A_h: rows {wr*128 + h*64 + [0,64)}
      ra = ra < p.M ? ra : p.M - 1;
      offA[h][q] = (unsigned)((ra * p.lda + c * 8) * 2);
      int64_t rb = n0 + (u >> 6) * 128 + h * 64 + (u & 63);   // unit B_h: cols {wc*128 + h*64 + [0,64)}
      rb = rb < p.N ? rb : p.N - 1;
      offB[h][q] = (unsigned)((rb * p.ldw + c * 8) * 2);
    }
  }
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Wb = reinterpret_cast<const char*>(p.W);
  const int nt = (int)(p.K / BK);
  auto kbyte = [&](int t) -> int64_t { return (int64_t)(t < nt ? t : nt - 1) * (BK * 2); };

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- fragment read addressing: lane (fr = row in a 16-row fragment, kq = 8-wide k chunk) ----
  const int fr = lane & 15, kq = lane >> 4;
  int a_off[2], b_off[2];   // byte offset within a unit per k-step (32 k each), minus the 16-row fragment term
  const int ar = wr * 64 + fr, br = wc * 64 + fr;   // (row + 16 i) keeps ((row >> 1) & 7)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_off[ks] = ar * 128 + (((ks * 4 + kq) ^ ((ar >> 1) & 7)) << 4);
    b_off[ks] = br * 128 + (((ks * 4 + kq) ^ ((br >> 1) & 7)) << 4);
  }
  constexpr int FROWS = 16 * 128;

#define GW_READ(DST_, UNIT_PTR_, OFF_)                                                           \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                               \
  _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                  \
    DST_[f][ks] = *reinterpret_cast<const bf16x8*>((UNIT_PTR_) + OFF_[ks] + f * FROWS);
// The 64 accumulators (256 registers) are pinned to AGPRs with "+a" constraints: left to itself hipcc keeps part of
// them in VGPRs and shuttles ~300 v_accvgpr_read/write/mov per K-tile.  Consecutive MFMAs use different accumulators
// (no back-to-back dependence inside the asm stream); operand waits (lgkmcnt) are the compiler's, it sees the inputs.
#define GW_READ_KS(DST_, UNIT_PTR_, OFF_, KS_)                                                   \
  _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                  \
    DST_[f][KS_] = *reinterpret_cast<const bf16x8*>((UNIT_PTR_) + OFF_[KS_] + f * FROWS);
#define GW_MFMA_KS(AF_, AH_, BF_, BH_, KS_)                                                      \
  {                                                                                              \
    __builtin_amdgcn_s_setprio(1);                                                               \
    _Pragma("unroll") for (int ks = (KS_); ks < (KS_) + 1; ++ks)                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                \
      asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                              \
          : "+a"(acc[(AH_) * 4 + i][(BH_) * 4 + j])                                              \
          : "v"(BF_[j][ks]), "v"(AF_[i][ks]));                                                   \
    __builtin_amdgcn_s_setprio(0);                                                               \
  }

  // ---- prologue: tile 0 complete + A0,B0 of tile 1 in flight; a0,b0 of tile 0 in registers ----
  dma_unit(Ab, offA[0], kbyte(0), smem + U_A0 * UNIT_BYTES, wave);
  dma_unit(Wb, offB[0], kbyte(0), smem + U_B0 * UNIT_BYTES, wave);
  dma_unit(Wb, offB[1], kbyte(0), smem + U_B1 * UNIT_BYTES, wave);
  dma_unit(Ab, offA[1], kbyte(0), smem + U_A1 * UNIT_BYTES, wave);
  dma_unit(Ab, offA[0], kbyte(1), smem + STAGE_BYTES + U_A0 * UNIT_BYTES, wave);
  dma_unit(Wb, offB[0], kbyte(1), smem + STAGE_BYTES + U_B0 * UNIT_BYTES, wave);
  GW_VMCNT12();      // A0(0), B0(0), B1(0) landed (three younger units in flight)
  G256_BARRIER();

  bf16x8 a0f[4][2], a1f[4][2], b0f[4][2], b1f[4][2];
  GW_READ(a0f, smem + U_A0 * UNIT_BYTES, a_off)
  GW_READ(b0f, smem + U_B0 * UNIT_BYTES, b_off)

  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * STAGE_BYTES;
    char* oth = smem + ((t + 1) & 1) * STAGE_BYTES;
    // Every phase: 16 MFMAs of k-step 0, the LDS reads for later phases, 16 MFMAs of k-step 1 — a read issued
    // in the middle of a phase has the second half's 128 MFMA cycles plus the barrier to land.
#define GW_SB() __builtin_amdgcn_sched_barrier(0)
    // ---------------- phase 1: a0 x b0 ; read b1(t) ----------------
    dma_unit(Wb, offB[1], kbyte(t + 1), oth + U_B1 * UNIT_BYTES, wave);
    GW_MFMA_KS(a0f, 0, b0f, 0, 0) GW_SB();
    GW_READ(b1f, cur + U_B1 * UNIT_BYTES, b_off) GW_SB();
    GW_MFMA_KS(a0f, 0, b0f, 0, 1) GW_SB();
    GW_VMCNT12();     // A1(t) landed (younger: A0(t+1), B0(t+1), B1(t+1))
    G256_BARRIER();
    // ---------------- phase 2: a0 x b1 ; read a1(t) ----------------
    dma_unit(Ab, offA[1], kbyte(t + 1), oth + U_A1 * UNIT_BYTES, wave);
    GW_MFMA_KS(a0f, 0, b1f, 1, 0) GW_SB();
    GW_READ(a1f, cur + U_A1 * UNIT_BYTES, a_off) GW_SB();
    GW_MFMA_KS(a0f, 0, b1f, 1, 1) GW_SB();
    GW_VMCNT12();     // A0(t+1) landed
    G256_BARRIER();
    // ---------------- phase 3: a1 x b1 ; read a0(t+1) once its k-step-0 half is free ----------------
    dma_unit(Ab, offA[0], kbyte(t + 2), cur + U_A0 * UNIT_BYTES, wave);
    GW_MFMA_KS(a1f, 1, b1f, 1, 0) GW_SB();
    GW_READ(a0f, oth + U_A0 * UNIT_BYTES, a_off) GW_SB();
    GW_MFMA_KS(a1f, 1, b1f, 1, 1) GW_SB();
    GW_VMCNT12();     // B0(t+1) landed
    G256_BARRIER();
    // ---------------- phase 4: a1 x b0 ; b0(t+1) refills each k-step half right after its last use ----------------
    dma_unit(Wb, offB[0], kbyte(t + 2), cur + U_B0 * UNIT_BYTES, wave);
    GW_MFMA_KS(a1f, 1, b0f, 0, 0) GW_SB();
    GW_READ_KS(b0f, oth + U_B0 * UNIT_BYTES, b_off, 0) GW_SB();
    GW_MFMA_KS(a1f, 1, b0f, 0, 1) GW_SB();
    GW_READ_KS(b0f, oth + U_B0 * UNIT_BYTES, b_off, 1) GW_SB();
    GW_VMCNT12();     // B1(t+1) landed
    G256_BARRIER();
  }
#undef GW_SB
#undef GW_MFMA_KS
#undef GW_READ_KS
#undef GW_READ
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain tail DMA before the LDS is released
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last asm MFMAs must have written their AGPRs before the epilogue reads them

  // ---- epilogue: a lane owns ONE row m and runs of 4 consecutive n (swapped MFMA operands) ----
#define G256_EMIT(M_, N_, V0_, V1_, V2_, V3_)                                                          \
  {                                                                                                    \
    const int64_t m = (M_), n = (N_);                                                                  \
    if (m < p.M && n < p.N) {                                                                          \
      float v0 = (V0_), v1 = (V1_), v2 = (V2_), v3 = (V3_);                                            \
      if (p.bias) {                                                                                    \
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);                                 \
        v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;                                                    \
      }                                                                                                \
      const int64_t off = icv_out_offset(m, n, p.ldo, p.N, p.nsplit, p.split_stride);                  \
      if (EPI == ICV_EPI_BF16 || EPI == ICV_EPI_GELU_BF16) {                                           \
        if (EPI == ICV_EPI_GELU_BF16) {                                                                \
          v0 = gelu_tanh(v0); v1 = gelu_tanh(v1); v2 = gelu_tanh(v2); v3 = gelu_tanh(v3);              \
        }                                                                                              \
        *reinterpret_cast<uint2*>((bf16_t*)p.out + off) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)); \
      } else if (EPI == ICV_EPI_RESID_F32) {                                                           \
        const float4 r = *reinterpret_cast<const float4*>(p.resid + m * p.ldr + n);                    \
        float4 o;                                                                                      \
        if (p.gate) {                                                                                  \
          const float4 gt = *reinterpret_cast<const float4*>(p.gate + n);                              \
          o = make_float4(r.x + gt.x * v0, r.y + gt.y * v1, r.z + gt.z * v2, r.w + gt.w * v3);         \
        } else {                                                                                       \
          o = make_float4(r.x + v0, r.y + v1, r.z + v2, r.w + v3);                                     \
        }                                                                                              \
        *reinterpret_cast<float4*>((float*)p.out + off) = o;                                           \
      } else {                                                                                         \
        *reinterpret_cast<float4*>((float*)p.out + off) = make_float4(v0, v1, v2, v3);                 \
      }                                                                                                \
    }                                                                                                  \
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      G256_EMIT(m0 + wr * 128 + i * 16 + fr, n0 + wc * 128 + j * 16 + kq * 4, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3])
#undef G256_EMIT
}


template <int EPI>
int launch(const Params& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm256w_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      icv_set_error("gemm256w: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  hipLaunchKernelGGL((gemm256w_kernel<EPI>), dim3((unsigned)nwg), dim3(256), LDS_BYTES, st, p);
  return icv_check_launch("icv_gemm_bf16(256w)");
}

}  // namespace g256w

// Called by icv_gemm_bf16 (gemm.hip) when the "gemm256" option is 3.
int icv_gemm256w_dispatch(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                          int64_t M, int64_t N, int64_t K, int epilogue, void* out, int64_t ldo,
                          int64_t nsplit, int64_t split_stride, const float* resid, int64_t ldr,
                          const float* gate, hipStream_t st) {
  g256w::Params p;
  p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.out = out; p.ldo = ldo; p.nsplit = nsplit; p.split_stride = split_stride;
  p.resid = resid; p.ldr = ldr; p.gate = gate;
  p.tiles_m = (int)((M + g256w::BM - 1) / g256w::BM);
  p.tiles_n = (int)((N + g256w::BN - 1) / g256w::BN);
  switch (epilogue) {
    case ICV_EPI_BF16: return g256w::launch<ICV_EPI_BF16>(p, st);
    case ICV_EPI_GELU_BF16: return g256w::launch<ICV_EPI_GELU_BF16>(p, st);
    case ICV_EPI_RESID_F32: return g256w::launch<ICV_EPI_RESID_F32>(p, st);
    case ICV_EPI_F32: return g256w::launch<ICV_EPI_F32>(p, st);
  }
  icv_set_error("icv_gemm_bf16: unknown epilogue %d", epilogue);
  return 1;
}
