// 256x256x64 bf16 MFMA GEMM, FOUR waves (2 x 2), wave tile 128 x 128, one wave per SIMD, ONE barrier per K-tile
// (icv_set_option("gemm256", 4)).  gemm256w.hip's LDS units, swizzle, DMA lane mapping, AGPR-pinned accumulators and
// epilogue; what changes is the main loop:
//   * a K-tile is staged WHOLE (4 units = 64 KiB) into one of two stages; tile t+2 is requested right after the barrier
//     in the middle of tile t, i.e. a full tile time (128 MFMAs per wave) before it is needed;
//   * the loop body is two k-step halves of 64 MFMAs; the 16 fragment reads of the NEXT half (k-step 1 of this tile, then
//     k-step 0 of the next tile from the other stage) are spread one per 4 MFMAs, and so are the 16 DMA instructions, so a
//     single in-order wave keeps the matrix pipe fed: per MFMA group [4 MFMA | 1 ds_read_b128 | (1 global_load_lds)];
//   * the only barrier sits between the halves: by then every wave has read all of stage s (its k-step-1 fragments were
//     fetched during the first half) so stage s may be refilled, and the wave's own share of tile t+1 has landed
//     (vmcnt(0) - nothing younger is in flight at that point), which the barrier publishes.
// LDS traffic per K-tile: 4 waves x 32 KiB of fragments + 64 KiB of DMA writes against 2048 MFMA cycles per SIMD
// (gemm256.hip: 8 x 24 KiB + 64 KiB against the same).
#include "icv_common.h"

namespace g256x {

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
  int ablate;   // timing experiments: 1 = no DMA in the loop, 2 = no mid-tile wait + barrier (results wrong)
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

#define G256_BARRIER()                      \
  do {                                      \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_s_barrier();           \
    asm volatile("" ::: "memory");          \
  } while (0)

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
__global__ __launch_bounds__(256) void gemm256x_kernel(Params p) {
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

  // One accumulator row i (8 MFMAs against b[0..7]) in two groups of 4; AGPR-pinned accumulators ("+a"): left to itself
  // hipcc keeps part of the 256 in VGPRs and shuttles them.  Consecutive MFMAs never share an accumulator.
#define GX_MFMA4(AF_, BF_, I_, J0_)                                                              \
  _Pragma("unroll") for (int j = (J0_); j < (J0_) + 4; ++j)                                      \
    asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                                \
        : "+a"(acc[(I_)][j])                                                                     \
        : "v"(BF_[j]), "v"(AF_[(I_)]));
  // fragment f (0..7) of a k-step: units X0 (f < 4) and X1, 16-row block f & 3
#define GX_RD(STAGE_, U0_, OFF_, KS_, F_) \
  *reinterpret_cast<const bf16x8*>((STAGE_) + ((U0_) + ((F_) >> 2)) * UNIT_BYTES + OFF_[KS_] + ((F_) & 3) * FROWS)
#define GX_SB() __builtin_amdgcn_sched_barrier(0)

  // ---- prologue: tiles 0 and 1 requested; tile 0 landed and published; k-step 0 fragments of tile 0 in registers ----
  dma_unit(Ab, offA[0], kbyte(0), smem + U_A0 * UNIT_BYTES, wave);
  dma_unit(Wb, offB[0], kbyte(0), smem + U_B0 * UNIT_BYTES, wave);
  dma_unit(Ab, offA[1], kbyte(0), smem + U_A1 * UNIT_BYTES, wave);
  dma_unit(Wb, offB[1], kbyte(0), smem + U_B1 * UNIT_BYTES, wave);
  dma_unit(Ab, offA[0], kbyte(1), smem + STAGE_BYTES + U_A0 * UNIT_BYTES, wave);
  dma_unit(Wb, offB[0], kbyte(1), smem + STAGE_BYTES + U_B0 * UNIT_BYTES, wave);
  dma_unit(Ab, offA[1], kbyte(1), smem + STAGE_BYTES + U_A1 * UNIT_BYTES, wave);
  dma_unit(Wb, offB[1], kbyte(1), smem + STAGE_BYTES + U_B1 * UNIT_BYTES, wave);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // tile 0's 16 instructions have landed (tile 1's 16 may fly)
  G256_BARRIER();

  bf16x8 af0[8], bf0[8], af1[8], bf1[8];   // fragments of k-step 0 / k-step 1
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    bf0[f] = GX_RD(smem, U_B0, b_off, 0, f);
    af0[f] = GX_RD(smem, U_A0, a_off, 0, f);
  }

  for (int t = 0; t < nt; ++t) {
    const char* cur = smem + (t & 1) * STAGE_BYTES;
    const char* oth = smem + ((t + 1) & 1) * STAGE_BYTES;
    char* curw = smem + (t & 1) * STAGE_BYTES;
    // ---------------- first half: k-step 0 MFMAs | fetch k-step 1 fragments of this tile (b first: row 0 needs all of them) ----
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      GX_MFMA4(af0, bf0, g >> 1, (g & 1) * 4)
      if (g < 8) bf1[g] = GX_RD(cur, U_B0, b_off, 1, g);
      else af1[g - 8] = GX_RD(cur, U_A0, a_off, 1, g - 8);
      GX_SB();
    }
    if (!(p.ablate & 2)) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // my share of tile t+1 landed; my reads of stage s retired
      G256_BARRIER();
    }
    GX_SB();
    // ---------------- second half: k-step 1 MFMAs | fetch k-step 0 fragments of tile t+1 | request tile t+2 into stage s ----
    const int64_t kb2 = kbyte(t + 2);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      GX_MFMA4(af1, bf1, g >> 1, (g & 1) * 4)
      if (g < 8) bf0[g] = GX_RD(oth, U_B0, b_off, 0, g);
      else af0[g - 8] = GX_RD(oth, U_A0, a_off, 0, g - 8);
      {   // DMA instruction g of this wave: unit g >> 2 (A0, A1, B0, B1), pass g & 3
        const int u = g >> 2, q = g & 3;
        const char* base = u < 2 ? Ab : Wb;
        const unsigned off = u == 0 ? offA[0][q] : u == 1 ? offA[1][q] : u == 2 ? offB[0][q] : offB[1][q];
        const char* srcp = base + (int64_t)off + kb2;
        char* dst = curw + u * UNIT_BYTES + q * 4096 + wave * 1024;
        if (!(p.ablate & 1)) __builtin_amdgcn_global_load_lds((gbl_void*)srcp, (lds_void*)dst, 16, 0, 0);
      }
      GX_SB();
    }
  }
#undef GX_SB
#undef GX_RD
#undef GX_MFMA4
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
    hipError_t e = hipFuncSetAttribute((const void*)gemm256x_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      icv_set_error("gemm256x: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  hipLaunchKernelGGL((gemm256x_kernel<EPI>), dim3((unsigned)nwg), dim3(256), LDS_BYTES, st, p);
  return icv_check_launch("icv_gemm_bf16(256x)");
}

}  // namespace g256x

// Called by icv_gemm_bf16 (gemm.hip) when the "gemm256" option is 4.
int icv_gemm256x_dispatch(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                          int64_t M, int64_t N, int64_t K, int epilogue, void* out, int64_t ldo,
                          int64_t nsplit, int64_t split_stride, const float* resid, int64_t ldr,
                          const float* gate, hipStream_t st) {
  g256x::Params p;
  p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.out = out; p.ldo = ldo; p.nsplit = nsplit; p.split_stride = split_stride;
  p.resid = resid; p.ldr = ldr; p.gate = gate;
  p.tiles_m = (int)((M + g256x::BM - 1) / g256x::BM);
  p.tiles_n = (int)((N + g256x::BN - 1) / g256x::BN);
  p.ablate = icv_get_option_int("gemm256x_ablate", 0);
  switch (epilogue) {
    case ICV_EPI_BF16: return g256x::launch<ICV_EPI_BF16>(p, st);
    case ICV_EPI_GELU_BF16: return g256x::launch<ICV_EPI_GELU_BF16>(p, st);
    case ICV_EPI_RESID_F32: return g256x::launch<ICV_EPI_RESID_F32>(p, st);
    case ICV_EPI_F32: return g256x::launch<ICV_EPI_F32>(p, st);
  }
  icv_set_error("icv_gemm_bf16: unknown epilogue %d", epilogue);
  return 1;
}
