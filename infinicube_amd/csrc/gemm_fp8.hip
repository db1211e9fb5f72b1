// fp8 (OCP e4m3) GEMM for BASELINE.json config #5 ("fp8 MFMA weights ... CDNA4 fp8 path"):
//   out = epilogue( (Aq[M,K] . Wq[N,K]^T) * a_scale[m] * w_scale[n] + bias )
// Aq/Wq are e4m3 bytes with one f32 scale per row (per token / per output channel); accumulation is f32.
//
// Only the K = 128 MFMA (v_mfma_f32_16x16x128_f8f6f4, here with unit block scales = its unscaled form)
// runs at twice the bf16 rate on gfx950 — the K = 32 fp8 MFMAs run at the bf16 rate — so this kernel
// is gemm256.hip's schedule re-cut for it: the LDS image of a K-tile is byte-identical (128 rows x 128 B
// per unit, same source-side XOR swizzle, same 4-phase counted-vmcnt DMA pipeline, same staggered wave
// groups), but the 128 B of a row now hold 128 k values and feed ONE MFMA per 16x16 fragment pair
// instead of two: lane (fr = lane & 15, g = lane >> 4) owns bytes [32g, 32g + 32) of its row
// (operand layout probed on hardware: tools/probe_f8.hip).  Same MFMA time per tile, twice the flops.
// Epilogues are those of icv_gemm_bf16 with the two row scales applied to the accumulator first.
#include "icv_common.h"

namespace gf8 {

constexpr int BM = 256, BN = 256, BKB = 128;  // BKB = bytes (= fp8 elements) of K per tile
constexpr int UNIT_BYTES = 128 * 128;
constexpr int STAGE_BYTES = 4 * UNIT_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;
constexpr int U_A0 = 0, U_A1 = 1, U_B0 = 2, U_B1 = 3;

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct Params {
  const char* A; int64_t lda;      // bytes
  const char* W; int64_t ldw;
  const float* a_scale; const float* w_scale;
  const float* bias;
  int64_t M, N, K;
  void* out; int64_t ldo; int64_t nsplit; int64_t split_stride;
  const float* resid; int64_t ldr;
  const float* gate;
  int tiles_m, tiles_n;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

#define GF8_BARRIER()                       \
  do {                                      \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_s_barrier();           \
    asm volatile("" ::: "memory");          \
  } while (0)
#define GF8_VMCNT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")

__device__ __forceinline__ void dma_unit(const char* __restrict__ base, const unsigned (&off)[2],
                                         int64_t kbyte, char* lds_unit, int wave) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const char* src = base + (int64_t)off[q] + kbyte;
    char* dst = lds_unit + q * 8192 + wave * 1024;  // wave-uniform; HW adds lane*16
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
  }
}

__device__ __forceinline__ i32x8 read_frag(const char* p0, const char* p1) {
  const i32x4 lo = *reinterpret_cast<const i32x4*>(p0);
  const i32x4 hi = *reinterpret_cast<const i32x4*>(p1);
  return (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// SCH bit 0: two phases of 16 MFMAs per K-tile (gemm256.hip's round-2 schedule: half the barriers); bit 1: the RESID
// epilogue issues the 16 residual loads of a half-tile before consuming any.  A/B switch "gemm_fp8_sched" (default 3).
template <int EPI, int SCH>
__global__ __launch_bounds__(512) void gemm_fp8_kernel(Params p) {
  constexpr bool TWO_PHASE = SCH & 1, BATCH_EPI = SCH & 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

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

  // ---- per-thread DMA source offsets (bytes, k = 0), 2 passes per unit ----
  unsigned offA[2][2], offB[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int u = q * 64 + (tid >> 3);            // unit row 0..127
    const int pc = tid & 7;
    const int c = pc ^ ((u >> 1) & 7);            // logical 16-B chunk held by physical chunk pc
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int64_t ra = m0 + (u >> 6) * 128 + h * 64 + (u & 63);
      ra = ra < p.M ? ra : p.M - 1;
      offA[h][q] = (unsigned)(ra * p.lda + c * 16);
      int64_t rb = n0 + (u >> 5) * 64 + h * 32 + (u & 31);
      rb = rb < p.N ? rb : p.N - 1;
      offB[h][q] = (unsigned)(rb * p.ldw + c * 16);
    }
  }
  const int nt = (int)(p.K / BKB);
  auto kbyte = [&](int t) -> int64_t { return (int64_t)(t < nt ? t : nt - 1) * BKB; };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- fragment addressing: lane (fr, kg) reads logical chunks kg and 4 + kg of its row ----
  // (round 6: it used to be chunks 2kg, 2kg + 1 - the contiguous 32 k values - and half of the kernel's LDS cycles were bank conflicts,
  // profiles/r06/rocprofv3_summary_i2v720_fp8.md: a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}
  // (MI355X_MICROARCH.md, LDS), i.e. 8 rows of one k group + the OTHER 8 rows of the next, and with the (row >> 1) & 7 swizzle two
  // k groups whose chunk indices differ by 2 land on the same 8 of 16 slots.  Chunk indices that differ by 1 do not.  Which 32 of a row's
  // 128 k values a lane holds is free as long as A and B agree - the MFMA sums over all of them.)
  const int fr = lane & 15, kg = lane >> 4;
  const int ar = wr * 64 + fr, br = wc * 32 + fr;   // (row + 16 i) keeps ((row >> 1) & 7)
  const int a_lo = ar * 128 + ((kg ^ ((ar >> 1) & 7)) << 4), a_hi = ar * 128 + (((4 + kg) ^ ((ar >> 1) & 7)) << 4);
  const int b_lo = br * 128 + ((kg ^ ((br >> 1) & 7)) << 4), b_hi = br * 128 + (((4 + kg) ^ ((br >> 1) & 7)) << 4);
  constexpr int FROWS = 16 * 128;

  // ---- prologue: tile 0 complete + A0,B0 of tile 1 ----
  dma_unit(p.A, offA[0], kbyte(0), smem + U_A0 * UNIT_BYTES, wave);
  dma_unit(p.W, offB[0], kbyte(0), smem + U_B0 * UNIT_BYTES, wave);
  dma_unit(p.W, offB[1], kbyte(0), smem + U_B1 * UNIT_BYTES, wave);
  dma_unit(p.A, offA[1], kbyte(0), smem + U_A1 * UNIT_BYTES, wave);
  dma_unit(p.A, offA[0], kbyte(1), smem + STAGE_BYTES + U_A0 * UNIT_BYTES, wave);
  dma_unit(p.W, offB[0], kbyte(1), smem + STAGE_BYTES + U_B0 * UNIT_BYTES, wave);
  if (TWO_PHASE) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // A0(0), B0(0), B1(0) landed (3 younger units in flight)
  else GF8_VMCNT8();
  GF8_BARRIER();
  if (wr == 1) GF8_BARRIER();  // stagger: group 1 runs one barrier behind group 0

  i32x8 af[4], b0f[2], b1f[2];

#define GF8_MFMA(AH, BF, BH)                                                                              \
  {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
      acc[(AH) * 4 + i][(BH) * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                 \
          BF[j], af[i], acc[(AH) * 4 + i][(BH) * 2 + j], 0, 0, 0, 0, 0, 0);                               \
    __builtin_amdgcn_s_setprio(0);                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  }

  if (TWO_PHASE) {
    // P1: read a0,b0,b1 | DMA B1(t+1),A1(t+1) -> s^1 | vmcnt(8) | a0 x b0, a0 x b1
    // P2: read a1       | DMA A0(t+2),B0(t+2) -> s   | vmcnt(6) | a1 x b0, a1 x b1      (hazards: see gemm256.hip)
    for (int t = 0; t < nt; ++t) {
      char* cur = smem + (t & 1) * STAGE_BYTES;
      char* oth = smem + ((t + 1) & 1) * STAGE_BYTES;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        b0f[j] = read_frag(cur + U_B0 * UNIT_BYTES + b_lo + j * FROWS, cur + U_B0 * UNIT_BYTES + b_hi + j * FROWS);
        b1f[j] = read_frag(cur + U_B1 * UNIT_BYTES + b_lo + j * FROWS, cur + U_B1 * UNIT_BYTES + b_hi + j * FROWS);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = read_frag(cur + U_A0 * UNIT_BYTES + a_lo + i * FROWS, cur + U_A0 * UNIT_BYTES + a_hi + i * FROWS);
      dma_unit(p.W, offB[1], kbyte(t + 1), oth + U_B1 * UNIT_BYTES, wave);
      dma_unit(p.A, offA[1], kbyte(t + 1), oth + U_A1 * UNIT_BYTES, wave);
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      GF8_BARRIER();
      GF8_MFMA(0, b0f, 0);
      GF8_MFMA(0, b1f, 1);
      GF8_BARRIER();
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = read_frag(cur + U_A1 * UNIT_BYTES + a_lo + i * FROWS, cur + U_A1 * UNIT_BYTES + a_hi + i * FROWS);
      dma_unit(p.A, offA[0], kbyte(t + 2), cur + U_A0 * UNIT_BYTES, wave);
      dma_unit(p.W, offB[0], kbyte(t + 2), cur + U_B0 * UNIT_BYTES, wave);
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      GF8_BARRIER();
      GF8_MFMA(1, b0f, 0);
      GF8_MFMA(1, b1f, 1);
      GF8_BARRIER();
    }
  } else
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * STAGE_BYTES;
    char* oth = smem + ((t + 1) & 1) * STAGE_BYTES;
    // ---------------- phase 1: a0 x b0 ----------------
#pragma unroll
    for (int j = 0; j < 2; ++j)
      b0f[j] = read_frag(cur + U_B0 * UNIT_BYTES + b_lo + j * FROWS, cur + U_B0 * UNIT_BYTES + b_hi + j * FROWS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      af[i] = read_frag(cur + U_A0 * UNIT_BYTES + a_lo + i * FROWS, cur + U_A0 * UNIT_BYTES + a_hi + i * FROWS);
    dma_unit(p.W, offB[1], kbyte(t + 1), oth + U_B1 * UNIT_BYTES, wave);
    GF8_VMCNT8();
    GF8_BARRIER();
    GF8_MFMA(0, b0f, 0);
    GF8_BARRIER();
    // ---------------- phase 2: a0 x b1 ----------------
#pragma unroll
    for (int j = 0; j < 2; ++j)
      b1f[j] = read_frag(cur + U_B1 * UNIT_BYTES + b_lo + j * FROWS, cur + U_B1 * UNIT_BYTES + b_hi + j * FROWS);
    dma_unit(p.A, offA[1], kbyte(t + 1), oth + U_A1 * UNIT_BYTES, wave);
    GF8_VMCNT8();
    GF8_BARRIER();
    GF8_MFMA(0, b1f, 1);
    GF8_BARRIER();
    // ---------------- phase 3: a1 x b1 ----------------
#pragma unroll
    for (int i = 0; i < 4; ++i)
      af[i] = read_frag(cur + U_A1 * UNIT_BYTES + a_lo + i * FROWS, cur + U_A1 * UNIT_BYTES + a_hi + i * FROWS);
    dma_unit(p.A, offA[0], kbyte(t + 2), cur + U_A0 * UNIT_BYTES, wave);
    GF8_BARRIER();
    GF8_MFMA(1, b1f, 1);
    GF8_BARRIER();
    // ---------------- phase 4: a1 x b0 ----------------
    dma_unit(p.W, offB[0], kbyte(t + 2), cur + U_B0 * UNIT_BYTES, wave);
    GF8_VMCNT8();
    GF8_BARRIER();
    GF8_MFMA(1, b0f, 0);
    GF8_BARRIER();
  }
#undef GF8_MFMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain tail DMA before the LDS is released
  if (wr == 0) GF8_BARRIER();                        // re-balance the stagger

  // ---- epilogue: a lane owns ONE row m and runs of 4 consecutive n (swapped MFMA operands) ----
  if (EPI == ICV_EPI_RESID_F32 && BATCH_EPI && n0 + BN <= p.N && p.nsplit == p.N) {
    // x[m, n] = resid[m, n] + gate[n] * (acc * a_scale[m] * w_scale[n] + bias[n]), the 16 residual loads of a half-tile
    // in flight together (gemm256.hip); rows past M load a clamped row and skip the store
    float4 bs[4], gt[4], sw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + kg * 4;
      sw[j] = *reinterpret_cast<const float4*>(p.w_scale + n);
      bs[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      gt[j] = p.gate ? *reinterpret_cast<const float4*>(p.gate + n) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float4 rs[4][4];
      float sa[4];
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int64_t m = m0 + wr * 128 + hf * 64 + ii * 16 + fr;
        const int64_t mc = m < p.M ? m : p.M - 1;
        sa[ii] = p.a_scale[mc];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          rs[ii][j] = *reinterpret_cast<const float4*>(p.resid + mc * p.ldr + n0 + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + kg * 4);
      }
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int64_t m = m0 + wr * 128 + hf * 64 + ii * 16 + fr;
        if (m < p.M) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + kg * 4;
            const f32x4 a = acc[hf * 4 + ii][j];
            const float4 r = rs[ii][j];
            // same operation order as the per-fragment path below: acc * (sa * sw) + bias, then r + gate * v
            const float v0 = a[0] * (sa[ii] * sw[j].x) + bs[j].x, v1 = a[1] * (sa[ii] * sw[j].y) + bs[j].y;
            const float v2 = a[2] * (sa[ii] * sw[j].z) + bs[j].z, v3 = a[3] * (sa[ii] * sw[j].w) + bs[j].w;
            *reinterpret_cast<float4*>((float*)p.out + m * p.ldo + n) =
                make_float4(r.x + gt[j].x * v0, r.y + gt[j].y * v1, r.z + gt[j].z * v2, r.w + gt[j].w * v3);
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t m = m0 + wr * 128 + (i >> 2) * 64 + (i & 3) * 16 + fr;
    const float sa = p.a_scale[m < p.M ? m : p.M - 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + kg * 4;
      if (m < p.M && n < p.N) {
        const float4 sw = *reinterpret_cast<const float4*>(p.w_scale + n);
        float v0 = acc[i][j][0] * (sa * sw.x), v1 = acc[i][j][1] * (sa * sw.y);
        float v2 = acc[i][j][2] * (sa * sw.z), v3 = acc[i][j][3] * (sa * sw.w);
        if (p.bias) {
          const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
          v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
        }
        const int64_t off = icv_out_offset(m, n, p.ldo, p.N, p.nsplit, p.split_stride);
        if (EPI == ICV_EPI_BF16 || EPI == ICV_EPI_GELU_BF16) {
          if (EPI == ICV_EPI_GELU_BF16) {
            v0 = gelu_tanh(v0); v1 = gelu_tanh(v1); v2 = gelu_tanh(v2); v3 = gelu_tanh(v3);
          }
          *reinterpret_cast<uint2*>((bf16_t*)p.out + off) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
        } else if (EPI == ICV_EPI_RESID_F32) {
          const float4 r = *reinterpret_cast<const float4*>(p.resid + m * p.ldr + n);
          float4 o;
          if (p.gate) {
            const float4 gt = *reinterpret_cast<const float4*>(p.gate + n);
            o = make_float4(r.x + gt.x * v0, r.y + gt.y * v1, r.z + gt.z * v2, r.w + gt.w * v3);
          } else {
            o = make_float4(r.x + v0, r.y + v1, r.z + v2, r.w + v3);
          }
          *reinterpret_cast<float4*>((float*)p.out + off) = o;
        } else {
          *reinterpret_cast<float4*>((float*)p.out + off) = make_float4(v0, v1, v2, v3);
        }
      }
    }
  }
}

template <int EPI, int SCH>
int launch(const Params& p, hipStream_t st) {
  static icv_dev_flags attr_set = {};
  if (int rc = icv_ensure_dynamic_lds((const void*)gemm_fp8_kernel<EPI, SCH>, LDS_BYTES, &attr_set, "icv_gemm_fp8")) return rc;
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  hipLaunchKernelGGL((gemm_fp8_kernel<EPI, SCH>), dim3((unsigned)nwg), dim3(512), LDS_BYTES, st, p);
  return icv_check_launch("icv_gemm_fp8");
}

}  // namespace gf8

extern "C" int icv_gemm_fp8(const void* A, int64_t lda, const float* a_scale, const void* W, int64_t ldw,
                            const float* w_scale, const float* bias, int64_t M, int64_t N, int64_t K, int epilogue,
                            void* out, int64_t ldo, int64_t nsplit, int64_t split_stride, const float* resid,
                            int64_t ldr, const float* gate, void* stream) {
  ICV_REQUIRE(A && W && a_scale && w_scale && out, "icv_gemm_fp8: null pointer");
  ICV_REQUIRE(M > 0 && N > 0 && K > 0 && N < (1LL << 31), "icv_gemm_fp8: empty problem or N too large");
  ICV_REQUIRE(K % 128 == 0, "icv_gemm_fp8: K=%lld must be a multiple of 128", (long long)K);
  ICV_REQUIRE(N % 4 == 0 && nsplit > 0 && nsplit % 4 == 0 && N % nsplit == 0, "icv_gemm_fp8: N=%lld / nsplit=%lld must be multiples of 4 with nsplit | N", (long long)N, (long long)nsplit);
  ICV_REQUIRE(lda % 16 == 0 && ldw % 16 == 0 && ldo % 4 == 0, "icv_gemm_fp8: lda/ldw must be multiples of 16 bytes, ldo of 4 elements");
  ICV_REQUIRE((uint64_t)M * (uint64_t)lda < (1ull << 32) && (uint64_t)N * (uint64_t)ldw < (1ull << 32), "icv_gemm_fp8: operand larger than 4 GiB");
  ICV_REQUIRE(epilogue != ICV_EPI_RESID_F32 || (resid && ldr % 4 == 0), "icv_gemm_fp8: RESID epilogue needs resid with ldr %% 4 == 0");
  gf8::Params p;
  p.A = (const char*)A; p.lda = lda; p.W = (const char*)W; p.ldw = ldw; p.a_scale = a_scale; p.w_scale = w_scale;
  p.bias = bias; p.M = M; p.N = N; p.K = K; p.out = out; p.ldo = ldo; p.nsplit = nsplit; p.split_stride = split_stride;
  p.resid = resid; p.ldr = ldr; p.gate = gate;
  p.tiles_m = (int)((M + gf8::BM - 1) / gf8::BM);
  p.tiles_n = (int)((N + gf8::BN - 1) / gf8::BN);
  hipStream_t st = (hipStream_t)stream;
  const bool r2 = icv_get_option_int("gemm_fp8_sched", 3) != 0;   // 0 = round 1's four-phase loop and per-fragment epilogue (A/B)
#define GF8_LAUNCH(E_) (r2 ? gf8::launch<E_, 3>(p, st) : gf8::launch<E_, 0>(p, st))
  switch (epilogue) {
    case ICV_EPI_BF16: return GF8_LAUNCH(ICV_EPI_BF16);
    case ICV_EPI_GELU_BF16: return GF8_LAUNCH(ICV_EPI_GELU_BF16);
    case ICV_EPI_RESID_F32: return GF8_LAUNCH(ICV_EPI_RESID_F32);
    case ICV_EPI_F32: return GF8_LAUNCH(ICV_EPI_F32);
  }
#undef GF8_LAUNCH
  icv_set_error("icv_gemm_fp8: unknown epilogue %d", epilogue);
  return 1;
}
