// Flash attention forward, ONE WAVE PER SIMD with explicit register placement.
// Same math, fragments, key-permutation trick and LDS images as attn.hip; structure of attn3.hip (4 waves
// x 64 query rows = two 32-row sub-blocks A, B per wave, so every K / V fragment read from LDS feeds two
// MFMAs), with the two things hipcc got wrong there fixed by hand:
//   * the O^T accumulators (2 x 64 registers) live in AGPRs: the PV MFMAs are inline asm with "+a"
//     accumulators and VGPR A/B operands, so the Q fragments stay in VGPRs (attn3: hipcc parked Q in
//     AGPRs and copied 8 registers back before every MFMA);
//   * K/V tiles arrive by inline-asm LDS-DMA into a 4-stage ring with a counted vmcnt (attn4.hip):
//     no staging VGPRs, no ds_write.
// Per 64-key tile the wave's stream is
//     QK^T_A (16 MFMA)  |  QK^T_B (16 MFMA) || softmax_A (VALU)  |  PV_A (16 MFMA) || softmax_B  |  PV_B
// i.e. each VALU-heavy softmax sits next to 16 independent MFMAs of the same wave.
// Hazards hipcc does not pad for asm (guide §5.7): a VALU-written B operand (P, from v_cvt_pk) needs wait
// states before the MFMA -> every asm MFMA group starts with s_nop 1; an AGPR written by an asm MFMA needs
// up to 18 wait states before a non-MFMA read -> agpr_fence() before the rare rescale and the epilogue.
#include "attn_common.h"

namespace att5 {

using attc::D;
using attc::NEG_BIG;
using attc::Params;
using attc::lds_read_tr16;
constexpr int KVB = 64;
constexpr int QB = 256;                        // 4 waves x 64 rows
constexpr int TILE_BYTES = KVB * D * 2;        // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // 32 KiB
constexpr int NSTAGE = 4;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// four PV MFMAs (d0 = 0..3) sharing one P fragment; accumulators in AGPRs
__device__ __forceinline__ void pv4(f32x16& o0, f32x16& o1, f32x16& o2, f32x16& o3, const bf16x8& v0,
                                    const bf16x8& v1, const bf16x8& v2, const bf16x8& v3, const bf16x8& pf) {
  // volatile: the PV groups keep their textual position, so softmax code written between two groups stays
  // between them (ordinary loads / VALU may still be scheduled around the statement)
  asm volatile("s_nop 1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %4, %8, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %1, %5, %8, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %2, %6, %8, %2\n\t"
      "v_mfma_f32_32x32x16_bf16 %3, %7, %8, %3"
      : "+a"(o0), "+a"(o1), "+a"(o2), "+a"(o3)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(pf));
}

// wait states between the last asm MFMA write of the accumulators and a non-MFMA access
__device__ __forceinline__ void agpr_fence(f32x16 (&o)[4]) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+a"(o[0]), "+a"(o[1]), "+a"(o[2]), "+a"(o[3]));
}

template <int VAR>
__global__ __launch_bounds__(256, 1) void attn5_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int l31 = lane & 31;

  int head, qb;
  attc::work_item(p, head, qb);
  const int64_t q0 = (int64_t)qb * QB + wave * 64;
  const bf16_t* qh = p.q + (int64_t)head * D;
  const bf16_t* kh = p.k + (int64_t)head * D;
  const bf16_t* vh = p.v + (int64_t)head * D;

  int64_t qr_c[2];
  f32x16 ot[2][4];
  float m_run[2], l_run[2];
  bf16x8 qf[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int64_t r = q0 + s * 32 + l31;
    qr_c[s] = r < p.Sq ? r : p.Sq - 1;
    attc::load_state(p, qr_c[s], head, hi, ot[s], m_run[s], l_run[s]);
    const bf16_t* qp = qh + qr_c[s] * p.ldq + hi * 8;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) qf[s][ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // retire the ordinary prologue loads before any DMA is in flight

  // ---- LDS-DMA mapping: 4 waves, instruction j (0..3) of this wave covers keys (wave*4 + j)*4 + lane/16 ----
  const int nt = (int)((p.Skv + KVB - 1) / KVB);
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const int pc = lane & 15;
#define A5_DMA_TILE(T_)                                                                              \
  {                                                                                                  \
    const int tt_ = (T_) < nt ? (T_) : nt - 1;                                                       \
    const unsigned l0_ = lds_base + (unsigned)(((T_) & (NSTAGE - 1)) * STAGE_BYTES + (wave * 4) * 1024); \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                               \
      const int dk_ = (wave * 4 + j_) * 4 + (lane >> 4);                                             \
      int64_t r_ = (int64_t)tt_ * KVB + dk_;                                                         \
      r_ = r_ < p.Skv ? r_ : p.Skv - 1;                                                              \
      dma16(kh + r_ * p.ldk + (pc ^ (dk_ & 15)) * 8, l0_ + j_ * 1024);                               \
      dma16(vh + r_ * p.ldv + (pc ^ ((dk_ & 3) << 2)) * 8, l0_ + TILE_BYTES + j_ * 1024);            \
    }                                                                                                \
  }
#define A5_VMCNT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define A5_BARRIER()                                          \
  do {                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                             \
    asm volatile("" ::: "memory");                            \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)

  A5_DMA_TILE(0);
  A5_DMA_TILE(1);
  A5_VMCNT8();
  A5_BARRIER();

  const int k_row_off = l31 * 256;
  const int k_sw = l31 & 15;
  const int g = lane >> 4, t16 = lane & 15;
  const int v_key_lo = 4 * hi + (t16 >> 2);
  const int v_byte_lo = (g & 1) * 32 + (t16 & 3) * 8;
  const int v_sw = (t16 >> 2) << 6;

// S^T_s = K Q_s^T (builtin MFMAs, VGPR accumulators)
#define A5_QK(S_, ST_)                                                                                 \
  {                                                                                                    \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                   \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) ST_[kb][r] = 0.f;                                   \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                   \
    _Pragma("unroll") for (int ds = 0; ds < 8; ++ds) {                                                 \
      const int c = ds * 2 + hi;                                                                       \
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + kb * 8192 + k_row_off + ((c ^ k_sw) << 4)); \
      ST_[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[S_][ds], ST_[kb], 0, 0, 0);             \
    }                                                                                                  \
  }
// softmax_s part 1: mask, running max (+ deferred rescale) -> MB_ = -m*sc
#define A5_SM_MAX(S_, ST_, MB_)                                                                        \
  {                                                                                                    \
    if (tail) {                                                                                        \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                 \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                 \
        const int64_t key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;                          \
        if (key >= p.Skv) ST_[kb][r] = NEG_BIG;                                                        \
      }                                                                                                \
    }                                                                                                  \
    float mloc = ST_[0][0];                                                                            \
    _Pragma("unroll") for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, ST_[0][r]);                      \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, ST_[1][r]);                      \
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));                                                      \
    if (__any((mloc - m_run[S_]) * p.sc > p.thr)) {                                                    \
      const float m_new = fmaxf(m_run[S_], mloc);                                                      \
      const float alpha = __builtin_amdgcn_exp2f((m_run[S_] - m_new) * p.sc);                          \
      m_run[S_] = m_new;                                                                               \
      l_run[S_] *= alpha;                                                                              \
      agpr_fence(ot[S_]);                                                                              \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) ot[S_][i][r] *= alpha;                            \
    }                                                                                                  \
    MB_ = -m_run[S_] * p.sc;                                                                           \
  }
// softmax_s part 2 for one 32-key block KB_: P = exp2(S*sc + mb) -> two bf16 fragments
#define A5_SM_EXP(S_, ST_, PF_, MB_, KB_)                                                              \
  {                                                                                                    \
    float ps = 0.f;                                                                                    \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                   \
      const float pv = __builtin_amdgcn_exp2f(fmaf(ST_[KB_][r], p.sc, MB_));                           \
      ps += pv;                                                                                        \
      PF_[(KB_) * 2 + (r >> 3)][r & 7] = (__bf16)pv;                                                   \
    }                                                                                                  \
    l_run[S_] += ps;                                                                                   \
  }
// O^T_s += V^T P_s^T (asm MFMAs, AGPR accumulators); V fragments are re-read per sub-block and
// double-buffered by hand: the reads of key-slice kk+1 are issued before the MFMAs of slice kk.
#define A5_VREAD(VF_, KK_)                                                                             \
  _Pragma("unroll") for (int d0 = 0; d0 < 4; ++d0) {                                                   \
    const int key_l = (KK_) * 16 + v_key_lo;                                                           \
    const int byte = (d0 * 64 + v_byte_lo) ^ v_sw;                                                     \
    const bf16x4 va = lds_read_tr16(vs + key_l * 256 + byte);                                          \
    const bf16x4 vb = lds_read_tr16(vs + (key_l + 8) * 256 + byte);                                    \
    VF_[d0][0] = va[0]; VF_[d0][1] = va[1]; VF_[d0][2] = va[2]; VF_[d0][3] = va[3];                    \
    VF_[d0][4] = vb[0]; VF_[d0][5] = vb[1]; VF_[d0][6] = vb[2]; VF_[d0][7] = vb[3];                    \
  }
#define A5_PV(S_, PF_)                                                                                 \
  {                                                                                                    \
    bf16x8 vfa[4], vfb[4];                                                                             \
    A5_VREAD(vfa, 0)                                                                                   \
    A5_VREAD(vfb, 1)                                                                                   \
    pv4(ot[S_][0], ot[S_][1], ot[S_][2], ot[S_][3], vfa[0], vfa[1], vfa[2], vfa[3], PF_[0]);           \
    A5_VREAD(vfa, 2)                                                                                   \
    pv4(ot[S_][0], ot[S_][1], ot[S_][2], ot[S_][3], vfb[0], vfb[1], vfb[2], vfb[3], PF_[1]);           \
    A5_VREAD(vfb, 3)                                                                                   \
    pv4(ot[S_][0], ot[S_][1], ot[S_][2], ot[S_][3], vfa[0], vfa[1], vfa[2], vfa[3], PF_[2]);           \
    pv4(ot[S_][0], ot[S_][1], ot[S_][2], ot[S_][3], vfb[0], vfb[1], vfb[2], vfb[3], PF_[3]);           \
  }

  for (int t = 0; t < nt; ++t) {
    const char* ks = smem + (t & (NSTAGE - 1)) * STAGE_BYTES;
    const char* vs = ks + TILE_BYTES;
    const int64_t key0 = (int64_t)t * KVB;
    const bool tail = key0 + KVB > p.Skv;

    A5_DMA_TILE(t + 2);

    f32x16 stA[2], stB[2];
    bf16x8 pfA[4], pfB[4];
    float mbA, mbB;
    // phase 0: QK^T_A (MFMA only)
    A5_QK(0, stA)
    __builtin_amdgcn_sched_barrier(0);
    // phase 1: QK^T_B (MFMA + K reads) shares a scheduling region with softmax_A (VALU)
    A5_QK(1, stB)
    A5_SM_MAX(0, stA, mbA)
    A5_SM_EXP(0, stA, pfA, mbA, 0)
    A5_SM_EXP(0, stA, pfA, mbA, 1)
    __builtin_amdgcn_sched_barrier(0);
    // phase 2: PV_A (asm MFMA groups) interleaved by hand with softmax_B
    {
      bf16x8 vfa[4], vfb[4];
      A5_VREAD(vfa, 0)
      A5_VREAD(vfb, 1)
      pv4(ot[0][0], ot[0][1], ot[0][2], ot[0][3], vfa[0], vfa[1], vfa[2], vfa[3], pfA[0]);
      A5_SM_MAX(1, stB, mbB)
      A5_VREAD(vfa, 2)
      pv4(ot[0][0], ot[0][1], ot[0][2], ot[0][3], vfb[0], vfb[1], vfb[2], vfb[3], pfA[1]);
      A5_SM_EXP(1, stB, pfB, mbB, 0)
      A5_VREAD(vfb, 3)
      pv4(ot[0][0], ot[0][1], ot[0][2], ot[0][3], vfa[0], vfa[1], vfa[2], vfa[3], pfA[2]);
      A5_SM_EXP(1, stB, pfB, mbB, 1)
      pv4(ot[0][0], ot[0][1], ot[0][2], ot[0][3], vfb[0], vfb[1], vfb[2], vfb[3], pfA[3]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // phase 3: PV_B
    A5_PV(1, pfB)

    A5_VMCNT8();
    A5_BARRIER();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  agpr_fence(ot[0]);
  agpr_fence(ot[1]);
  attc::store_result(p, q0 + l31, head, hi, ot[0], m_run[0], l_run[0]);
  attc::store_result(p, q0 + 32 + l31, head, hi, ot[1], m_run[1], l_run[1]);
}

}  // namespace att5

int icv_attn5_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st) {
  att5::Params p;
  attc::fill_params(p, q, ldq, k, ldk, v, ldv, o, ldo, acc, ldacc, ml, state_in, state_out, Sq, Skv, heads, scale, att5::QB);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)att5::attn5_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, att5::LDS_BYTES);
    if (e != hipSuccess) {
      icv_set_error("attn5: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  (void)var;
  hipLaunchKernelGGL(att5::attn5_kernel<0>, dim3((unsigned)((int64_t)p.heads * p.nqb)), dim3(256), att5::LDS_BYTES, st, p);
  return icv_check_launch("icv_attention(5)");
}
