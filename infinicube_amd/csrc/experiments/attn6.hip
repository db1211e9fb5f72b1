// Flash attention forward, "ping-pong" structure (same math, fragments and LDS images as attn.hip / attn2.hip).
//
// Why: in attn.hip / attn2.hip every barrier interval of every wave contains MFMA work (QK^T or PV) AND the
// softmax VALU work, so the two waves that share a SIMD want the matrix pipe at the same time, and while either
// of them is in softmax the pipe idles behind the barrier — measured ~53 % MFMA-pipe occupancy.  Here the work
// of a 64-key tile is re-cut so that a phase is either ALL matrix or ALL vector for a wave:
//     BURST(t) = K(t) reads, S(t) = K(t) Q^T (16 MFMA)  +  V(t-1) reads, O += V(t-1)^T P(t-1)^T (16 MFMA)
//     SOFT(t)  = mask / running max / deferred rescale / P(t) = exp2(...)  + this wave's share of the staging
// (PV is software-pipelined one tile behind QK^T), one barrier per phase, and the two wave groups (waves 0-3 /
// 4-7 = one wave of each per SIMD) run ONE PHASE APART: while group 0 bursts 32 MFMAs group 1 does its vector
// work, then they swap.  Each SIMD's matrix pipe is fed by exactly one wave per phase.
//
// LDS: two stages of a UNIT U(t) = { K(t), V(t-1) } (what BURST(t) reads), 32 KiB each.  U(t) is read in phases
// 2t (group 0) and 2t+1 (group 1); group 0 writes its half of U(t+1) in its SOFT(t) (phase 2t+1), group 1 writes
// its half of U(t+2) in its SOFT(t) (phase 2t+2): a stage is rewritten only after both groups have read it and
// is complete one barrier before its first read.  Global loads run one iteration ahead in VGPRs.
#include "attn_common.h"

namespace att6 {

using attc::D;
using attc::NEG_BIG;
using attc::Params;
using attc::lds_read_tr16;
constexpr int KVB = 64;                  // keys per tile
constexpr int QB = 256;                  // query rows per block (8 waves x 32)
constexpr int KT_BYTES = KVB * D * 2;    // 16 KiB (K or V part of a unit)
constexpr int STAGE_BYTES = 2 * KT_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;  // 64 KiB

// phase timeline of block 0, waves 0 and 4 (VAR bit 8; tools/attn6_trace.py): s_memtime at phase boundaries
__device__ unsigned long long g_trace[2][1024];

// VAR bit flags: 1 = run the wave groups one phase apart (ping-pong), 4 = s_setprio(1) during a burst, 8 = trace
template <int VAR>
__global__ __launch_bounds__(512) void attn6_kernel(Params p) {
  constexpr bool PINGPONG = VAR & 1, SETPRIO = VAR & 4, TRACE = VAR & 8;
  constexpr bool ABL_NOSOFT = VAR & 16, ABL_NOSTAGE = VAR & 32, ABL_NOLDS = VAR & 64;   // timing ablations (wrong results)
  int trace_n = 0;
#define A6_STAMP()                                                                                            \
  if (TRACE && blockIdx.x == 0 && (threadIdx.x & 255) == 0 && trace_n < 1024) g_trace[threadIdx.x >> 8][trace_n++] = __builtin_amdgcn_s_memtime();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int hi = lane >> 5;
  const int l31 = lane & 31;

  int head, qb;
  attc::work_item(p, head, qb);
  const int64_t q0 = (int64_t)qb * QB + wave * 32;

  const bf16_t* qh = p.q + (int64_t)head * D;
  const bf16_t* kh = p.k + (int64_t)head * D;
  const bf16_t* vh = p.v + (int64_t)head * D;

  int64_t qr_c = q0 + l31;  // clamped query row (tail rows recompute the last row, never stored)
  qr_c = qr_c < p.Sq ? qr_c : p.Sq - 1;

  bf16x8 qf[8];
  {
    const bf16_t* qp = qh + qr_c * p.ldq + hi * 8;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) qf[ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16);
  }

  // ---- staging: thread owns chunk sc0 of keys sk0 and sk0 + 32 of K(u) and of V(u-1) ----
  const int sk0 = tid >> 4, sc0 = tid & 15;
  uint4 kreg0, kreg1, vreg0, vreg1;
  const int64_t last = p.Skv - 1;
#define A6_LOAD_UNIT(U_)                                                                  \
  {                                                                                       \
    int64_t kr0_ = (int64_t)(U_) * KVB + sk0, kr1_ = kr0_ + 32;                           \
    int64_t vr0_ = kr0_ - KVB, vr1_ = kr1_ - KVB;                                         \
    kr0_ = kr0_ < last ? kr0_ : last; kr1_ = kr1_ < last ? kr1_ : last;                   \
    vr0_ = vr0_ < 0 ? 0 : (vr0_ < last ? vr0_ : last);                                    \
    vr1_ = vr1_ < 0 ? 0 : (vr1_ < last ? vr1_ : last);                                    \
    kreg0 = *reinterpret_cast<const uint4*>(kh + kr0_ * p.ldk + sc0 * 8);                 \
    kreg1 = *reinterpret_cast<const uint4*>(kh + kr1_ * p.ldk + sc0 * 8);                 \
    vreg0 = *reinterpret_cast<const uint4*>(vh + vr0_ * p.ldv + sc0 * 8);                 \
    vreg1 = *reinterpret_cast<const uint4*>(vh + vr1_ * p.ldv + sc0 * 8);                 \
  }
  const int k_wr_off = sk0 * 256 + ((sc0 ^ (sk0 & 15)) << 4);
  const int v_wr_off = KT_BYTES + sk0 * 256 + ((sc0 << 4) ^ ((sk0 & 3) << 6));
#define A6_WRITE_UNIT(U_)                                                     \
  {                                                                           \
    char* s_ = smem + ((U_) & 1) * STAGE_BYTES;                               \
    *reinterpret_cast<uint4*>(s_ + k_wr_off) = kreg0;                         \
    *reinterpret_cast<uint4*>(s_ + k_wr_off + 8192) = kreg1;                  \
    *reinterpret_cast<uint4*>(s_ + v_wr_off) = vreg0;                         \
    *reinterpret_cast<uint4*>(s_ + v_wr_off + 8192) = vreg1;                  \
  }
#define A6_BARRIER()                                          \
  do {                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                             \
    asm volatile("" ::: "memory");                            \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)

  const int grp = PINGPONG ? __builtin_amdgcn_readfirstlane(wave >> 2) : 0;

  // ---- softmax state: O^T accumulator (query in the lane), running max, partial row sum ----
  f32x16 ot[4];
  float m_run, l_run;
  attc::load_state(p, qr_c, head, hi, ot, m_run, l_run);

  const int nt = (int)((p.Skv + KVB - 1) / KVB);   // tiles; units 0..nt (U(nt) = {-, V(nt-1)})
  A6_LOAD_UNIT(0);
  A6_WRITE_UNIT(0);
  if (grp == 1) {            // group 1 stages one unit further ahead (see header)
    A6_LOAD_UNIT(1);
    A6_WRITE_UNIT(1);
  }
  A6_LOAD_UNIT(1 + grp);
  // retire every prologue load with a wait the waitcnt pass can see (see attn.hip)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  A6_BARRIER();
  if (grp == 1) A6_BARRIER();

  const int k_row_off = l31 * 256;
  const int k_sw = l31 & 15;
  const int g = lane >> 4, t16 = lane & 15;
  const int v_key_lo = 4 * hi + (t16 >> 2);
  const int v_byte_lo = (g & 1) * 32 + (t16 & 3) * 8;
  const int v_sw = (t16 >> 2) << 6;

  f32x16 st[2];
  bf16x8 pf[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[i][j] = (__bf16)0.f;

// LDS fragment reads and MFMAs of a burst, written as an explicit software pipeline: hipcc otherwise issues every
// ds_read just before the MFMA that consumes it (lookahead ~1 MFMA), which exposes the LDS latency 24 times per
// burst — a burst is ONE wave's work, nothing else on the SIMD covers it.  A6_FENCE keeps the reads of a pipeline
// slot, and the MFMAs that consume earlier slots, in their slot (a sched_barrier: a plain compiler fence lets hipcc hoist
// the MFMAs back up to their reads); K fragments are read 4 MFMA pairs (~256 cycles) ahead of their use, V^T fragments 2.
#define A6_FENCE() __builtin_amdgcn_sched_barrier(0)
#define A6_READ_K(DS_)                                                                                   \
  if (!ABL_NOLDS) {                                                                                      \
    const int c_ = (DS_) * 2 + hi;                                                                       \
    kf[(DS_) * 2] = *reinterpret_cast<const bf16x8*>(ks + k_row_off + ((c_ ^ k_sw) << 4));               \
    kf[(DS_) * 2 + 1] = *reinterpret_cast<const bf16x8*>(ks + 8192 + k_row_off + ((c_ ^ k_sw) << 4));    \
  }
#define A6_QK(DS_)                                                                                       \
  {                                                                                                      \
    st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[(DS_) * 2], qf[DS_], st[0], 0, 0, 0);             \
    st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[(DS_) * 2 + 1], qf[DS_], st[1], 0, 0, 0);         \
  }
// PV step J_ (0..15): 16-key step kk = J_ >> 2, head-dim block d0 = J_ & 3; its V^T fragment lives in vf[J_ & 7]
#define A6_READ_V(J_)                                                                                    \
  if (!ABL_NOLDS) {                                                                                      \
    const int key0_ = ((J_) >> 2) * 16 + v_key_lo;                                                       \
    const int byte_ = (((J_) & 3) * 64 + v_byte_lo) ^ v_sw;                                              \
    const bf16x4 va_ = lds_read_tr16(vs + key0_ * 256 + byte_);                                          \
    const bf16x4 vb_ = lds_read_tr16(vs + (key0_ + 8) * 256 + byte_);                                    \
    bf16x8& d_ = vf[(J_) & 7];                                                                           \
    d_[0] = va_[0]; d_[1] = va_[1]; d_[2] = va_[2]; d_[3] = va_[3];                                      \
    d_[4] = vb_[0]; d_[5] = vb_[1]; d_[6] = vb_[2]; d_[7] = vb_[3];                                      \
  }
#define A6_PV(J_) ot[(J_) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[(J_) & 7], pf[(J_) >> 2], ot[(J_) & 3], 0, 0, 0);
// BURST(t): matrix work only.  DO_QK_ / DO_PV_ are literal 0/1 (first tile has no PV, the extra last burst no QK).
// Steady state = 8 quads of {QK^T k-step i (2 MFMA: st[0], st[1]), PV steps 2i, 2i+1 (2 MFMA: ot[.], ot[.])}: an
// accumulator is touched again only 4 (st) / 8 (ot) MFMAs later, so no MFMA waits for its own previous result
// (a kb-outer QK^T loop chains 8 dependent MFMAs); fragments are read two quads (8 MFMAs, ~256 cycles) ahead.
#define A6_BURST(T_, DO_QK_, DO_PV_)                                                                    \
  {                                                                                                     \
    const char* ks = smem + ((T_) & 1) * STAGE_BYTES;                                                   \
    const char* vs = ks + KT_BYTES;                                                                     \
    bf16x8 kf[16] = {}, vf[8] = {};                                                                     \
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);                                                         \
    if (DO_QK_) {                                                                                       \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                  \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;                                   \
    }                                                                                                   \
    if (DO_QK_ && DO_PV_) {                                                                             \
      A6_READ_K(0) A6_READ_K(1) A6_READ_V(0) A6_READ_V(1) A6_READ_V(2) A6_READ_V(3) A6_FENCE();         \
      A6_QK(0) A6_PV(0) A6_PV(1) A6_READ_K(2) A6_READ_V(4) A6_READ_V(5) A6_FENCE();                     \
      A6_QK(1) A6_PV(2) A6_PV(3) A6_READ_K(3) A6_READ_V(6) A6_READ_V(7) A6_FENCE();                     \
      A6_QK(2) A6_PV(4) A6_PV(5) A6_READ_K(4) A6_READ_V(8) A6_READ_V(9) A6_FENCE();                     \
      A6_QK(3) A6_PV(6) A6_PV(7) A6_READ_K(5) A6_READ_V(10) A6_READ_V(11) A6_FENCE();                   \
      A6_QK(4) A6_PV(8) A6_PV(9) A6_READ_K(6) A6_READ_V(12) A6_READ_V(13) A6_FENCE();                   \
      A6_QK(5) A6_PV(10) A6_PV(11) A6_READ_K(7) A6_READ_V(14) A6_READ_V(15) A6_FENCE();                 \
      A6_QK(6) A6_PV(12) A6_PV(13) A6_FENCE();                                                          \
      A6_QK(7) A6_PV(14) A6_PV(15)                                                                      \
    } else if (DO_QK_) {                                                                                \
      A6_READ_K(0) A6_READ_K(1) A6_READ_K(2) A6_READ_K(3) A6_FENCE();                                   \
      A6_QK(0) A6_READ_K(4) A6_FENCE();                                                                 \
      A6_QK(1) A6_READ_K(5) A6_FENCE();                                                                 \
      A6_QK(2) A6_READ_K(6) A6_FENCE();                                                                 \
      A6_QK(3) A6_READ_K(7) A6_FENCE();                                                                 \
      A6_QK(4) A6_QK(5) A6_QK(6) A6_QK(7)                                                               \
    } else {                                                                                            \
      A6_READ_V(0) A6_READ_V(1) A6_READ_V(2) A6_READ_V(3) A6_FENCE();                                   \
      A6_PV(0) A6_PV(1) A6_READ_V(4) A6_READ_V(5) A6_FENCE();                                           \
      A6_PV(2) A6_PV(3) A6_READ_V(6) A6_READ_V(7) A6_FENCE();                                           \
      A6_PV(4) A6_PV(5) A6_READ_V(8) A6_READ_V(9) A6_FENCE();                                           \
      A6_PV(6) A6_PV(7) A6_READ_V(10) A6_READ_V(11) A6_FENCE();                                         \
      A6_PV(8) A6_PV(9) A6_READ_V(12) A6_READ_V(13) A6_FENCE();                                         \
      A6_PV(10) A6_PV(11) A6_READ_V(14) A6_READ_V(15) A6_FENCE();                                       \
      A6_PV(12) A6_PV(13) A6_PV(14) A6_PV(15)                                                           \
    }                                                                                                   \
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);                                                         \
  }
// SOFT(t): vector work only: mask, running max (+ deferred rescale of O, which holds tiles < t), P(t) = exp2.
#define A6_SOFT(T_)                                                                                     \
  {                                                                                                     \
    const int64_t key0 = (int64_t)(T_) * KVB;                                                           \
    if (key0 + KVB > p.Skv) {                                                                           \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                  \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                  \
        const int64_t key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;                           \
        if (key >= p.Skv) st[kb][r] = NEG_BIG;                                                          \
      }                                                                                                 \
    }                                                                                                   \
    float mloc = st[0][0];                                                                              \
    _Pragma("unroll") for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[0][r]);                        \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[1][r]);                        \
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));                                                       \
    if (__any((mloc - m_run) * p.sc > p.thr)) {                                                         \
      const float m_new = fmaxf(m_run, mloc);                                                           \
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);                               \
      m_run = m_new;                                                                                    \
      l_run *= alpha;                                                                                   \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;                                 \
    }                                                                                                   \
    const float mb = -m_run * p.sc;                                                                     \
    float psum = 0.f;                                                                                   \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                    \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
      const float pv = __builtin_amdgcn_exp2f(fmaf(st[kb][r], p.sc, mb));                               \
      psum += pv;                                                                                       \
      pf[kb * 2 + (r >> 3)][r & 7] = (__bf16)pv;                                                        \
    }                                                                                                   \
    l_run += psum;                                                                                      \
  }
// this wave's share of the staging, done in its vector phase
#define A6_STAGE(T_)                                                                                    \
  {                                                                                                     \
    const int uw_ = (T_) + 1 + grp;                                                                     \
    if (TRACE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); A6_STAMP() }                          \
    if (uw_ <= nt) A6_WRITE_UNIT(uw_)                                                                   \
    if (TRACE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); A6_STAMP() }                        \
    if (uw_ + 1 <= nt) A6_LOAD_UNIT(uw_ + 1)                                                            \
    if (TRACE) { A6_STAMP() }                                                                           \
  }

  // t = 0: no PV yet
  A6_BURST(0, 1, 0)
  A6_BARRIER();
  A6_STAGE(0)
  A6_SOFT(0)
  A6_BARRIER();
  for (int t = 1; t < nt; ++t) {
    A6_STAMP()
    A6_BURST(t, 1, 1)
    A6_STAMP()
    A6_BARRIER();
    A6_STAMP()
    if (!ABL_NOSTAGE) A6_STAGE(t)     // first: the LDS writes then drain under the softmax VALU instead of under the barrier
    if (!ABL_NOSOFT) A6_SOFT(t)
    A6_STAMP()
    A6_BARRIER();
  }
  // t = nt: the pipelined PV of the last tile
  A6_BURST(nt, 0, 1)
  A6_BARRIER();
  A6_BARRIER();
  if (grp == 0) A6_BARRIER();

  attc::store_result(p, q0 + l31, head, hi, ot, m_run, l_run);
}

template <int VAR>
int launch(const Params& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn6_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      icv_set_error("attn6: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  const int64_t nwg = (int64_t)p.heads * p.nqb;
  hipLaunchKernelGGL(attn6_kernel<VAR>, dim3((unsigned)nwg), dim3(512), LDS_BYTES, st, p);
  return icv_check_launch("icv_attention(6)");
}

}  // namespace att6

int icv_attn6_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st) {
  attc::Params p;
  attc::fill_params(p, q, ldq, k, ldk, v, ldv, o, ldo, acc, ldacc, ml, state_in, state_out, Sq, Skv, heads, scale, att6::QB);
  const int64_t nwg = (int64_t)p.heads * p.nqb;
  ICV_REQUIRE(nwg < (1LL << 31), "icv_attention: grid too large");
  switch (var) {
    case 0: return att6::launch<0>(p, st);
    case 1: return att6::launch<1>(p, st);
    case 4: return att6::launch<4>(p, st);
    case 5: return att6::launch<5>(p, st);
    case 13: return att6::launch<13>(p, st);
    case 29: return att6::launch<29>(p, st);
    case 45: return att6::launch<45>(p, st);
    case 61: return att6::launch<61>(p, st);
    case 77: return att6::launch<77>(p, st);
    case 125: return att6::launch<125>(p, st);
  }
  icv_set_error("icv_attention: unknown attn6_variant %d", var);
  return 1;
}

// debug only (not part of include/icvideo.h): copy the phase timeline of the last attn6 variant-13 launch to the host
extern "C" int icv_debug_attn6_trace(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(att6::g_trace), sizeof(unsigned long long) * 2 * 1024) == hipSuccess ? 0 : 1;
}
