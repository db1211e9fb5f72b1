// Flash attention forward, second structure: same math, fragments and LDS
// images as experiments/attn1.hip (see there for the MFMA / key-permutation / swizzle design), but
//   * KV tile = 128 keys staged per barrier pair, consumed as two 64-key halves: ONE COMPLETE half
//     (K reads -> 16 QK^T MFMAs -> softmax -> V tr-reads -> 16 PV MFMAs) per barrier interval, i.e.
//     32 MFMAs per wave per barrier instead of 16, and no accumulator is live across a barrier;
//   * the two wave groups (waves 0-3 / 4-7, one of each per SIMD) run one barrier apart, so the 128-key
//     stage written during interval 2t+1 (group 0 at the end of its half 1, group 1 at the end of its
//     half 0) is never being read;
//   * optional carried softmax state (fp32 O accumulator + running max / sum per row) so that one
//     attention can be split over several launches along the KEY axis — the sequence-parallel path
//     consumes K/V chunks as the RCCL all-gather delivers them (seqpar.py).
#include "attn_common.h"

namespace att2 {

using attc::D;
using attc::NEG_BIG;
using attc::Params;
using attc::lds_read_tr16;
constexpr int KVB = 128;                 // keys per staged tile
constexpr int QB = 256;                  // query rows per block (8 waves x 32)
constexpr int KT_BYTES = KVB * D * 2;    // 32 KiB (K or V part of a stage)
constexpr int STAGE_BYTES = 2 * KT_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;  // 128 KiB
constexpr int HALF_BYTES = 64 * D * 2;      // 16 KiB
// VAR bit flags: 1 = stagger wave groups, 4 = s_setprio(1) around MFMA clusters, 8 = lazy max (see A2_HALF),
// 16 = unit scale (needs 8): scale * log2(e) == 1, i.e. the caller folded the softmax scale into K (the DiT does it in the
// K RMSNorm weight, same single bf16 rounding) -> the S accumulator starts at -m_ref and P = exp2(S) with no fma per element
template <int VAR>
__global__ __launch_bounds__(512) void attn2_kernel(Params p) {
  constexpr bool STAGGER = VAR & 1, SETPRIO = VAR & 4, LAZYMAX = VAR & 8, UNIT = VAR & 16;
  static_assert(!UNIT || LAZYMAX, "unit-scale path is built on the lazy-max path");
  const float p_lim = __builtin_amdgcn_exp2f(p.thr);   // lazy max: largest row-partial sum of P accepted without a rescale
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

  // ---- staging: thread owns chunk sc0 of keys sk0 + 32u (u = 0..3) of the 128-key tile ----
  const int sk0 = tid >> 4, sc0 = tid & 15;
  uint4 kreg0, kreg1, kreg2, kreg3, vreg0, vreg1, vreg2, vreg3;
  // Steady state: the tile base is wave-uniform (SGPR pair) and each thread adds a fixed 32-bit byte offset, so a
  // tile costs 8 loads and no address VALU; only a partial last tile takes the clamped 64-bit path.  (The kernel
  // is bound by the VALU issue port the MFMAs share — measured with s_memtime in experiments/attn6.hip — so every VALU
  // instruction removed from the loop is MFMA issue time.)
  const unsigned koff0 = (unsigned)(((int64_t)sk0 * p.ldk + sc0 * 8) * 2), kstep = (unsigned)(32 * p.ldk * 2);
  const unsigned voff0 = (unsigned)(((int64_t)sk0 * p.ldv + sc0 * 8) * 2), vstep = (unsigned)(32 * p.ldv * 2);
#define A2_LOAD_ROW(KR_, VR_, U_, T_)                                        \
  {                                                                          \
    int64_t kr_ = (int64_t)(T_) * KVB + sk0 + 32 * (U_);                     \
    kr_ = kr_ < p.Skv ? kr_ : p.Skv - 1;                                     \
    KR_ = *reinterpret_cast<const uint4*>(kh + kr_ * p.ldk + sc0 * 8);       \
    VR_ = *reinterpret_cast<const uint4*>(vh + kr_ * p.ldv + sc0 * 8);       \
  }
#define A2_LOAD_TILE(T_)                                                                       \
  {                                                                                            \
    if ((int64_t)((T_) + 1) * KVB <= p.Skv) {                                                  \
      const char* kt_ = reinterpret_cast<const char*>(kh + (int64_t)(T_) * KVB * p.ldk);       \
      const char* vt_ = reinterpret_cast<const char*>(vh + (int64_t)(T_) * KVB * p.ldv);       \
      kreg0 = *reinterpret_cast<const uint4*>(kt_ + koff0);                                    \
      vreg0 = *reinterpret_cast<const uint4*>(vt_ + voff0);                                    \
      kreg1 = *reinterpret_cast<const uint4*>(kt_ + (koff0 + kstep));                          \
      vreg1 = *reinterpret_cast<const uint4*>(vt_ + (voff0 + vstep));                          \
      kreg2 = *reinterpret_cast<const uint4*>(kt_ + (koff0 + 2 * kstep));                      \
      vreg2 = *reinterpret_cast<const uint4*>(vt_ + (voff0 + 2 * vstep));                      \
      kreg3 = *reinterpret_cast<const uint4*>(kt_ + (koff0 + 3 * kstep));                      \
      vreg3 = *reinterpret_cast<const uint4*>(vt_ + (voff0 + 3 * vstep));                      \
    } else {                                                                                   \
      A2_LOAD_ROW(kreg0, vreg0, 0, T_)                                                         \
      A2_LOAD_ROW(kreg1, vreg1, 1, T_)                                                         \
      A2_LOAD_ROW(kreg2, vreg2, 2, T_)                                                         \
      A2_LOAD_ROW(kreg3, vreg3, 3, T_)                                                         \
    }                                                                                          \
  }
  const int k_wr_off = sk0 * 256 + ((sc0 ^ (sk0 & 15)) << 4);
  const int v_wr_off = KT_BYTES + sk0 * 256 + ((sc0 << 4) ^ ((sk0 & 3) << 6));
#define A2_WRITE_TILE(STAGE_)                                                 \
  {                                                                           \
    char* s_ = (STAGE_);                                                      \
    *reinterpret_cast<uint4*>(s_ + k_wr_off) = kreg0;                         \
    *reinterpret_cast<uint4*>(s_ + k_wr_off + 8192) = kreg1;                  \
    *reinterpret_cast<uint4*>(s_ + k_wr_off + 16384) = kreg2;                 \
    *reinterpret_cast<uint4*>(s_ + k_wr_off + 24576) = kreg3;                 \
    *reinterpret_cast<uint4*>(s_ + v_wr_off) = vreg0;                         \
    *reinterpret_cast<uint4*>(s_ + v_wr_off + 8192) = vreg1;                  \
    *reinterpret_cast<uint4*>(s_ + v_wr_off + 16384) = vreg2;                 \
    *reinterpret_cast<uint4*>(s_ + v_wr_off + 24576) = vreg3;                 \
  }
#define A2_BARRIER()                                          \
  do {                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                             \
    asm volatile("" ::: "memory");                            \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)

  const int grp = STAGGER ? __builtin_amdgcn_readfirstlane(wave >> 2) : 0;

  // ---- softmax state: O^T accumulator (query in the lane), running max, partial row sum ----
  f32x16 ot[4];
  float m_run, l_run;
  attc::load_state(p, qr_c, head, hi, ot, m_run, l_run);

  const int nt = (int)((p.Skv + KVB - 1) / KVB);
  A2_LOAD_TILE(0);
  A2_WRITE_TILE(smem);
  // retire every prologue load with a wait the waitcnt pass can see (see experiments/attn1.hip)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  A2_BARRIER();
  if (nt > 1) A2_LOAD_TILE(1);
  if (grp == 1) A2_BARRIER();

  const int k_row_off = l31 * 256;
  const int k_sw = l31 & 15;
  const int g = lane >> 4, t16 = lane & 15;
  const int v_key_lo = 4 * hi + (t16 >> 2);
  const int v_byte_lo = (g & 1) * 32 + (t16 & 3) * 8;
  const int v_sw = (t16 >> 2) << 6;

// One 64-key half: S^T = K Q^T, mask, running max (+ deferred rescale), P = exp2, O^T += V^T P^T.
#define A2_HALF(KS_, VS_, KEY0_)                                                                      \
  {                                                                                                   \
    const char* ks = (KS_);                                                                           \
    const char* vs = (VS_);                                                                           \
    f32x16 st[2];                                                                                     \
    /* UNIT: reference baked into the accumulators of this half (0 while there is no reference yet) */ \
    const bool no_ref = UNIT && m_run < -1.0e29f;                                                     \
    float m_base = no_ref ? 0.f : m_run;                                                              \
    const float st0 = UNIT ? -m_base : 0.f;                                                           \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                  \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) st[kb][r] = st0;                                   \
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);                                                       \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                  \
    _Pragma("unroll") for (int ds = 0; ds < 8; ++ds) {                                                \
      const int c = ds * 2 + hi;                                                                      \
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + kb * 8192 + k_row_off + ((c ^ k_sw) << 4)); \
      st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], st[kb], 0, 0, 0);                  \
    }                                                                                                 \
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);                                                       \
    if ((KEY0_) + 64 > p.Skv) {                                                                       \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                \
        const int64_t key = (KEY0_) + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;                      \
        if (key >= p.Skv) st[kb][r] = NEG_BIG;                                                        \
      }                                                                                               \
    }                                                                                                 \
    if (!LAZYMAX) {                                                                                   \
      float mloc = st[0][0];                                                                          \
      _Pragma("unroll") for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[0][r]);                    \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[1][r]);                    \
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));                                                   \
      if (__any((mloc - m_run) * p.sc > p.thr)) {                                                     \
        const float m_new = fmaxf(m_run, mloc);                                                       \
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);                           \
        m_run = m_new;                                                                                \
        l_run *= alpha;                                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;                             \
      }                                                                                               \
    }                                                                                                 \
    float mb = -m_run * p.sc;                                                                         \
    float psum = 0.f;                                                                                 \
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);                                                       \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                \
      bf16x8 pf[2];                                                                                   \
      float ps = 0.f;                                                                                 \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                \
        const float pv = UNIT ? __builtin_amdgcn_exp2f(st[kb][r]) : __builtin_amdgcn_exp2f(fmaf(st[kb][r], p.sc, mb)); \
        ps += pv;                                                                                     \
        pf[r >> 3][r & 7] = (__bf16)pv;                                                               \
      }                                                                                               \
      /* lazy max: no row-max pass at all while P stays small against the current reference m_run.   \
         P > 0, so the 16-key partial sum bounds every P of the block: ps <= 2^thr proves no element \
         outgrew the reference by more than thr (inf / NaN fail the test too).  Only then is the     \
         block's true max taken, O and l rescaled to it, and P recomputed (first tiles, rare later). */ \
      if (LAZYMAX && __any(!(ps <= p_lim) || no_ref)) {                                               \
        float mloc = st[kb][0];                                                                       \
        _Pragma("unroll") for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[kb][r]);                 \
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));                                                 \
        if (UNIT) mloc += m_base;                      /* st = s - m_base */                           \
        const float m_new = fmaxf(m_run, mloc);                                                       \
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);                           \
        m_run = m_new;                                                                                \
        l_run = (l_run + psum) * alpha;                                                               \
        psum = 0.f;                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;                             \
        mb = -m_run * p.sc;                                                                           \
        if (UNIT) {                                    /* re-base this and the later blocks of the half */ \
          const float dm = m_new - m_base;                                                            \
          m_base = m_new;                                                                             \
          _Pragma("unroll") for (int j = 0; j < 2; ++j)                                               \
            if (j >= kb) { _Pragma("unroll") for (int r = 0; r < 16; ++r) st[j][r] -= dm; }           \
        }                                                                                             \
        ps = 0.f;                                                                                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                              \
          const float pv = UNIT ? __builtin_amdgcn_exp2f(st[kb][r]) : __builtin_amdgcn_exp2f(fmaf(st[kb][r], p.sc, mb)); \
          ps += pv;                                                                                   \
          pf[r >> 3][r & 7] = (__bf16)pv;                                                             \
        }                                                                                             \
      }                                                                                               \
      psum += ps;                                                                                     \
      _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                              \
        const int kk = kb * 2 + hf;                                                                   \
        _Pragma("unroll") for (int d0 = 0; d0 < 4; ++d0) {                                            \
          const int key0 = kk * 16 + v_key_lo;                                                        \
          const int byte = (d0 * 64 + v_byte_lo) ^ v_sw;                                              \
          const bf16x4 va = lds_read_tr16(vs + key0 * 256 + byte);                                    \
          const bf16x4 vb = lds_read_tr16(vs + (key0 + 8) * 256 + byte);                              \
          bf16x8 vf;                                                                                  \
          vf[0] = va[0]; vf[1] = va[1]; vf[2] = va[2]; vf[3] = va[3];                                 \
          vf[4] = vb[0]; vf[5] = vb[1]; vf[6] = vb[2]; vf[7] = vb[3];                                 \
          ot[d0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[hf], ot[d0], 0, 0, 0);              \
        }                                                                                             \
      }                                                                                               \
    }                                                                                                 \
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);                                                       \
    l_run += psum;                                                                                    \
  }

  for (int t = 0; t < nt; ++t) {
    char* stage = smem + (t & 1) * STAGE_BYTES;
    char* other = smem + ((t + 1) & 1) * STAGE_BYTES;
    const int64_t key0 = (int64_t)t * KVB;
    // ---- interval A: keys [key0, key0 + 64) (never empty) ----
    A2_HALF(stage, stage + KT_BYTES, key0)
    if (grp == 1) {
      if (t + 1 < nt) A2_WRITE_TILE(other);
      if (t + 2 < nt) A2_LOAD_TILE(t + 2);
    }
    A2_BARRIER();
    // ---- interval B: keys [key0 + 64, key0 + 128); skipped when entirely past Skv ----
    if (key0 + 64 < p.Skv) A2_HALF(stage + HALF_BYTES, stage + KT_BYTES + HALF_BYTES, key0 + 64)
    if (grp == 0) {
      if (t + 1 < nt) A2_WRITE_TILE(other);
      if (t + 2 < nt) A2_LOAD_TILE(t + 2);
    }
    A2_BARRIER();
  }
  if (grp == 0) A2_BARRIER();
#undef A2_HALF

  attc::store_result(p, q0 + l31, head, hi, ot, m_run, l_run);
}

template <int VAR>
int launch(const Params& p, hipStream_t st) {
  static icv_dev_flags attr_set = {};
  if (int rc = icv_ensure_dynamic_lds((const void*)attn2_kernel<VAR>, LDS_BYTES, &attr_set, "attn2")) return rc;
  const int64_t nwg = (int64_t)p.heads * p.nqb;
  hipLaunchKernelGGL(attn2_kernel<VAR>, dim3((unsigned)nwg), dim3(512), LDS_BYTES, st, p);
  return icv_check_launch("icv_attention(2)");
}

}  // namespace att2

int icv_attn2_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st) {
  att2::Params p;
  attc::fill_params(p, q, ldq, k, ldk, v, ldv, o, ldo, acc, ldacc, ml, state_in, state_out, Sq, Skv, heads, scale, att2::QB);
  if (p.sc == 1.0f && (var & 8) && icv_get_option_int("attn_unit_scale", 1)) var |= 16;   // unit scale (attn_common.h fill_params snaps |sc - 1| < 1e-6 to exactly 1)
  switch (var) {
    case 0: return att2::launch<0>(p, st);
    case 1: return att2::launch<1>(p, st);
    case 4: return att2::launch<4>(p, st);
    case 5: return att2::launch<5>(p, st);
    case 12: return att2::launch<12>(p, st);
    case 13: return att2::launch<13>(p, st);
    case 28: return att2::launch<28>(p, st);
    case 29: return att2::launch<29>(p, st);
  }
  icv_set_error("icv_attention_fwd: unknown attn2 variant %d", var);
  return 1;
}
