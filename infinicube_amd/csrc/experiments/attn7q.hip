// EXPERIMENT (built with ICV_EXPERIMENTS=1 only; measured slower than attn7p.hip, see attention.hip and profiles/r06/attn7q_pipelined_bf16_negative.txt).
// K6, bf16, software-pipelined (round 6): the long-key self-attention launch (plain, or one key chunk with carried state) with
// csrc/attn8.hip's loop structure carried over to the bf16 MFMA - what the e4m3 kernel's counters and A/B showed to be the lever
// (profiles/r06/attn8_pipelined_ab.txt): inside EACH wave, tile t's softmax is issued in the shadow of matrix instructions that do
// not depend on it.  One iteration of the key loop:
//     S(t+1) = K(t+1) Q^T   16 x v_mfma_f32_32x32x16_bf16, each followed by 1/16 of tile t's softmax (2 exponentials, 2 row-sum
//                           adds, one packed bf16 convert) and the K fragment read of the MFMA two steps on;
//     lazy-max check of each 32-key block (attn7.hip's rule; the rare re-base also corrects the already started S(t+1));
//     O += V(t) P(t)        16 MFMAs fed by transposing LDS reads one step ahead;
//     s_waitcnt vmcnt(0) + barrier (tile t+2 was requested at the top of the iteration into the stage K(t-2) / V(t-2) left).
// attn7.hip / attn7p.hip run S -> softmax -> O strictly in that order per wave and rely on the OTHER wave of the SIMD to be in the
// opposite phase; here the matrix pipe has its own wave's next S under every exponential.
// Fragment, swizzle and LDS-DMA conventions: attn7.hip (8 waves x 32 query rows, 64-key tiles, 4 stages of K|V = 128 KiB, unit scale
// with the running reference in the first MFMA's C operand).  The ring position is a compile-time constant in each of the 4 unrolled
// bodies (every LDS address = a lane register + an immediate), the two tile pointers are scalars that advance by a constant.
// Unit scale only (the DiT folds scale * log2 e into K): other scales stay on attn7p.hip.
// Replaces: flash_attention of the fork's self-attention [EXT]; SURVEY.md §8a K6.
#include "attn_common.h"

namespace att7q {

using attc::D;
using attc::NEG_BIG;
using attc::lds_read_tr16;
constexpr int KVB = 64;
constexpr int TILE_BYTES = KVB * D * 2;      // 16 KiB (K or V)
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // 32 KiB
constexpr int NST = 4, NW = 8, NI = 16 / NW, QB = NW * 32;
constexpr int LDS_BYTES = NST * STAGE_BYTES;

__device__ __forceinline__ void dma16s(const void* base, unsigned off, unsigned lds_dst) {
  // wave-uniform base (SGPR pair) + per-lane 32-bit byte offset; m0 declared clobbered (nothing else lives in it)
  const uint64_t b = (uint64_t)(uintptr_t)base;
  const uint64_t sbase = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)b);
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(off), "s"(sbase), "s"(lds_dst)
      : "memory", "m0");
}

template <int PFD>   // K fragment reads issued PFD MFMAs ahead (1 or 2)
__global__ __launch_bounds__(NW * 64) void attn7q_kernel(attc::Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int l31 = lane & 31;
  const float p_lim = __builtin_amdgcn_exp2f(p.thr);

  int head, qb;
  attc::work_item(p, head, qb);
  const int64_t q0 = (int64_t)qb * QB + wave * 32;
  const unsigned long long t_start = p.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
  int64_t qr_c = q0 + l31;
  qr_c = qr_c < p.Sq ? qr_c : p.Sq - 1;

  bf16x8 qf[8];
  {
    const bf16_t* qp = p.q + (int64_t)head * D + qr_c * p.ldq + hi * 8;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) qf[ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16);
  }
  f32x16 ot[4];
  float m_run, l_run;
  attc::load_state(p, qr_c, head, hi, ot, m_run, l_run);
  __builtin_amdgcn_s_waitcnt(0x0F70);     // every VGPR-destination load retires before the first LDS-DMA request (and before the loop: a
                                          // pending one would make the compiler's wait for it a vmcnt(0) inside the loop)
  // the running reference of the exponentials (0 while there is none): subtracted in the softmax segment - two more vector instructions
  // per MFMA, in its shadow - instead of riding in the first MFMA's C operand as in attn7.hip: S(t) and S(t+1) both in flight leave no
  // 16 registers for that operand, and the re-base needs no fix-up of the already started S(t+1)
  float m_base = m_run < -1.0e29f ? 0.f : m_run;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // ---- LDS-DMA lane mapping (attn7.hip): a request = 4 key rows x 256 B; lane -> (row, 16-byte chunk) ----
  const int pc = lane & 15;
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const int nt = (int)((p.Skv + KVB - 1) / KVB);
  const int rag_rows = (int)(p.Skv - (int64_t)(nt - 1) * KVB);          // rows of the last tile (1..64)
  unsigned ko[NI], vo[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int dkey = (wave * NI + j) * 4 + (lane >> 4);
    ko[j] = (unsigned)(((int64_t)dkey * p.ldk + (pc ^ (dkey & 15)) * 8) * 2);
    vo[j] = (unsigned)(((int64_t)dkey * p.ldv + (pc ^ ((dkey & 3) << 2)) * 8) * 2);
  }
  const bf16_t* kp_ = p.k + (int64_t)head * D;      // the tile the next request reads (scalar), advancing by a constant
  const bf16_t* vp_ = p.v + (int64_t)head * D;
  const int64_t kstep = (int64_t)KVB * p.ldk, vstep = (int64_t)KVB * p.ldv;
  int d_t = 0;
  // tile min(T_, nt - 1) -> ring stage T_ & 3 (a request past the end re-reads the last tile into a dead stage: uniform request counts)
#define Q_DMA_TILE(T_)                                                                                  \
  {                                                                                                     \
    const unsigned l0_ = lds_base + (unsigned)(((T_) & (NST - 1)) * STAGE_BYTES + (wave * NI) * 1024);  \
    if (d_t == nt - 1 && rag_rows < KVB) {                                                              \
      _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_) { /* the ragged last tile re-reads its last row into the padding rows */ \
        const int dkey_ = (wave * NI + j_) * 4 + (lane >> 4);                                           \
        const int dk_ = dkey_ < rag_rows ? dkey_ : rag_rows - 1;                                        \
        dma16s(kp_, (unsigned)(((int64_t)dk_ * p.ldk + (pc ^ (dkey_ & 15)) * 8) * 2), l0_ + j_ * 1024); \
        dma16s(vp_, (unsigned)(((int64_t)dk_ * p.ldv + (pc ^ ((dkey_ & 3) << 2)) * 8) * 2), l0_ + TILE_BYTES + j_ * 1024); \
      }                                                                                                 \
    } else {                                                                                            \
      _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_) dma16s(kp_, ko[j_], l0_ + j_ * 1024);           \
      _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_) dma16s(vp_, vo[j_], l0_ + TILE_BYTES + j_ * 1024); \
    }                                                                                                   \
    if (d_t < nt - 1) {                                                                                 \
      d_t = d_t + 1;                                                                                    \
      kp_ += kstep;                                                                                     \
      vp_ += vstep;                                                                                     \
    }                                                                                                   \
  }
#define Q_BARRIER()                                           \
  do {                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                             \
    asm volatile("" ::: "memory");                            \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)

  // fragment read offsets (attn7.hip)
  const int k_row_off = l31 * 256;
  const int k_sw = l31 & 15;
  const int g4 = lane >> 4, t16 = lane & 15;
  const int v_key_lo = 4 * hi + (t16 >> 2);
  const int v_byte_lo = (g4 & 1) * 32 + (t16 & 3) * 8;
  const int v_sw = (t16 >> 2) << 6;

  const unsigned ka0 = (unsigned)(k_row_off + ((hi ^ k_sw) << 4));            // K: row l31, chunk (2 ds + hi) ^ k_sw = (hi ^ k_sw) ^ (ds << 1)
  const unsigned va0 = (unsigned)(v_key_lo * 256 + (v_byte_lo ^ v_sw));       // V: row v_key_lo (+ 16 kk, + 8), byte (64 d0 + v_byte_lo) ^ v_sw
  Q_DMA_TILE(0);
  Q_DMA_TILE(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  Q_BARRIER();

  f32x16 stx[2][2];
  {                                   // S(0)
    const char* ks = smem + k_row_off;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + kb * 8192 + (((ds * 2 + hi) ^ k_sw) << 4));
        stx[0][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], ds == 0 ? zero16 : stx[0][kb], 0, 0, 0);
      }
  }

  for (int t4 = 0; t4 < nt; t4 += NST) {
#pragma unroll
    for (int ti = 0; ti < NST; ++ti) {
      const int t = t4 + ti;
      if (t >= nt) break;
      f32x16(&st)[2] = stx[ti & 1];
      f32x16(&sn)[2] = stx[(ti + 1) & 1];
      if (t == nt - 1 && rag_rows < KVB) {            // the ragged last tile: its padding keys do not exist (S(t) was finished an iteration ago)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= rag_rows) st[kb][r] = NEG_BIG;
          }
      }
      Q_DMA_TILE(t + 2);
      const bool no_ref = m_run < -1.0e29f;
      // fragment addresses = ONE lane register per operand, re-derived per read with an XOR of an immediate (the swizzles are XORs of
      // bits the k-step / d-block index owns alone) + an immediate offset: the eight + four precomputed address registers of attn7.hip do
      // not fit next to two S tiles in flight.  The empty asm keeps the compiler from hoisting the XORs back out of the loop.
      unsigned ka = ka0, va0_ = va0;
      asm volatile("" : "+v"(ka), "+v"(va0_));
      const char* const kst = smem + ((ti + 1) & (NST - 1)) * STAGE_BYTES;
      const char* const vs = smem + ti * STAGE_BYTES + TILE_BYTES;
      auto kfrag = [&](int g) -> bf16x8 {              // MFMA g of S(t+1): k-step ds = g >> 1, key block kb = g & 1
        return *reinterpret_cast<const bf16x8*>(kst + (g & 1) * 8192 + (ka ^ (unsigned)((g >> 1) << 5)));
      };
      auto vfrag = [&](int j) -> bf16x8 {              // MFMA j of O += V P: 16-key group kk = j >> 2, d block d0 = j & 3
        const char* vp = vs + (j >> 2) * 16 * 256 + (va0_ ^ (unsigned)((j & 3) << 6));
        const bf16x4 va = lds_read_tr16(vp);
        const bf16x4 vb = lds_read_tr16(vp + 8 * 256);
        bf16x8 vf;
        vf[0] = va[0]; vf[1] = va[1]; vf[2] = va[2]; vf[3] = va[3];
        vf[4] = vb[0]; vf[5] = vb[1]; vf[6] = vb[2]; vf[7] = vb[3];
        return vf;
      };
      // the rare re-base of the lazy max for key block kb_ (attn7.hip's rule): O and l are scaled, the block's P recomputed
#define Q_REBASE(kb_)                                                                   \
      {                                                                                 \
        float mloc = st[kb_][0];                                                        \
        _Pragma("unroll") for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[kb_][r]);  \
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));                                   \
        const float m_new = fmaxf(m_run, mloc);                                         \
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);                      \
        m_run = m_new;                                                                  \
        m_base = m_new;                                                                 \
        l_run = (l_run + psum) * alpha;                                                 \
        psum = 0.f;                                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                   \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;             \
        ps = 0.f;                                                                       \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                \
          const float pv = __builtin_amdgcn_exp2f(st[kb_][r] - m_new);                  \
          ps += pv;                                                                     \
          pf[kb_][r >> 3][r & 7] = (__bf16)pv;                                          \
        }                                                                               \
      }
      bf16x8 pf[2][2];
      float psum = 0.f, ps = 0.f;
      bf16x8 kf0 = kfrag(0), kf1 = kf0;
      if (PFD == 2) kf1 = kfrag(1);
      __builtin_amdgcn_s_setprio(1);
      // ---- block A: the first half of S(t+1) (k-steps 0..3) with key block 0's softmax behind its 8 MFMAs ----
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        bf16x8 nk = kf0;
        nk = kfrag(g + PFD);
        sn[g & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[g >> 1], g < 2 ? zero16 : sn[g & 1], 0, 0, 0);
        const int r0 = g * 2;
        const float pv0 = __builtin_amdgcn_exp2f(st[0][r0] - m_base);
        const float pv1 = __builtin_amdgcn_exp2f(st[0][r0 + 1] - m_base);
        ps += pv0;
        ps += pv1;
        pf[0][r0 >> 3][r0 & 7] = (__bf16)pv0;
        pf[0][r0 >> 3][(r0 & 7) + 1] = (__bf16)pv1;
        if (PFD == 2) { kf0 = kf1; kf1 = nk; } else kf0 = nk;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (__any(!(ps <= p_lim) || no_ref)) Q_REBASE(0);
      psum += ps;
      ps = 0.f;
      // ---- block B: the second half of S(t+1) alternating with O += V(t) P(t) of key block 0 (16 MFMAs), key block 1's softmax behind them ----
      bf16x8 vf0 = vfrag(0);
#pragma unroll
      for (int sstep = 0; sstep < 16; ++sstep) {
        if ((sstep & 1) == 0) {
          const int g = 8 + (sstep >> 1);
          bf16x8 nk = kf0;
          if (g + PFD < 16) nk = kfrag(g + PFD);
          sn[g & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[g >> 1], sn[g & 1], 0, 0, 0);
          if (PFD == 2) { kf0 = kf1; kf1 = nk; } else kf0 = nk;
        } else {
          const int j = sstep >> 1;                    // 0..7: kk = 0, 1
          const bf16x8 nv = vfrag(j + 1);              // j + 1 = 8 is the first fragment of block C
          ot[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, pf[0][j >> 2], ot[j & 3], 0, 0, 0);
          vf0 = nv;
        }
        const float pv = __builtin_amdgcn_exp2f(st[1][sstep] - m_base);
        ps += pv;
        pf[1][sstep >> 3][sstep & 7] = (__bf16)pv;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (__any(!(ps <= p_lim) || no_ref)) Q_REBASE(1);
      psum += ps;
      // ---- block C: O += V(t) P(t) of key block 1 ----
#pragma unroll
      for (int j = 8; j < 16; ++j) {
        bf16x8 nv = vf0;
        if (j + 1 < 16) nv = vfrag(j + 1);
        ot[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, pf[1][(j >> 2) & 1], ot[j & 3], 0, 0, 0);
        vf0 = nv;
      }
      __builtin_amdgcn_s_setprio(0);
#undef Q_REBASE
      l_run += psum;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      Q_BARRIER();
    }
  }

  attc::store_result(p, q0 + l31, head, hi, ot, m_run, l_run);
  if (p.trace && tid == 0 && (int)blockIdx.x < p.trace_cap) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* tr = p.trace + (size_t)blockIdx.x * 4;
    tr[0] = t_start; tr[1] = __builtin_amdgcn_s_memrealtime(); tr[2] = hwid; tr[3] = xcc;
  }
#undef Q_DMA_TILE
#undef Q_BARRIER
}

template <int PFD>
int launch(const attc::Params& p, hipStream_t st) {
  static icv_dev_flags attr_set = {};
  if (int rc = icv_ensure_dynamic_lds((const void*)attn7q_kernel<PFD>, LDS_BYTES, &attr_set, "attn7q")) return rc;
  const int64_t nwg = (int64_t)p.heads * p.nqb;
  hipLaunchKernelGGL((attn7q_kernel<PFD>), dim3((unsigned)nwg), dim3(NW * 64), LDS_BYTES, st, p);
  return icv_check_launch("icv_attention_fwd");
}

}  // namespace att7q

// the plain launch / one carried-state chunk at unit scale; returns -1 when the problem is not its (the caller falls back to attn7p.hip)
int icv_attn7q_single(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, float* acc,
                      int64_t ldacc, float* ml, int state_in, int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int variant,
                      hipStream_t st) {
  attc::Params p;
  attc::fill_params(p, q, ldq, k, ldk, v, ldv, o, ldo, acc, ldacc, ml, state_in, state_out, Sq, Skv, heads, scale, 256);
  if (p.sc != 1.0f || !icv_get_option_int("attn_unit_scale", 1)) return -1;
  if (!(Skv > 0 && Skv < (1LL << 31) / 64 && 64 * ldk * 2 < (1LL << 32) && 64 * ldv * 2 < (1LL << 32))) return -1;
  p.trace = icv_attention_trace_buffer(&p.trace_cap);
  return variant == 1 ? att7q::launch<1>(p, st) : att7q::launch<2>(p, st);
}
