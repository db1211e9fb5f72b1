// Flash attention forward: experiments/attn4.hip's LDS-DMA ring (no staging VGPRs, counted vmcnt, one barrier per 64-key tile)
// with the vector-side diet of attn2.hip, which the freed registers make affordable:
//   * lazy max: P = exp2(S - m_ref) is taken against the current reference and the 16-key partial row sum (needed
//     anyway) bounds every P; only when it exceeds 2^thr is the block's true max taken, O/l rescaled, P recomputed;
//   * unit scale (scale * log2 e == 1: the DiT folds the softmax scale into the K RMSNorm weight): the reference
//     lives in a persistent 16-register vector cinit = -m_ref that is the C operand of the first MFMA of each
//     S^T chain, so there is neither a per-element fma nor a per-tile accumulator initialisation;
//   * DMA addresses: tile base in an SGPR pair + four fixed 32-bit per-lane byte offsets (saddr form of
//     global_load_lds_dwordx4); only a partial last tile takes the clamped 64-bit path.
// Everything else (fragments, swizzled LDS images, ring invariants, stagger option) is experiments/attn4.hip's: see there.
#include "attn_common.h"

namespace att7 {

using attc::D;
using attc::NEG_BIG;
using attc::Params;
using attc::lds_read_tr16;
constexpr int KVB = 64;
constexpr int TILE_BYTES = KVB * D * 2;      // 16 KiB (K or V)
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // 32 KiB

// LDS-DMA through inline asm: hipcc does not count it, so it never guards the (alias-info-free)
// ds_read_b64_tr_b16 reads with vmcnt(0); completion is tracked by our own counted s_waitcnt vmcnt.
// M0 (the DMA's LDS base) is compiler-reserved: it is saved, set and restored inside ONE statement
// (cdna guide §5.7).  lds_dst must be wave-uniform; the hardware adds lane*16.
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

// saddr form: 64-bit wave-uniform base in SGPRs + 32-bit per-lane byte offset
__device__ __forceinline__ void dma16s(const void* base, unsigned off, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(off), "s"(base), "s"(lds_dst)
      : "memory");
}

#ifndef ATTN7_SHORT_DEFAULT
#define ATTN7_SHORT_DEFAULT 1024
#endif

// VAR bit flags: 1 = stagger wave groups, 2 = issue all 16 K-fragment reads ahead of the QK^T MFMAs,
//                4 = s_setprio(1) around MFMA clusters, 8 = K/V DMA three tiles ahead instead of two (not with 1),
//                16 = unit scale, 32 = QK^T MFMAs in key-block-major order (round 1's; the default is d-step major)
// (tried and dropped: row sums with v_pk_add_f32 — 8 fewer VALU issues per 32-key block, 1142.6 vs 1149.5 TF/s, noise:
//  the kernel is power-limited, profiles/r02/power_limit_probes.md)
// Geometry (template NW, NST): NW waves x 32 query rows per block, NST ring stages of one 64-key K + V tile (32 KiB).
//   <8, 4> (default, self-attention): 256 rows, 128 KiB ring, one block per CU, counted vmcnt (DMA two tiles ahead);
//   <4, 2> (short key sequences, i.e. the 512 / 257-key cross-attention, option "attn7_short"): 128 rows, 64 KiB, so TWO
//          blocks share a CU and one block's prologue (Q load, ring fill) and epilogue (O store) - half of a launch that
//          has only 8 key tiles per row - run under the other's MFMAs; DMA one interval ahead, vmcnt(0) + barrier per tile.
template <int VAR, int NW, int NST>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2))) void attn7_kernel(Params p) {
  constexpr bool STAGGER = VAR & 1, KPREFETCH = VAR & 2, SETPRIO = VAR & 4, DEEP = VAR & 8, UNIT = VAR & 16;
  // QK^T MFMA order: d-step major (consecutive MFMAs share the Q fragment: less operand toggling, +0.6...1.4 % under the power
  // cap, profiles/r02/attention_variants.md) unless bit 32 asks for round 1's key-block-major order
  constexpr bool DSMAJOR = !(VAR & 32);
  // bit 128: 128-key publish granularity - the ring is used as two halves of two 64-key tiles; an interval computes tiles
  // (t, t+1) from one half while the DMA of (t+2, t+3) fills the other, ONE vmcnt(0) + barrier per 128 keys instead of a
  // counted wait + barrier per 64 (round 3's bounded attempt at the per-tile barrier cost; profiles/r03/attention_variants.md)
  constexpr bool PAIR = (VAR & 128) != 0;
  static_assert(!(PAIR && (STAGGER || DEEP)), "the paired schedule has its own DMA distance");
  static_assert(!(STAGGER && DEEP), "the staggered group already runs its DMA three tiles ahead");   // 16: unit scale (set by the dispatcher)
  static_assert(NST == 4 || !(PAIR || STAGGER || DEEP), "the two-stage ring has one schedule");
  static_assert(NW == 8 || !STAGGER, "the stagger pairs waves 0-3 with 4-7");
  constexpr int NI = 16 / NW;                    // DMA instructions per wave per K (or V) tile: 16 KiB = 16 x 1 KiB
  constexpr int QB = NW * 32;
  const float p_lim = __builtin_amdgcn_exp2f(p.thr);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int l31 = lane & 31;

  int head, qb;
  attc::work_item(p, head, qb);
  const int64_t q0 = (int64_t)qb * QB + wave * 32;
  const unsigned long long t_start = p.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;

  const bf16_t* qh = p.q + (int64_t)head * D;
  const bf16_t* kh = p.k + (int64_t)head * D;
  const bf16_t* vh = p.v + (int64_t)head * D;

  int64_t qr_c = q0 + l31;
  qr_c = qr_c < p.Sq ? qr_c : p.Sq - 1;

  // ---- softmax state ----
  f32x16 ot[4];
  float m_run, l_run;
  attc::load_state(p, qr_c, head, hi, ot, m_run, l_run);
  float m_base = m_run < -1.0e29f ? 0.f : m_run;   // UNIT: the reference currently baked into cinit
  f32x16 cinit;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; ++r) cinit[r] = UNIT ? -m_base : 0.f;

  bf16x8 qf[8];
#define A7_LOAD_Q()                                                                                   \
  {                                                                                                   \
    const bf16_t* qp = qh + qr_c * p.ldq + hi * 8;                                                    \
    _Pragma("unroll") for (int ds = 0; ds < 8; ++ds) qf[ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16); \
  }
  if (NST != 2) {
    A7_LOAD_Q();
    // Retire the ordinary (VGPR-destination) prologue loads before any LDS-DMA is in flight: beside a
    // DMA hipcc waits vmcnt(0) for every ordinary load, which would drain the ring (guide §5 trap (b)).
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }

  // ---- LDS-DMA lane mapping: instruction j of this wave covers keys (wave*NI + j)*4 + lane/16 ----
  const int pc = lane & 15;
  const int nt = (int)((p.Skv + KVB - 1) / KVB);
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  int dkey[NI], kcol[NI], vcol[NI];              // key row inside the tile, swizzled source column (elements)
  unsigned ko[NI], vo[NI];                       // per-lane byte offsets from the tile base (saddr form)
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    dkey[j] = (wave * NI + j) * 4 + (lane >> 4);
    kcol[j] = (pc ^ (dkey[j] & 15)) * 8;
    vcol[j] = (pc ^ ((dkey[j] & 3) << 2)) * 8;
    ko[j] = (unsigned)(((int64_t)dkey[j] * p.ldk + kcol[j]) * 2);
    vo[j] = (unsigned)(((int64_t)dkey[j] * p.ldv + vcol[j]) * 2);
  }
  // tile T_ (clamped to the last one: requests past the end re-read it into a dead stage, keeping the waits uniform)
  // -> stage T_ % NST.  (A running-pointer form of the two 64-bit tile-base multiplies measured 1 % slower in round 2.)
#define A7_DMA_TILE(T_)                                                                              \
  {                                                                                                  \
    const int tt_ = (T_) < nt ? (T_) : nt - 1;                                                       \
    const unsigned l0_ = lds_base + (unsigned)(((T_) & (NST - 1)) * STAGE_BYTES + (wave * NI) * 1024); \
    if ((int64_t)(tt_ + 1) * KVB <= p.Skv) {                                                         \
      const bf16_t* kt_ = kh + (int64_t)tt_ * KVB * p.ldk;                                           \
      const bf16_t* vt_ = vh + (int64_t)tt_ * KVB * p.ldv;                                           \
      _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_) dma16s(kt_, ko[j_], l0_ + j_ * 1024);        \
      _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_) dma16s(vt_, vo[j_], l0_ + TILE_BYTES + j_ * 1024); \
    } else {                                                                                         \
      _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_) {                                            \
        int64_t r_ = (int64_t)tt_ * KVB + dkey[j_];                                                  \
        r_ = r_ < p.Skv ? r_ : p.Skv - 1;                                                            \
        dma16(kh + r_ * p.ldk + kcol[j_], l0_ + j_ * 1024);                                          \
        dma16(vh + r_ * p.ldv + vcol[j_], l0_ + TILE_BYTES + j_ * 1024);                             \
      }                                                                                              \
    }                                                                                                \
  }
// counted wait (8 waves: 4 DMA instructions per wave and tile): the youngest tile - DEEP: the two youngest - may still be in flight
#define A7_VMCNT4()                                              \
  do {                                                           \
    if (DEEP) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   \
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        \
  } while (0)
#define A7_BARRIER()                                          \
  do {                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                             \
    asm volatile("" ::: "memory");                            \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)

  const int grp = STAGGER ? (wave >> 2) : 0;

  // ---- prologue: tiles 0 and 1 in flight; tile 0 landed + published ----
  A7_DMA_TILE(0);
  A7_DMA_TILE(1);
  if (DEEP) A7_DMA_TILE(2);
  if (NST == 2) {
    // short rows (8 key tiles): the Q fragments are requested AFTER the first two tiles' DMA and one vmcnt(0) retires all
    // three (vmcnt is in order), so the Q latency and the ring fill overlap instead of adding up - a visible share of a
    // block that lives for 8 tiles; both ring stages are then full, which the loop's first wait expects anyway
    A7_LOAD_Q();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (PAIR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // both tiles of the first half
  else A7_VMCNT4();
  A7_BARRIER();
  if (grp == 1) {   // group 1's idle interval I_0: it still owes its DMA duties (issue tile 2, retire tile 1)
    A7_DMA_TILE(2);
    A7_VMCNT4();
    A7_BARRIER();
  }

  const int k_row_off = l31 * 256;
  const int k_sw = l31 & 15;
  const int g = lane >> 4, t16 = lane & 15;
  const int v_key_lo = 4 * hi + (t16 >> 2);
  const int v_byte_lo = (g & 1) * 32 + (t16 & 3) * 8;
  const int v_sw = (t16 >> 2) << 6;
  const int ahead = 2 + grp + (DEEP ? 1 : 0);   // group 1 runs one tile behind, so its DMA duties are one tile further ahead

  // one 64-key tile: K reads -> 16 QK^T MFMAs -> softmax -> V tr-reads -> 16 PV MFMAs (no DMA, no waits)
  auto tile = [&](const int t) __attribute__((always_inline)) {
    const char* ks = smem + (t & (NST - 1)) * STAGE_BYTES;
    const char* vs = ks + TILE_BYTES;
    const int64_t key0 = (int64_t)t * KVB;

    f32x16 st[2];
    const bool no_ref = UNIT && m_run < -1.0e29f;
    if (KPREFETCH) {
      // hipcc otherwise recycles ONE register quad for consecutive K fragments (read -> lgkmcnt(0) ->
      // MFMA -> read ...), exposing the LDS latency 16 times per tile: read everything first.
      bf16x8 kf[2][8];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ds = 0; ds < 8; ++ds) {
          const int c = ds * 2 + hi;
          kf[kb][ds] = *reinterpret_cast<const bf16x8*>(ks + kb * 8192 + k_row_off + ((c ^ k_sw) << 4));
        }
      __builtin_amdgcn_sched_barrier(0);
      if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ds = 0; ds < 8; ++ds)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][ds], qf[ds], ds == 0 ? (UNIT ? cinit : zero16) : st[kb], 0, 0, 0);
      if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    } else if (DSMAJOR) {
#pragma unroll
      for (int ds = 0; ds < 8; ++ds)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const int c = ds * 2 + hi;
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + kb * 8192 + k_row_off + ((c ^ k_sw) << 4));
          st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], ds == 0 ? (UNIT ? cinit : zero16) : st[kb], 0, 0, 0);
        }
    } else {
      if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ds = 0; ds < 8; ++ds) {
          const int c = ds * 2 + hi;
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + kb * 8192 + k_row_off + ((c ^ k_sw) << 4));
          st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], ds == 0 ? (UNIT ? cinit : zero16) : st[kb], 0, 0, 0);
        }
      if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    }
    if (key0 + KVB > p.Skv) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= p.Skv) st[kb][r] = NEG_BIG;
        }
    }
    float mb = -m_run * p.sc;
    float psum = 0.f;
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 pf[2];
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = UNIT ? __builtin_amdgcn_exp2f(st[kb][r]) : __builtin_amdgcn_exp2f(fmaf(st[kb][r], p.sc, mb));
        ps += pv;
        pf[r >> 3][r & 7] = (__bf16)pv;
      }
      // lazy max (see attn2.hip): the partial row sum bounds every P of the block
      if (__any(!(ps <= p_lim) || no_ref)) {
        float mloc = st[kb][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[kb][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        if (UNIT) mloc += m_base;                        // st = s - m_base
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);
        m_run = m_new;
        l_run = (l_run + psum) * alpha;
        psum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
        mb = -m_run * p.sc;
        if (UNIT) {                                      // re-base this and the later block of the tile, and cinit
          const float dm = m_new - m_base;
          m_base = m_new;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (j >= kb) {
#pragma unroll
              for (int r = 0; r < 16; ++r) st[j][r] -= dm;
            }
#pragma unroll
          for (int r = 0; r < 16; ++r) cinit[r] = -m_new;
        }
        ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = UNIT ? __builtin_amdgcn_exp2f(st[kb][r]) : __builtin_amdgcn_exp2f(fmaf(st[kb][r], p.sc, mb));
          ps += pv;
          pf[r >> 3][r & 7] = (__bf16)pv;
        }
      }
      psum += ps;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int kk = kb * 2 + hf;
#pragma unroll
        for (int d0 = 0; d0 < 4; ++d0) {
          const int key_l = kk * 16 + v_key_lo;
          const int byte = (d0 * 64 + v_byte_lo) ^ v_sw;
          const bf16x4 va = lds_read_tr16(vs + key_l * 256 + byte);
          const bf16x4 vb = lds_read_tr16(vs + (key_l + 8) * 256 + byte);
          bf16x8 vf;
          vf[0] = va[0]; vf[1] = va[1]; vf[2] = va[2]; vf[3] = va[3];
          vf[4] = vb[0]; vf[5] = vb[1]; vf[6] = vb[2]; vf[7] = vb[3];
          ot[d0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[hf], ot[d0], 0, 0, 0);
        }
      }
    }
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    l_run += psum;
  };

  if (NST == 2) {
    // two-stage ring: tile t+1 was requested one interval ago; its stage is re-requested (tile t+2 -> stage t % 2) right
    // after the barrier that ends the last read of tile t
    for (int t = 0; t < nt; ++t) {
      tile(t);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      A7_BARRIER();
      A7_DMA_TILE(t + 2);
    }
  } else if (PAIR) {
    for (int t = 0; t < nt; t += 2) {
      if (!(p.ablate & 1)) {           // the other half of the ring: last read in the previous interval, before its barrier
        A7_DMA_TILE(t + 2);
        A7_DMA_TILE(t + 3);
      }
      tile(t);
      if (t + 1 < nt) tile(t + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // issued a whole 128-key interval ago
      if (!(p.ablate & 2)) A7_BARRIER();
    }
  } else {
    for (int t = 0; t < nt; ++t) {
      // DMA of tile t+ahead first (longest possible flight), counted wait at the end of the interval
      if (!(p.ablate & 1)) A7_DMA_TILE(t + ahead);
      tile(t);
      A7_VMCNT4();    // this wave's share of tile t+ahead-1 has landed (tile t+ahead may still be in flight)
      if (!(p.ablate & 2)) A7_BARRIER();   // ... and is published to the block
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail DMAs before the LDS is released
  if (grp == 0 && STAGGER) A7_BARRIER();             // re-balance the stagger

  attc::store_result(p, q0 + l31, head, hi, ot, m_run, l_run);
  if (p.trace && tid == 0 && (int)blockIdx.x < p.trace_cap) {      // diagnostics: where and when did this work-group run
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* t = p.trace + (size_t)blockIdx.x * 4;
    t[0] = t_start; t[1] = __builtin_amdgcn_s_memrealtime(); t[2] = hwid; t[3] = xcc;
  }
}

template <int VAR, int NW = 8, int NST = 4>
int launch(const Params& p, hipStream_t st) {
  constexpr int LDS_BYTES = NST * STAGE_BYTES;   // 128 KiB (one block per CU) or 64 KiB (two)
  static icv_dev_flags attr_set = {};
  if (int rc = icv_ensure_dynamic_lds((const void*)attn7_kernel<VAR, NW, NST>, LDS_BYTES, &attr_set, "attn7")) return rc;
  const int64_t nwg = (int64_t)p.heads * p.nqb;
  hipLaunchKernelGGL((attn7_kernel<VAR, NW, NST>), dim3((unsigned)nwg), dim3(NW * 64), LDS_BYTES, st, p);
  return icv_check_launch("icv_attention(7)");
}

}  // namespace att7

int icv_attn7_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st) {
  att7::Params p;
  // short key sequences (cross-attention: 512 text / 257 image keys = 8 / 5 tiles per row): 4-wave blocks on a two-stage ring,
  // two per CU (see the kernel header); "attn7_short" = largest Skv that takes this shape (0 = never)
  int short_max = icv_get_option_int("attn7_short", -1);
  if (short_max < 0) short_max = ATTN7_SHORT_DEFAULT;            // negative = the built-in default
  const bool short_kv = Skv <= short_max;
  attc::fill_params(p, q, ldq, k, ldk, v, ldv, o, ldo, acc, ldacc, ml, state_in, state_out, Sq, Skv, heads, scale, short_kv ? 128 : 256);
  if (p.sc == 1.0f && icv_get_option_int("attn_unit_scale", 1)) var |= 16;
  p.ablate = icv_get_option_int("attn7_ablate", 0);   // timing experiments only (tools/attn_bench.py)
  p.trace = icv_attention_trace_buffer(&p.trace_cap);
  if (short_kv) return (var & 16) ? att7::launch<16, 4, 2>(p, st) : att7::launch<0, 4, 2>(p, st);
  switch (var) {
    case 0: return att7::launch<0>(p, st);
    case 1: return att7::launch<1>(p, st);
    case 4: return att7::launch<4>(p, st);
    case 5: return att7::launch<5>(p, st);
    case 6: return att7::launch<6>(p, st);
    case 7: return att7::launch<7>(p, st);
    case 8: return att7::launch<8>(p, st);
    case 24: return att7::launch<24>(p, st);
    case 32: return att7::launch<32>(p, st);
    case 48: return att7::launch<48>(p, st);
    case 128: return att7::launch<128>(p, st);
    case 144: return att7::launch<144>(p, st);
    case 132: return att7::launch<132>(p, st);
    case 148: return att7::launch<148>(p, st);
    case 16: return att7::launch<16>(p, st);
    case 17: return att7::launch<17>(p, st);
    case 20: return att7::launch<20>(p, st);
    case 21: return att7::launch<21>(p, st);
    case 22: return att7::launch<22>(p, st);
    case 23: return att7::launch<23>(p, st);
  }
  icv_set_error("icv_attention_fwd: unknown attn7 variant %d", var);
  return 1;
}
