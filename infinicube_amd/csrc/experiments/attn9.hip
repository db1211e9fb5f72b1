// EXPERIMENT (icv_set_option("attn_kernel", 9); built only with ICV_EXPERIMENTS=1).  MEASURED: ties attn7 — 14B self-attention
// 1183 vs 1186 TF/s, 1.3B 1162-1175 vs 1183 (same box, interleaved rounds, profiles/r02/attention_variants.md): the kernel
// is power-limited, not schedule-limited (attn7: MFMA busy 58 % at an EFFECTIVE clock of 1.97 GHz; a denser schedule buys
// a lower clock).  Kept as the record of the "overlap the softmax with the next QK^T" attempt.
// Flash attention forward, software-pipelined across key tiles ("compute[next] || finish[cur]"):
// attn7.hip's structure (LDS-DMA ring with counted vmcnt, one barrier per 64-key tile, swapped QK^T so the softmax is
// lane-local, lazy max, unit scale with the reference in the MFMA C operand) with ONE change of schedule:
//   the work is cut in 32-key blocks and the 8 QK^T MFMAs of block h+1 (into a second score vector) sit in the same
//   basic block as the exp2 / row-sum / bf16-pack VALU of block h, so the matrix pipe runs while the vector pipe does
//   the softmax of the previous block, instead of QK^T -> softmax -> PV one after the other (rocprofv3 on attn7: MFMA busy 58 % at the effective
//   clock, i.e. the pipe idles through the vector phase even with two waves per SIMD - they run in step between the
//   per-tile barriers).  The lazy-max test moves behind the whole tile's exps (one rare branch per tile, not per
//   32-key block) so that block stays a single basic block.
// Ring: tile t+1's K must have landed when iteration t starts, so the DMA runs 3 tiles ahead (4 stages: V(t), K(t+1),
// and tiles t+2, t+3 in flight); the counted wait is still vmcnt(4) (= the youngest tile's 4 DMA instructions).
#include "attn_common.h"

namespace att9 {

using attc::D;
using attc::NEG_BIG;
using attc::Params;
using attc::lds_read_tr16;
constexpr int KVB = 64;
constexpr int QB = 256;
constexpr int TILE_BYTES = KVB * D * 2;      // 16 KiB (K or V)
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // 32 KiB
constexpr int NSTAGE = 4;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;  // 128 KiB

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

// v_add_f32 through inline asm: the compiler can move it but cannot fuse two of them into a packed add
__device__ __forceinline__ float add_f32(float a, float b) {
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ float tree_sum16(const float (&e)[16]) {
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = add_f32(e[2 * i], e[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = add_f32(s[2 * i], s[2 * i + 1]);
  return add_f32(add_f32(s[0], s[1]), add_f32(s[2], s[3]));
}

// VAR bit flags: 4 = s_setprio(1) around the MFMA-carrying blocks, 16 = unit scale (set by the dispatcher).
// (A variant with sched_group_barrier interleave hints was 2-4 % slower and failed one parity case; removed.)
template <int VAR>
__global__ __launch_bounds__(512) void attn9_kernel(Params p) {
  constexpr bool SETPRIO = VAR & 4, UNIT = VAR & 16;
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

  const bf16_t* qh = p.q + (int64_t)head * D;
  const bf16_t* kh = p.k + (int64_t)head * D;
  const bf16_t* vh = p.v + (int64_t)head * D;

  int64_t qr_c = q0 + l31;
  qr_c = qr_c < p.Sq ? qr_c : p.Sq - 1;

  f32x16 ot[4];
  float m_run, l_run;
  attc::load_state(p, qr_c, head, hi, ot, m_run, l_run);
  float m_base = m_run < -1.0e29f ? 0.f : m_run;   // UNIT: the reference currently baked into cinit
  f32x16 cinit;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; ++r) cinit[r] = UNIT ? -m_base : 0.f;

  bf16x8 qf[8];
  {
    const bf16_t* qp = qh + qr_c * p.ldq + hi * 8;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) qf[ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // retire the ordinary prologue loads before any LDS-DMA is in flight

  const int dkey0 = (wave * 2 + 0) * 4 + (lane >> 4);
  const int dkey1 = (wave * 2 + 1) * 4 + (lane >> 4);
  const int pc = lane & 15;
  const int kcol0 = (pc ^ (dkey0 & 15)) * 8, kcol1 = (pc ^ (dkey1 & 15)) * 8;
  const int vcol0 = (pc ^ ((dkey0 & 3) << 2)) * 8, vcol1 = (pc ^ ((dkey1 & 3) << 2)) * 8;
  const int nt = (int)((p.Skv + KVB - 1) / KVB);
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const unsigned ko0 = (unsigned)(((int64_t)dkey0 * p.ldk + kcol0) * 2), ko1 = (unsigned)(((int64_t)dkey1 * p.ldk + kcol1) * 2);
  const unsigned vo0 = (unsigned)(((int64_t)dkey0 * p.ldv + vcol0) * 2), vo1 = (unsigned)(((int64_t)dkey1 * p.ldv + vcol1) * 2);
#define A9_DMA_TILE(T_)                                                                              \
  {                                                                                                  \
    const int tt_ = (T_) < nt ? (T_) : nt - 1;                                                       \
    const unsigned l0_ = lds_base + (unsigned)(((T_) & (NSTAGE - 1)) * STAGE_BYTES + (wave * 2) * 1024); \
    if ((int64_t)(tt_ + 1) * KVB <= p.Skv) {                                                         \
      const bf16_t* kt_ = kh + (int64_t)tt_ * KVB * p.ldk;                                           \
      const bf16_t* vt_ = vh + (int64_t)tt_ * KVB * p.ldv;                                           \
      dma16s(kt_, ko0, l0_);                                                                         \
      dma16s(kt_, ko1, l0_ + 1024);                                                                  \
      dma16s(vt_, vo0, l0_ + TILE_BYTES);                                                            \
      dma16s(vt_, vo1, l0_ + TILE_BYTES + 1024);                                                     \
    } else {                                                                                         \
      int64_t r0_ = (int64_t)tt_ * KVB + dkey0, r1_ = (int64_t)tt_ * KVB + dkey1;                    \
      r0_ = r0_ < p.Skv ? r0_ : p.Skv - 1;                                                           \
      r1_ = r1_ < p.Skv ? r1_ : p.Skv - 1;                                                           \
      dma16(kh + r0_ * p.ldk + kcol0, l0_);                                                          \
      dma16(kh + r1_ * p.ldk + kcol1, l0_ + 1024);                                                   \
      dma16(vh + r0_ * p.ldv + vcol0, l0_ + TILE_BYTES);                                             \
      dma16(vh + r1_ * p.ldv + vcol1, l0_ + TILE_BYTES + 1024);                                      \
    }                                                                                                \
  }
#define A9_VMCNT4() asm volatile("s_waitcnt vmcnt(4)" ::: "memory")
#define A9_BARRIER()                                          \
  do {                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                             \
    asm volatile("" ::: "memory");                            \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)

  const int k_row_off = l31 * 256;
  const int k_sw = l31 & 15;
  const int g = lane >> 4, t16 = lane & 15;
  const int v_key_lo = 4 * hi + (t16 >> 2);
  const int v_byte_lo = (g & 1) * 32 + (t16 & 3) * 8;
  const int v_sw = (t16 >> 2) << 6;

  // S^T of one 32-key block: st[r] = score(key = kb*32 + (r&3) + 8*(r>>2) + 4*hi, query = lane&31) - reference (UNIT)
#define A9_QK(KS_, KB_, ST_)                                                                                      \
  _Pragma("unroll") for (int ds = 0; ds < 8; ++ds) {                                                              \
    const int c_ = ds * 2 + hi;                                                                                   \
    const bf16x8 kf_ = *reinterpret_cast<const bf16x8*>((KS_) + (KB_) * 8192 + k_row_off + ((c_ ^ k_sw) << 4));   \
    ST_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf_, qf[ds], ds == 0 ? (UNIT ? cinit : zero16) : ST_, 0, 0, 0);  \
  }

  // ---- prologue: tiles 0, 1, 2 in flight; tiles 0 and 1 landed + published; scores of (tile 0, block 0) ----
  A9_DMA_TILE(0);
  A9_DMA_TILE(1);
  A9_DMA_TILE(2);
  A9_VMCNT4();
  A9_BARRIER();
  f32x16 sa, sb;
  A9_QK(smem, 0, sa)

  // One half-step: SC = scores of block KB_ of tile T_ (complete); SN receives the scores of the NEXT block
  // (block KBN_ of the K tile at KSN_), whose 8 MFMAs share a basic block with the softmax VALU of SC.
#define A9_HALF(T_, KB_, SC, SN, KSN_, KBN_)                                                                      \
  {                                                                                                               \
    const int64_t key0_ = (int64_t)(T_) * KVB + (KB_) * 32;                                                       \
    const char* vs_ = smem + ((T_) & (NSTAGE - 1)) * STAGE_BYTES + TILE_BYTES;                                    \
    if (key0_ + 32 > p.Skv) {                                                                                     \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                            \
        const int64_t key = key0_ + (r & 3) + 8 * (r >> 2) + 4 * hi;                                              \
        if (key >= p.Skv) SC[r] = NEG_BIG;                                                                        \
      }                                                                                                           \
    }                                                                                                             \
    const bool no_ref_ = UNIT && m_run < -1.0e29f;                                                                \
    float mb_ = -m_run * p.sc;                                                                                    \
    bf16x8 pf_[2];                                                                                                \
    float ps_;                                                                                                    \
    /* ---- block 1: 8 QK^T MFMAs of the next block on the matrix pipe; exp2 / row sum / bf16 pack of this block */ \
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);                                                                   \
    A9_QK(KSN_, KBN_, SN)                                                                                         \
    {                                                                                                             \
      /* the row sum as a 4-level tree of plain v_add_f32: a serial += chain is 16 dependent adds, and hipcc fuses  \
         neighbouring chains into v_pk_add_f32, an anti-lever beside MFMAs (MI355X guide, per-instruction constants) */ \
      float e_[16];                                                                                               \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                            \
        e_[r] = UNIT ? __builtin_amdgcn_exp2f(SC[r]) : __builtin_amdgcn_exp2f(fmaf(SC[r], p.sc, mb_));            \
        pf_[r >> 3][r & 7] = (__bf16)e_[r];                                                                       \
      }                                                                                                           \
      ps_ = tree_sum16(e_);                                                                                       \
      /* pin the packed P fragments in this block: hipcc otherwise sinks the v_cvt_pk below the rare branch, in     \
         front of the PV MFMAs that wait for them */                                                               \
      asm volatile("" :: "v"(pf_[0]), "v"(pf_[1]));                                                               \
    }                                                                                                             \
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);                                                                   \
    /* lazy max: the 16-key partial row sum bounds every P of the block; rare path = true max, rescale, re-base */ \
    if (__any(!(ps_ <= p_lim) || no_ref_)) {                                                                      \
      float mloc = SC[0];                                                                                         \
      _Pragma("unroll") for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, SC[r]);                                   \
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));                                                               \
      if (UNIT) mloc += m_base;                                                                                   \
      const float m_new = fmaxf(m_run, mloc);                                                                     \
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);                                         \
      m_run = m_new;                                                                                              \
      l_run *= alpha;                                                                                             \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;                                           \
      mb_ = -m_run * p.sc;                                                                                        \
      if (UNIT) { /* re-base this block, the already computed next block, and cinit */                            \
        const float dm = m_new - m_base;                                                                          \
        m_base = m_new;                                                                                           \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { SC[r] -= dm; SN[r] -= dm; }                              \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) cinit[r] = -m_new;                                         \
      }                                                                                                           \
      ps_ = 0.f;                                                                                                  \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                            \
        const float a_ = UNIT ? __builtin_amdgcn_exp2f(SC[r]) : __builtin_amdgcn_exp2f(fmaf(SC[r], p.sc, mb_));   \
        ps_ += a_;                                                                                                \
        pf_[r >> 3][r & 7] = (__bf16)a_;                                                                          \
      }                                                                                                           \
    }                                                                                                             \
    l_run += ps_;                                                                                                 \
    /* ---- block 2: O^T += V^T P^T of this block (8 MFMAs) */                                                    \
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                                            \
      const int kk = (KB_) * 2 + hf;                                                                              \
      _Pragma("unroll") for (int d0 = 0; d0 < 4; ++d0) {                                                          \
        const int key_l = kk * 16 + v_key_lo;                                                                     \
        const int byte = (d0 * 64 + v_byte_lo) ^ v_sw;                                                            \
        const bf16x4 va = lds_read_tr16(vs_ + key_l * 256 + byte);                                                \
        const bf16x4 vb = lds_read_tr16(vs_ + (key_l + 8) * 256 + byte);                                          \
        bf16x8 vf;                                                                                                \
        vf[0] = va[0]; vf[1] = va[1]; vf[2] = va[2]; vf[3] = va[3];                                               \
        vf[4] = vb[0]; vf[5] = vb[1]; vf[6] = vb[2]; vf[7] = vb[3];                                               \
        ot[d0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf_[hf], ot[d0], 0, 0, 0);                           \
      }                                                                                                           \
    }                                                                                                             \
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);                                                                   \
  }

  for (int t = 0; t < nt; ++t) {
    const char* ks_t = smem + (t & (NSTAGE - 1)) * STAGE_BYTES;
    const char* ks_n = smem + ((t + 1) & (NSTAGE - 1)) * STAGE_BYTES;   // past the end: the clamped (last) tile again, discarded
    A9_DMA_TILE(t + 3);
    A9_HALF(t, 0, sa, sb, ks_t, 1)
    A9_HALF(t, 1, sb, sa, ks_n, 0)
    A9_VMCNT4();    // this wave's share of tile t+2 has landed (tile t+3 may still be in flight)
    A9_BARRIER();   // ... and is published; everybody is done with V(t), K(t) and block 0 of K(t+1)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail DMAs before the LDS is released

  attc::store_result(p, q0 + l31, head, hi, ot, m_run, l_run);
}

template <int VAR>
int launch(const Params& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn9_kernel<VAR>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      icv_set_error("attn9: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  const int64_t nwg = (int64_t)p.heads * p.nqb;
  hipLaunchKernelGGL(attn9_kernel<VAR>, dim3((unsigned)nwg), dim3(512), LDS_BYTES, st, p);
  return icv_check_launch("icv_attention(9)");
}

}  // namespace att9

int icv_attn9_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st) {
  att9::Params p;
  attc::fill_params(p, q, ldq, k, ldk, v, ldv, o, ldo, acc, ldacc, ml, state_in, state_out, Sq, Skv, heads, scale, att9::QB);
  if (p.sc == 1.0f && icv_get_option_int("attn_unit_scale", 1)) var |= 16;
  switch (var) {
    case 0: return att9::launch<0>(p, st);
    case 4: return att9::launch<4>(p, st);
    case 16: return att9::launch<16>(p, st);
    case 20: return att9::launch<20>(p, st);
  }
  icv_set_error("icv_attention_fwd: unknown attn9 variant %d", var);
  return 1;
}
