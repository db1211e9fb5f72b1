// Flash attention forward, LDS-DMA staged.  Math, MFMA fragments, the
// key-permutation trick and the swizzled LDS images are those of attn.hip; what changes is how K/V
// reach the LDS.  A timing ablation of the register-staged kernels showed that HBM -> VGPR -> ds_write
// staging costs ~12 % (ds_write_b128 runs at ~1/3 of the LDS read rate and the staging VGPRs and
// vmcnt waits sit in the MFMA stream), so here
//   * K/V tiles (64 keys) are written by LDS-DMA (global_load_lds_dwordx4: 1 KiB = 4 keys per wave
//     instruction, 4 instructions per wave per tile) into a 4-stage ring (4 x 32 KiB); the DMA
//     destination is lane-linear, so the bank swizzles (K: chunk ^= key&15, V: chunk ^= (key&3)<<2)
//     are applied to the per-lane SOURCE address — same LDS images as before, no ds_write, no
//     staging VGPRs;
//   * DMA runs 2 tiles ahead with a COUNTED wait (vmcnt(4): the younger tile stays in flight across
//     the barrier), one barrier per tile, each interval = one complete 64-key tile
//     (K reads -> 16 QK^T MFMAs -> softmax -> V tr-reads -> 16 PV MFMAs);
//   * optional stagger: wave group 1 (waves 4-7) runs one tile behind group 0 so the two waves of a
//     SIMD are in different phases; its DMA duties are shifted by one tile so that every wave's share
//     of tile t+1 has landed one barrier before group 0 reads it.
//   Ring invariants (interval I_t: group 0 on tile t, group 1 on tile t-1): in I_t every wave issues the
//   DMA of tile t+2 (stage (t+2)&3, last read in I_{t-1} by group 1's tile t-2) and then waits
//   vmcnt(4) = its share of tile t+1 has landed; the barrier ending I_t publishes it; tile t+1 is first
//   read in I_{t+1}.  Tail tiles re-load the last tile (clamped) to keep the counted waits uniform.
//   * carried softmax state as in attn2.hip (key-axis chunking for the sequence-parallel path).
#include "attn_common.h"

namespace att4 {

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

// VAR bit flags: 1 = stagger wave groups, 2 = issue all 16 K-fragment reads ahead of the QK^T MFMAs,
//                4 = s_setprio(1) around MFMA clusters
template <int VAR>
__global__ __launch_bounds__(512) void attn4_kernel(Params p) {
  constexpr bool STAGGER = VAR & 1, KPREFETCH = VAR & 2, SETPRIO = VAR & 4;
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

  // ---- softmax state ----
  f32x16 ot[4];
  float m_run, l_run;
  attc::load_state(p, qr_c, head, hi, ot, m_run, l_run);

  bf16x8 qf[8];
  {
    const bf16_t* qp = qh + qr_c * p.ldq + hi * 8;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) qf[ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16);
  }
  // Retire the ordinary (VGPR-destination) prologue loads before any LDS-DMA is in flight: beside a
  // DMA hipcc waits vmcnt(0) for every ordinary load, which would drain the ring (guide §5 trap (b)).
  __builtin_amdgcn_s_waitcnt(0x0F70);

  // ---- LDS-DMA lane mapping: instruction j of this wave covers keys (wave*2 + j)*4 + lane/16 ----
  const int dkey0 = (wave * 2 + 0) * 4 + (lane >> 4);
  const int dkey1 = (wave * 2 + 1) * 4 + (lane >> 4);
  const int pc = lane & 15;
  const int kcol0 = (pc ^ (dkey0 & 15)) * 8, kcol1 = (pc ^ (dkey1 & 15)) * 8;        // elements
  const int vcol0 = (pc ^ ((dkey0 & 3) << 2)) * 8, vcol1 = (pc ^ ((dkey1 & 3) << 2)) * 8;
  const int nt = (int)((p.Skv + KVB - 1) / KVB);
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
#define A4_DMA_TILE(T_)                                                                              \
  {                                                                                                  \
    const int tt_ = (T_) < nt ? (T_) : nt - 1;                                                       \
    int64_t r0_ = (int64_t)tt_ * KVB + dkey0, r1_ = (int64_t)tt_ * KVB + dkey1;                      \
    r0_ = r0_ < p.Skv ? r0_ : p.Skv - 1;                                                             \
    r1_ = r1_ < p.Skv ? r1_ : p.Skv - 1;                                                             \
    const unsigned l0_ = lds_base + (unsigned)(((T_) & (NSTAGE - 1)) * STAGE_BYTES + (wave * 2) * 1024); \
    dma16(kh + r0_ * p.ldk + kcol0, l0_);                                                            \
    dma16(kh + r1_ * p.ldk + kcol1, l0_ + 1024);                                                     \
    dma16(vh + r0_ * p.ldv + vcol0, l0_ + TILE_BYTES);                                               \
    dma16(vh + r1_ * p.ldv + vcol1, l0_ + TILE_BYTES + 1024);                                        \
  }
#define A4_VMCNT4() asm volatile("s_waitcnt vmcnt(4)" ::: "memory")
#define A4_BARRIER()                                          \
  do {                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                             \
    asm volatile("" ::: "memory");                            \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)

  const int grp = STAGGER ? (wave >> 2) : 0;

  // ---- prologue: tiles 0 and 1 in flight; tile 0 landed + published ----
  A4_DMA_TILE(0);
  A4_DMA_TILE(1);
  A4_VMCNT4();
  A4_BARRIER();
  if (grp == 1) {   // group 1's idle interval I_0: it still owes its DMA duties (issue tile 2, retire tile 1)
    A4_DMA_TILE(2);
    A4_VMCNT4();
    A4_BARRIER();
  }

  const int k_row_off = l31 * 256;
  const int k_sw = l31 & 15;
  const int g = lane >> 4, t16 = lane & 15;
  const int v_key_lo = 4 * hi + (t16 >> 2);
  const int v_byte_lo = (g & 1) * 32 + (t16 & 3) * 8;
  const int v_sw = (t16 >> 2) << 6;
  const int ahead = 2 + grp;   // group 1 runs one tile behind, so its DMA duties are one tile further ahead

  for (int t = 0; t < nt; ++t) {
    const char* ks = smem + (t & (NSTAGE - 1)) * STAGE_BYTES;
    const char* vs = ks + TILE_BYTES;
    const int64_t key0 = (int64_t)t * KVB;

    // DMA of tile t+ahead first (longest possible flight), counted wait at the end of the interval
    A4_DMA_TILE(t + ahead);

    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
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
          st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][ds], qf[ds], st[kb], 0, 0, 0);
      if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    } else {
      if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ds = 0; ds < 8; ++ds) {
          const int c = ds * 2 + hi;
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + kb * 8192 + k_row_off + ((c ^ k_sw) << 4));
          st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], st[kb], 0, 0, 0);
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
    float mloc = st[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[1][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    if (__any((mloc - m_run) * p.sc > p.thr)) {   // defer-max (see attn.hip)
      const float m_new = fmaxf(m_run, mloc);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
    }
    const float mb = -m_run * p.sc;
    float psum = 0.f;
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 pf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(st[kb][r], p.sc, mb));
        psum += pv;
        pf[r >> 3][r & 7] = (__bf16)pv;
      }
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

    A4_VMCNT4();    // this wave's share of tile t+ahead-1 has landed (tile t+ahead may still be in flight)
    A4_BARRIER();   // ... and is published to the block
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail DMAs before the LDS is released
  if (grp == 0 && STAGGER) A4_BARRIER();             // re-balance the stagger

  attc::store_result(p, q0 + l31, head, hi, ot, m_run, l_run);
}

template <int VAR>
int launch(const Params& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn4_kernel<VAR>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      icv_set_error("attn4: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  const int64_t nwg = (int64_t)p.heads * p.nqb;
  hipLaunchKernelGGL(attn4_kernel<VAR>, dim3((unsigned)nwg), dim3(512), LDS_BYTES, st, p);
  return icv_check_launch("icv_attention(4)");
}

}  // namespace att4

int icv_attn4_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st) {
  att4::Params p;
  attc::fill_params(p, q, ldq, k, ldk, v, ldv, o, ldo, acc, ldacc, ml, state_in, state_out, Sq, Skv, heads, scale, att4::QB);
  switch (var) {
    case 0: return att4::launch<0>(p, st);
    case 1: return att4::launch<1>(p, st);
    case 4: return att4::launch<4>(p, st);
    case 5: return att4::launch<5>(p, st);
    case 6: return att4::launch<6>(p, st);
    case 7: return att4::launch<7>(p, st);
  }
  icv_set_error("icv_attention_fwd: unknown attn4 variant %d", var);
  return 1;
}
