// Flash attention forward, third structure: ONE WAVE PER SIMD, 64 query rows
// per wave.  Same math, MFMA fragments, key-permutation trick and LDS images as attn.hip / attn2.hip.
//
//   block = 4 waves (256 threads, one wave per SIMD, up to 512 VGPR+AGPR each) = 256 query rows of one
//   head; each wave owns two 32-row sub-blocks A and B.  Per 64-key tile the wave's stream is
//       K frags (16 x ds_read_b128, kept in VGPRs)
//       QK^T_A (16 MFMA)
//       QK^T_B (16 MFMA, same K frags)   ||  softmax_A  (VALU: max, deferred rescale, exp2, pack)
//       V^T frags (32 x ds_read_b64_tr_b16, kept)
//       PV_A   (16 MFMA)                 ||  softmax_B
//       PV_B   (16 MFMA, same V frags)   ||  staging of the next tile
//   so every VALU-heavy softmax has 16 independent MFMAs next to it inside the SAME wave (the compiler
//   interleaves them; the matrix pipe is not shared with another wave), and each K/V fragment read from
//   LDS feeds two MFMAs instead of one.  K/V tiles (64 keys) are staged HBM -> VGPR -> LDS one tile
//   ahead, double-buffered, one barrier per tile.  Carried softmax state as in attn2.hip.
#include "attn_common.h"

namespace att3 {

using attc::D;
using attc::NEG_BIG;
using attc::Params;
using attc::lds_read_tr16;
constexpr int KVB = 64;
constexpr int QB = 256;                   // 4 waves x 64 rows
constexpr int TILE_BYTES = KVB * D * 2;   // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
template <int VAR>
__global__ __launch_bounds__(256, 1) void attn3_kernel(Params p) {
  constexpr bool SETPRIO = VAR & 4;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];  // 64 KiB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int hi = lane >> 5;
  const int l31 = lane & 31;

  int head, qb;
  attc::work_item(p, head, qb);
  const int64_t q0 = (int64_t)qb * QB + wave * 64;

  const bf16_t* qh = p.q + (int64_t)head * D;
  const bf16_t* kh = p.k + (int64_t)head * D;
  const bf16_t* vh = p.v + (int64_t)head * D;

  int64_t qr_c[2];
  bf16x8 qf[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    int64_t r = q0 + s * 32 + l31;
    qr_c[s] = r < p.Sq ? r : p.Sq - 1;
    const bf16_t* qp = qh + qr_c[s] * p.ldq + hi * 8;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) qf[s][ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16);
  }

  // ---- staging: 256 threads, thread owns chunk sc0 of keys sk0 + 16u (u = 0..3) ----
  const int sk0 = tid >> 4, sc0 = tid & 15;   // sk0 in 0..15
  uint4 kreg0, kreg1, kreg2, kreg3, vreg0, vreg1, vreg2, vreg3;
#define A3_LOAD_ROW(KR_, VR_, U_, T_)                                        \
  {                                                                          \
    int64_t kr_ = (int64_t)(T_) * KVB + sk0 + 16 * (U_);                     \
    kr_ = kr_ < p.Skv ? kr_ : p.Skv - 1;                                     \
    KR_ = *reinterpret_cast<const uint4*>(kh + kr_ * p.ldk + sc0 * 8);       \
    VR_ = *reinterpret_cast<const uint4*>(vh + kr_ * p.ldv + sc0 * 8);       \
  }
#define A3_LOAD_TILE(T_)                  \
  {                                       \
    A3_LOAD_ROW(kreg0, vreg0, 0, T_)      \
    A3_LOAD_ROW(kreg1, vreg1, 1, T_)      \
    A3_LOAD_ROW(kreg2, vreg2, 2, T_)      \
    A3_LOAD_ROW(kreg3, vreg3, 3, T_)      \
  }
  // keys sk0 + 16u share (key & 15) and (key & 3): one swizzled offset, + u * 16 rows
  const int k_wr_off = sk0 * 256 + ((sc0 ^ (sk0 & 15)) << 4);
  const int v_wr_off = TILE_BYTES + sk0 * 256 + ((sc0 << 4) ^ ((sk0 & 3) << 6));
#define A3_WRITE_TILE(STAGE_)                                                 \
  {                                                                           \
    char* s_ = (STAGE_);                                                      \
    *reinterpret_cast<uint4*>(s_ + k_wr_off) = kreg0;                         \
    *reinterpret_cast<uint4*>(s_ + k_wr_off + 4096) = kreg1;                  \
    *reinterpret_cast<uint4*>(s_ + k_wr_off + 8192) = kreg2;                  \
    *reinterpret_cast<uint4*>(s_ + k_wr_off + 12288) = kreg3;                 \
    *reinterpret_cast<uint4*>(s_ + v_wr_off) = vreg0;                         \
    *reinterpret_cast<uint4*>(s_ + v_wr_off + 4096) = vreg1;                  \
    *reinterpret_cast<uint4*>(s_ + v_wr_off + 8192) = vreg2;                  \
    *reinterpret_cast<uint4*>(s_ + v_wr_off + 12288) = vreg3;                 \
  }

  f32x16 ot[2][4];
  float m_run[2], l_run[2];
  attc::load_state(p, qr_c[0], head, hi, ot[0], m_run[0], l_run[0]);
  attc::load_state(p, qr_c[1], head, hi, ot[1], m_run[1], l_run[1]);

  const int nt = (int)((p.Skv + KVB - 1) / KVB);
  A3_LOAD_TILE(0);
  A3_WRITE_TILE(smem);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // retire every prologue load visibly (see attn.hip)
  __syncthreads();
  if (nt > 1) A3_LOAD_TILE(1);

  const int k_row_off = l31 * 256;
  const int k_sw = l31 & 15;
  const int g = lane >> 4, t16 = lane & 15;
  const int v_key_lo = 4 * hi + (t16 >> 2);
  const int v_byte_lo = (g & 1) * 32 + (t16 & 3) * 8;
  const int v_sw = (t16 >> 2) << 6;

  for (int t = 0; t < nt; ++t) {
    const char* ks = smem + (t & 1) * STAGE_BYTES;
    const char* vs = ks + TILE_BYTES;
    const bool tail = (int64_t)(t + 1) * KVB > p.Skv;
    const int64_t kv0 = (int64_t)t * KVB;

    f32x16 st[2][2];
    bf16x8 pf[2][4];
    float psum[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // ---- S^T_s = K Q_s^T  (K fragments are re-read per sub-block: LDS has the headroom, VGPRs do not) ----
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[s][kb][r] = 0.f;
      }
      if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ds = 0; ds < 8; ++ds) {
          const int c = ds * 2 + hi;
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + kb * 8192 + k_row_off + ((c ^ k_sw) << 4));
          st[s][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s][ds], st[s][kb], 0, 0, 0);
        }
      if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // ---- softmax_s: mask, running max (+ deferred rescale), P = exp2 -> bf16 ----
      if (tail) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= p.Skv) st[s][kb][r] = NEG_BIG;
          }
      }
      float mloc = st[s][0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[s][0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[s][1][r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      if (__any((mloc - m_run[s]) * p.sc > p.thr)) {
        const float m_new = fmaxf(m_run[s], mloc);
        const float alpha = __builtin_amdgcn_exp2f((m_run[s] - m_new) * p.sc);
        m_run[s] = m_new;
        l_run[s] *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[s][i][r] *= alpha;
      }
      const float mb = -m_run[s] * p.sc;
      float ps = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(st[s][kb][r], p.sc, mb));
          ps += pv;
          pf[s][kb * 2 + (r >> 3)][r & 7] = (__bf16)pv;
        }
      psum[s] = ps;
    }
    // ---- O^T_s += V^T P_s^T ----
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int d0 = 0; d0 < 4; ++d0) {
          const int key0 = kk * 16 + v_key_lo;
          const int byte = (d0 * 64 + v_byte_lo) ^ v_sw;
          const bf16x4 va = lds_read_tr16(vs + key0 * 256 + byte);
          const bf16x4 vb = lds_read_tr16(vs + (key0 + 8) * 256 + byte);
          bf16x8 vf;
          vf[0] = va[0]; vf[1] = va[1]; vf[2] = va[2]; vf[3] = va[3];
          vf[4] = vb[0]; vf[5] = vb[1]; vf[6] = vb[2]; vf[7] = vb[3];
          ot[s][d0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s][kk], ot[s][d0], 0, 0, 0);
        }
      }
    }
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    l_run[0] += psum[0];
    l_run[1] += psum[1];

    if (t + 1 < nt) A3_WRITE_TILE(smem + ((t + 1) & 1) * STAGE_BYTES);
    __syncthreads();
    if (t + 2 < nt) A3_LOAD_TILE(t + 2);
  }

  attc::store_result(p, q0 + l31, head, hi, ot[0], m_run[0], l_run[0]);
  attc::store_result(p, q0 + 32 + l31, head, hi, ot[1], m_run[1], l_run[1]);
}

}  // namespace att3

int icv_attn3_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st) {
  att3::Params p;
  attc::fill_params(p, q, ldq, k, ldk, v, ldv, o, ldo, acc, ldacc, ml, state_in, state_out, Sq, Skv, heads, scale, att3::QB);
  dim3 grid((unsigned)((int64_t)p.heads * p.nqb)), block(256);
  switch (var) {
    case 0: hipLaunchKernelGGL(att3::attn3_kernel<0>, grid, block, 0, st, p); break;
    case 4: hipLaunchKernelGGL(att3::attn3_kernel<4>, grid, block, 0, st, p); break;
    default: icv_set_error("icv_attention_fwd: unknown attn3 variant %d", var); return 1;
  }
  return icv_check_launch("icv_attention(3)");
}
