// EXPERIMENT (icv_set_option("attn_kernel", 1); built only with ICV_EXPERIMENTS=1): the first attention kernel of this
// repo, kept for A/B.  Flash attention forward for the DiT (SURVEY.md §8a-3 K6 self-attention, K9 cross-attention):
// non-causal, head_dim 128, bf16 in/out, fp32 softmax + accumulation, arbitrary Sq / Skv.
//
// Structure (CDNA4, wave64, v_mfma_f32_32x32x16_bf16):
//   * block = 8 waves x 32 query rows = 256 rows of one head; KV tile = 64 keys, K and V tiles
//     staged HBM -> VGPR -> LDS (issue loads one tile ahead, write after the compute: the HBM
//     latency hides under the MFMA phase), double-buffered, ONE barrier per tile.
//   * "swapped" QK^T: S^T = K Q^T (K fragment = A operand, Q fragment = B operand, Q lives in
//     VGPRs for the whole kernel).  The 32x32 result puts query (lane&31) in the lane and 16 keys in
//     its registers, so row max / row sum are in-lane reductions + one exchange with lane^32.
//   * O^T = V^T P^T: the output tile also has the query in the lane -> the online-softmax rescale
//     is a per-lane scalar.  The MFMA k-slot -> key assignment is a free permutation (sums over
//     keys commute), so it is chosen to be exactly the order in which the S^T accumulator holds a
//     lane's keys: P goes from accumulator to bf16 B-operand with no cross-lane traffic; V^T
//     fragments come from LDS with ds_read_b64_tr_b16 (hardware 4x4 transpose) at matching keys.
//   * LDS images: K [64 keys][16 chunks of 16 B], chunk ^= key&15 (ds_read_b128 conflict-free);
//     V [64 keys][256 B], byte ^= (key&3)<<6 (the 32 lanes of a tr-read hit 32 distinct 8-B slots).
//   * the two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run ONE BARRIER APART: each
//     KV tile is two segments, S1 = {K reads, QK^T MFMAs, softmax VALU} and S2 = {V reads, PV MFMAs},
//     so while one wave of a SIMD is in softmax (VALU) its partner feeds the matrix pipe, and K/V
//     LDS bursts of the groups interleave instead of colliding.  Tile t+1 is written to LDS in the
//     interval where nobody reads its stage (group 0: end of S2(t); group 1: end of S1(t)).
//   * 1-D grid with XCD-aware remap: the 32 CUs of one XCD work on consecutive query blocks of
//     the SAME head, so they stream the same K/V tiles through that XCD's L2 together.
#include "icv_common.h"

namespace {

constexpr int D = 128;
constexpr int KVB = 64;
constexpr int NWAVE = 8;
constexpr int QB = 32 * NWAVE;
constexpr int TILE_BYTES = KVB * D * 2;  // 16 KiB
constexpr float NEG_BIG = -1.0e30f;

struct AttnParams {
  const bf16_t* q; int64_t ldq;
  const bf16_t* k; int64_t ldk;
  const bf16_t* v; int64_t ldv;
  bf16_t* o; int64_t ldo;
  int64_t Sq, Skv;
  int heads, nqb;
  float sc;   // scale * log2(e)
  float thr;  // defer-max threshold in log2 units (0 = rescale whenever the max grows)
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x4 lds_read_tr16(const char* p) {
  // ds_read_b64_tr_b16: within each 16-lane group the 16 x (4 x b16) loaded words are transposed:
  // lane t receives element (t&3) of the words loaded by lanes 4j + (t>>2), j = 0..3.
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  return __builtin_bit_cast(bf16x4, r);
}

// VAR bit flags (A/B switches, icv_set_option("attn_variant", v)):
//   1 = stagger the two wave groups by one barrier      2 = interleave the two QK^T accumulation chains
//   4 = s_setprio(1) around MFMA clusters                8/16 = timing ablations: no exp / no running max
template <int VAR>
__global__ __launch_bounds__(512) void attn_fwd_kernel(AttnParams p) {
  constexpr bool STAGGER = VAR & 1, QK_INTERLEAVE = VAR & 2, SETPRIO = VAR & 4;
  constexpr bool ABL_NOEXP = VAR & 8, ABL_NOMAX = VAR & 16;  // ablations (wrong results; timing only)
  constexpr bool ABL_NOK = VAR & 32, ABL_NOV = VAR & 64, ABL_NOSTAGE = VAR & 128;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [stage][K|V]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int hi = lane >> 5;
  const int l31 = lane & 31;

  // ---- work item: (head, query block) with bijective XCD remap ----
  const int nwg = p.heads * p.nqb;
  int wg;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int qn = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (qn + 1) : r * (qn + 1) + (xcd - r) * qn) + local;
  }
  const int head = wg / p.nqb;
  const int qb = wg - head * p.nqb;
  const int64_t q0 = (int64_t)qb * QB + wave * 32;

  const bf16_t* qh = p.q + (int64_t)head * D;
  const bf16_t* kh = p.k + (int64_t)head * D;
  const bf16_t* vh = p.v + (int64_t)head * D;

  // ---- Q fragments (B operand): lane (q = l31, hi) holds Q[q][ds*16 + hi*8 .. +8] ----
  bf16x8 qf[8];
  {
    int64_t qr = q0 + l31;
    qr = qr < p.Sq ? qr : p.Sq - 1;
    const bf16_t* qp = qh + qr * p.ldq + hi * 8;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) qf[ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16);
  }

  // ---- staging: thread owns chunks ci = tid and tid + 512 of each 64 x (16 chunks) tile ----
  const int sk0 = tid >> 4, sk1 = sk0 + 32, sc0 = tid & 15;  // (key, chunk); same chunk for both
  uint4 kreg0, kreg1, vreg0, vreg1;
#define LOAD_TILE(t_)                                                              \
  {                                                                                \
    int64_t kr0 = (int64_t)(t_) * KVB + sk0, kr1 = kr0 + 32;                       \
    kr0 = kr0 < p.Skv ? kr0 : p.Skv - 1;                                           \
    kr1 = kr1 < p.Skv ? kr1 : p.Skv - 1;                                           \
    kreg0 = *reinterpret_cast<const uint4*>(kh + kr0 * p.ldk + sc0 * 8);           \
    kreg1 = *reinterpret_cast<const uint4*>(kh + kr1 * p.ldk + sc0 * 8);           \
    vreg0 = *reinterpret_cast<const uint4*>(vh + kr0 * p.ldv + sc0 * 8);           \
    vreg1 = *reinterpret_cast<const uint4*>(vh + kr1 * p.ldv + sc0 * 8);           \
  }
  // (sk1 & 15) == (sk0 & 15) and (sk1 & 3) == (sk0 & 3): both rows share the swizzle
  const int k_wr_off = sk0 * 256 + ((sc0 ^ (sk0 & 15)) << 4);
  const int v_wr_off = TILE_BYTES + sk0 * 256 + ((sc0 << 4) ^ ((sk0 & 3) << 6));
#define WRITE_TILE(stage_)                                                         \
  {                                                                                \
    char* s_ = (stage_);                                                           \
    *reinterpret_cast<uint4*>(s_ + k_wr_off) = kreg0;                              \
    *reinterpret_cast<uint4*>(s_ + k_wr_off + 32 * 256) = kreg1;                   \
    *reinterpret_cast<uint4*>(s_ + v_wr_off) = vreg0;                              \
    *reinterpret_cast<uint4*>(s_ + v_wr_off + 32 * 256) = vreg1;                   \
  }

#define ATT_BARRIER()                                         \
  do {                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                             \
    asm volatile("" ::: "memory");                            \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)
  // group 0: waves 0-3, group 1: waves 4-7 (without STAGGER everyone behaves as group 0)
  const int grp = STAGGER ? __builtin_amdgcn_readfirstlane(wave >> 2) : 0;

  f32x16 ot[4];  // O^T: ot[d0][r] = O[q = l31][d = d0*32 + (r&3) + 8*(r>>2) + 4*hi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;

  const int nt = (int)((p.Skv + KVB - 1) / KVB);
  LOAD_TILE(0);
  WRITE_TILE(smem);
  // Retire EVERY prologue load here with a wait the compiler's waitcnt pass can see (the builtin, not
  // inline asm).  Otherwise hipcc (which hoists the tile-0 loads above the Q loads) must assume Q may
  // still be pending at the loop header and guards each QK^T MFMA with vmcnt(7..0) -- which in steady
  // state forces the prefetched tile t+1 loads to land at once and exposes HBM latency every tile.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
  ATT_BARRIER();
  if (nt > 1) LOAD_TILE(1);
  if (grp == 1) ATT_BARRIER();  // stagger: group 1 runs one barrier behind group 0

  // per-lane LDS addressing constants
  const int k_row_off = l31 * 256;          // K fragment: key = kb*32 + l31
  const int k_sw = l31 & 15;                // (kb*32 + l31) & 15
  const int g = lane >> 4, t16 = lane & 15;
  // V tr-read: key = kb*32 + hf*16 + 8u + 4hi + (t16>>2); byte = d0*64 + (g&1)*32 + (t16&3)*8
  const int v_key_lo = 4 * hi + (t16 >> 2);
  const int v_byte_lo = (g & 1) * 32 + (t16 & 3) * 8;
  const int v_sw = (t16 >> 2) << 6;         // (key & 3) << 6

  for (int t = 0; t < nt; ++t) {
    const char* ks = smem + (t & 1) * (2 * TILE_BYTES);
    const char* vs = ks + TILE_BYTES;

    // ---- S^T = K Q^T : 2 key blocks x 8 d-slices ----
    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);
    if (QK_INTERLEAVE) {
#pragma unroll
      for (int ds = 0; ds < 8; ++ds) {
        const int c = ds * 2 + hi;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + kb * (32 * 256) + k_row_off + ((c ^ k_sw) << 4));
          st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], st[kb], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int ds = 0; ds < 8; ++ds) {
          const int c = ds * 2 + hi;
          bf16x8 kf;
          if (ABL_NOK) kf = qf[(ds + kb) & 7];
          else kf = *reinterpret_cast<const bf16x8*>(ks + kb * (32 * 256) + k_row_off + ((c ^ k_sw) << 4));
          st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], st[kb], 0, 0, 0);
        }
      }
    }
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    // ---- mask the tail tile ----
    if ((int64_t)(t + 1) * KVB > p.Skv) {
      const int64_t kv0 = (int64_t)t * KVB;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= p.Skv) st[kb][r] = NEG_BIG;
        }
    }
    // ---- S1 tail: running max (lane-local; one exchange with lane^32) + deferred rescale ----
    float mloc = st[0][0];
    if (!ABL_NOMAX) {
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[1][r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    }
    // Defer-max: keep the old running max while this tile's max exceeds it by <= p.thr (in
    // log2 units): P is then bounded by 2^THR instead of 1 (fine for bf16 P / fp32 l, O) and the
    // 64-register O rescale is skipped.  The previous tile's PV finished before this point (it is
    // in the previous S2), so O, l are the only state at the old max: both are scaled exactly once.
    if (__any((mloc - m_run) * p.sc > p.thr)) {
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

    if (grp == 1 && !ABL_NOSTAGE) {  // group 1 stages tile t+1 at the end of its S1 (see header)
      if (t + 1 < nt) WRITE_TILE(smem + ((t + 1) & 1) * (2 * TILE_BYTES));
      if (t + 2 < nt) LOAD_TILE(t + 2);
    }
    ATT_BARRIER();

    // ---- S2: P = exp2(S*sc - m*sc) -> bf16 (32 keys at a time) interleaved with O^T += V^T P^T ----
    float psum = 0.f;
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 pf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = ABL_NOEXP ? fmaf(st[kb][r], p.sc, mb) : __builtin_amdgcn_exp2f(fmaf(st[kb][r], p.sc, mb));
        psum += pv;
        pf[r >> 3][r & 7] = (__bf16)pv;
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int kk = kb * 2 + hf;
#pragma unroll
        for (int d0 = 0; d0 < 4; ++d0) {
          const int key0 = kk * 16 + v_key_lo;
          const int byte = (d0 * 64 + v_byte_lo) ^ v_sw;
          const bf16x4 va = lds_read_tr16(vs + key0 * 256 + byte);
          const bf16x4 vb = lds_read_tr16(vs + (key0 + 8) * 256 + byte);
          bf16x8 vf;
          vf[0] = va[0]; vf[1] = va[1]; vf[2] = va[2]; vf[3] = va[3];
          vf[4] = vb[0]; vf[5] = vb[1]; vf[6] = vb[2]; vf[7] = vb[3];
          if (ABL_NOV) vf = qf[(d0 + kk) & 7];
          ot[d0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[hf], ot[d0], 0, 0, 0);
        }
      }
    }
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    l_run += psum;

    if (grp == 0 && !ABL_NOSTAGE) {  // group 0 stages tile t+1 at the end of its S2
      if (t + 1 < nt) WRITE_TILE(smem + ((t + 1) & 1) * (2 * TILE_BYTES));
      if (t + 2 < nt) LOAD_TILE(t + 2);
    }
    ATT_BARRIER();
  }
  if (grp == 0) ATT_BARRIER();  // re-balance the stagger

  // ---- epilogue: normalise, bf16, 8-byte stores ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int64_t qr = q0 + l31;
  if (qr < p.Sq) {
    bf16_t* op = p.o + qr * p.ldo + (int64_t)head * D + 4 * hi;
#pragma unroll
    for (int d0 = 0; d0 < 4; ++d0)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float a = ot[d0][rr * 4 + 0] * inv, b = ot[d0][rr * 4 + 1] * inv;
        const float c = ot[d0][rr * 4 + 2] * inv, d = ot[d0][rr * 4 + 3] * inv;
        *reinterpret_cast<uint2*>(op + d0 * 32 + rr * 8) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
      }
  }
}

}  // namespace

// attn_kernel = 1: icv_attention_fwd only (no carried state)
int icv_attn1_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                       int64_t ldo, int64_t Sq, int64_t Skv, int64_t heads, float scale, void* stream) {
  AttnParams p;
  p.q = (const bf16_t*)q; p.ldq = ldq; p.k = (const bf16_t*)k; p.ldk = ldk;
  p.v = (const bf16_t*)v; p.ldv = ldv; p.o = (bf16_t*)o; p.ldo = ldo;
  p.Sq = Sq; p.Skv = Skv; p.heads = (int)heads;
  p.nqb = (int)((Sq + QB - 1) / QB);
  p.sc = scale * 1.4426950408889634f;
  p.thr = (float)icv_get_option_int("attn_defer_max_log2", 8);
  const int64_t nwg = (int64_t)p.heads * p.nqb;
  ICV_REQUIRE(nwg < (1LL << 31), "icv_attention_fwd: grid too large");
  const int var = icv_get_option_int("attn_variant", 5);
  dim3 grid((unsigned)nwg), block(512);
  hipStream_t st = (hipStream_t)stream;
  switch (var) {
#define ATT_CASE(V) case V: hipLaunchKernelGGL(attn_fwd_kernel<V>, grid, block, 0, st, p); break;
    ATT_CASE(0) ATT_CASE(1) ATT_CASE(2) ATT_CASE(3) ATT_CASE(5) ATT_CASE(7) ATT_CASE(13) ATT_CASE(29) ATT_CASE(37) ATT_CASE(69) ATT_CASE(133) ATT_CASE(229) ATT_CASE(253)
#undef ATT_CASE
    default: icv_set_error("icv_attention_fwd: unknown attn_variant %d", var); return 1;
  }
  return icv_check_launch("icv_attention_fwd");
}
