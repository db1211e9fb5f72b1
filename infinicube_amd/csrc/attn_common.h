// Pieces shared by the flash-attention kernel variants (attn7.hip default, attn2.hip, and the experiments/ family):
// launch parameters, (head, query-block) work-item mapping, the transposed LDS read, the carried
// online-softmax state (load / init) and the epilogue (state write-back or normalised bf16 output).
// Fragment conventions (see experiments/attn1.hip for the derivation): a wave owns 32 query rows, query = lane&31;
// O^T accumulator ot[d0][r] = O[q][d = d0*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)]; the row sum l is kept as
// two half-lane partials (lanes q and q+32).
#pragma once
#include "icv_common.h"

unsigned long long* icv_attention_trace_buffer(int* capacity);   // attention.hip (icv_attention_trace)

namespace attc {

constexpr int D = 128;
constexpr float NEG_BIG = -1.0e30f;

struct Params {
  const bf16_t* q; int64_t ldq;
  const bf16_t* k; int64_t ldk;
  const bf16_t* v; int64_t ldv;
  bf16_t* o; int64_t ldo;
  float* acc; int64_t ldacc;   // carried O^T state, f32 [Sq, heads*128] (may be NULL)
  float* ml;                   // carried (m, l) per (row, head): f32 [Sq, heads, 2]
  int64_t Sq, Skv;
  int heads, nqb;
  int state_in, state_out;   // state_out: 0 = normalise + store bf16, 1 = write the carried state, 2 = normalise + ADD into bf16 o
  float sc;   // scale * log2(e)
  float thr;  // defer-max threshold, log2 units
  int ablate; // timing ablations (attn7: 1 = no K/V DMA after the prologue, 2 = no per-tile barrier); results are then WRONG
  // diagnostics (icv_attention_trace; attn7 only): per work-group {start, end} in 100 MHz s_memrealtime ticks, HW_ID, XCC_ID
  unsigned long long* trace;
  int trace_cap;
};


inline void fill_params(Params& p, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                        int64_t ldv, void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                        int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int rows_per_block) {
  p.q = (const bf16_t*)q; p.ldq = ldq; p.k = (const bf16_t*)k; p.ldk = ldk;
  p.v = (const bf16_t*)v; p.ldv = ldv; p.o = (bf16_t*)o; p.ldo = ldo;
  p.acc = acc; p.ldacc = ldacc; p.ml = ml; p.state_in = state_in; p.state_out = state_out;
  p.Sq = Sq; p.Skv = Skv; p.heads = (int)heads;
  p.nqb = (int)((Sq + rows_per_block - 1) / rows_per_block);
  p.sc = scale * 1.4426950408889634f;
  if (fabsf(p.sc - 1.0f) < 1e-6f) p.sc = 1.0f;   // "unit scale": the caller folded scale * log2(e) into K (scale = ln 2)
  p.thr = (float)icv_get_option_int("attn_defer_max_log2", 8);
  p.ablate = 0;
  p.trace = nullptr;
  p.trace_cap = 0;
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// ds_read_b64_tr_b16: within each 16-lane group the 16 x (4 x b16) loaded words are transposed: lane t
// receives element (t&3) of the words loaded by lanes 4j + (t>>2), j = 0..3 (probed: tools/probe_tr.hip).
__device__ __forceinline__ bf16x4 lds_read_tr16(const char* p) {
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  return __builtin_bit_cast(bf16x4, r);
}

// (head, query block) of this workgroup: block b runs on XCD b%8, so give each XCD a contiguous run of
// work items (head-major): its 32 CUs then stream the SAME head's K/V through that XCD's L2 together.
__device__ __forceinline__ void work_item(const Params& p, int& head, int& qb) {
  const int nwg = p.heads * p.nqb;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, local = bid >> 3;
  const int qn = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (qn + 1) : r * (qn + 1) + (xcd - r) * qn) + local;
  head = wg / p.nqb;
  qb = wg - head * p.nqb;
}

// softmax state of one 32-row sub-block: from the carried buffers (row qr_c, clamped) or empty
__device__ __forceinline__ void load_state(const Params& p, int64_t qr_c, int head, int hi, f32x16 (&ot)[4],
                                           float& m_run, float& l_run) {
  if (p.state_in) {
    const float* ap = p.acc + qr_c * p.ldacc + (int64_t)head * D + 4 * hi;
#pragma unroll
    for (int d0 = 0; d0 < 4; ++d0)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float4 a = *reinterpret_cast<const float4*>(ap + d0 * 32 + rr * 8);
        ot[d0][rr * 4 + 0] = a.x; ot[d0][rr * 4 + 1] = a.y; ot[d0][rr * 4 + 2] = a.z; ot[d0][rr * 4 + 3] = a.w;
      }
    const float2 mlv = *reinterpret_cast<const float2*>(p.ml + (qr_c * p.heads + head) * 2);
    m_run = mlv.x;
    l_run = hi == 0 ? mlv.y : 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
    m_run = NEG_BIG;
    l_run = 0.f;
  }
}

// epilogue of one 32-row sub-block: write the state back, or normalise and store bf16 (8-byte stores)
__device__ __forceinline__ void store_result(const Params& p, int64_t qr, int head, int hi, const f32x16 (&ot)[4],
                                             float m_run, float l_run) {
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qr >= p.Sq) return;
  if (p.state_out == 1) {
    float* ap = p.acc + qr * p.ldacc + (int64_t)head * D + 4 * hi;
#pragma unroll
    for (int d0 = 0; d0 < 4; ++d0)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        *reinterpret_cast<float4*>(ap + d0 * 32 + rr * 8) =
            make_float4(ot[d0][rr * 4 + 0], ot[d0][rr * 4 + 1], ot[d0][rr * 4 + 2], ot[d0][rr * 4 + 3]);
    if (hi == 0) *reinterpret_cast<float2*>(p.ml + (qr * p.heads + head) * 2) = make_float2(m_run, l_tot);
  } else {
    const float inv = 1.0f / l_tot;
    bf16_t* op = p.o + qr * p.ldo + (int64_t)head * D + 4 * hi;
#pragma unroll
    for (int d0 = 0; d0 < 4; ++d0)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        float a = ot[d0][rr * 4 + 0] * inv, b = ot[d0][rr * 4 + 1] * inv;
        float c = ot[d0][rr * 4 + 2] * inv, d = ot[d0][rr * 4 + 3] * inv;
        if (p.state_out == 2) {   // o += result (i2v: image cross-attention summed with the text one)
          const uint2 prev = *reinterpret_cast<const uint2*>(op + d0 * 32 + rr * 8);
          a += __uint_as_float(prev.x << 16); b += __uint_as_float(prev.x & 0xFFFF0000u);
          c += __uint_as_float(prev.y << 16); d += __uint_as_float(prev.y & 0xFFFF0000u);
        }
        *reinterpret_cast<uint2*>(op + d0 * 32 + rr * 8) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
      }
  }
}

}  // namespace attc
