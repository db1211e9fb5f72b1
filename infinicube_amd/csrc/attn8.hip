// fp8 (OCP e4m3) flash attention forward for the fp8 mode (BASELINE.json config #5 "CDNA4 fp8 path"): both GEMMs of
// attention on v_mfma_scale_f32_32x32x64_f8f6f4 (twice the bf16 MFMA rate, half the LDS / DMA bytes per key).
//
//   icv_attention_fp8_prepare : per-head abs-max of Q, K, V -> power-of-two scales 2^e (e = ceil(log2(amax / 448)));
//                               Qq, Kq = e4m3(x * 2^-e) row-major [S, H*128] bytes;  V is written TRANSPOSED and
//                               key-permuted per 64-key tile: Vt[head][tile][d = 0..127][64 bytes], byte g*32 + j of a
//                               row = V[key(g, j)][d] with key(g, j) = (j>>4)*32 + (j&3) + 8*((j&15)>>2) + 4*g — exactly
//                               the k-slot order in which a lane of the S^T accumulator holds its 32 keys of the tile.
//   icv_attention_fp8_fwd     : attn7.hip's structure (LDS-DMA ring, lazy max, unit scale with the reference in the
//                               first MFMA's C operand) with
//       S^T = Kq Qq^T : 2 MFMAs (K = 64 each) per 32-key block instead of 8; the power-of-two scales of Q and K ride
//                       in the MFMA's E8M0 block-scale operands, so S arrives in log2 units with no VALU at all
//                       (K already carries (1/sqrt d) log2 e: the DiT folds it into the K RMSNorm weight);
//       O^T += Vt P^T : ONE MFMA (K = 64 keys) per 32-row d block; P is packed to e4m3 lane-locally in accumulator
//                       order (16 v_cvt_pk_fp8_f32), Vt fragments are two plain ds_read_b128 (no transposing read).
// Operand layout of the f8f6f4 MFMAs (lane (r = lane & 31, g = lane >> 5) holds K-bytes [32 g, 32 g + 32) of row r)
// and the scale semantics (E8M0 127 = 1.0) were probed on hardware: tools/probe_f8.hip.
#include "attn_common.h"

namespace att8 {

using attc::D;
using attc::NEG_BIG;
constexpr int KVB = 64;
constexpr int QB = 256;
constexpr int KT_BYTES = KVB * D;          // 8 KiB: K tile [64 keys][128 B]
constexpr int VT_BYTES = D * KVB;          // 8 KiB: V^T tile [128 d][64 B]
constexpr int STAGE_BYTES = KT_BYTES + VT_BYTES;
constexpr int NSTAGE = 4;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;   // 64 KiB
constexpr float FP8_MAX = 448.0f;

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int scale_exp(float amax) {   // e with 2^e >= amax / 448
  return amax > 0.f ? (int)ceilf(log2f(amax / FP8_MAX)) : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// prepare, pass 1: per-head abs-max (bits of a non-negative float order like ints -> atomicMax on the int view)
// block = 256 threads = 16 rows x 16 lanes (a lane owns 8 consecutive elements of a head's 128), grid (heads, row blocks)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void amax_kernel(const bf16_t* __restrict__ x, int64_t ld, int64_t rows, int* __restrict__ amax_bits) {
  const int head = blockIdx.x;
  const int lane16 = threadIdx.x & 15, rsub = threadIdx.x >> 4;
  float m = 0.f;
  for (int64_t r = (int64_t)blockIdx.y * 256 + rsub; r < min(rows, (int64_t)(blockIdx.y + 1) * 256); r += 16) {
    const uint4 u = *reinterpret_cast<const uint4*>(x + r * ld + head * D + lane16 * 8);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) m = fmaxf(m, fmaxf(fabsf(__uint_as_float(w[i] << 16)), fabsf(__uint_as_float(w[i] & 0xFFFF0000u))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(amax_bits + head, __float_as_int(m));
}

__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}
// the same when the destination's previous content may be anything (both halves are overwritten): no zero-initialising v_mov per dword
__device__ __forceinline__ int pack_fp8x4_over(float a, float b, float c, float d, int old) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}

// pass 2a: Q / K rows -> e4m3 [rows, H*128] with the head's power-of-two scale
__global__ __launch_bounds__(256) void quant_rows_kernel(const bf16_t* __restrict__ x, int64_t ld, int64_t rows,
                                                         const float* __restrict__ amax, unsigned char* __restrict__ out, int64_t ldo) {
  const int head = blockIdx.x;
  const int lane16 = threadIdx.x & 15, rsub = threadIdx.x >> 4;
  const float inv = exp2f((float)-scale_exp(amax[head]));
  for (int64_t r = (int64_t)blockIdx.y * 256 + rsub; r < min(rows, (int64_t)(blockIdx.y + 1) * 256); r += 16) {
    const uint4 u = *reinterpret_cast<const uint4*>(x + r * ld + head * D + lane16 * 8);
    const float f0 = __uint_as_float(u.x << 16), f1 = __uint_as_float(u.x & 0xFFFF0000u);
    const float f2 = __uint_as_float(u.y << 16), f3 = __uint_as_float(u.y & 0xFFFF0000u);
    const float f4 = __uint_as_float(u.z << 16), f5 = __uint_as_float(u.z & 0xFFFF0000u);
    const float f6 = __uint_as_float(u.w << 16), f7 = __uint_as_float(u.w & 0xFFFF0000u);
    *reinterpret_cast<uint2*>(out + r * ldo + head * D + lane16 * 8) =
        make_uint2(pack_fp8x4(f0 * inv, f1 * inv, f2 * inv, f3 * inv), pack_fp8x4(f4 * inv, f5 * inv, f6 * inv, f7 * inv));
  }
}

// key of k-slot j (0..31) of lane group g (0..1) inside a 64-key tile: the S^T accumulator order (see the header)
__device__ __host__ __forceinline__ int tile_key(int g, int j) { return (j >> 4) * 32 + (j & 3) + 8 * ((j & 15) >> 2) + 4 * g; }

// pass 2b: V -> transposed, key-permuted e4m3 tiles.  One block per (tile, head): 256 threads, thread t -> d = t >> 1, g = t & 1.
__global__ __launch_bounds__(256) void quant_vt_kernel(const bf16_t* __restrict__ v, int64_t ld, int64_t Skv,
                                                       const float* __restrict__ amax, unsigned char* __restrict__ vt, int ntiles) {
  __shared__ bf16_t tile[KVB][D + 8];   // +8: column reads hit different banks
  const int t = blockIdx.x, head = blockIdx.y;
  const float inv = exp2f((float)-scale_exp(amax[head]));
  for (int i = threadIdx.x; i < KVB * 16; i += 256) {   // 64 rows x 16 chunks of 8 elements
    const int key = i >> 4, c = i & 15;
    const int64_t r = (int64_t)t * KVB + key;
    uint4 u = make_uint4(0, 0, 0, 0);                      // keys past Skv: zeros (their P is masked to 0 anyway)
    if (r < Skv) u = *reinterpret_cast<const uint4*>(v + r * ld + head * D + c * 8);
    *reinterpret_cast<uint4*>(&tile[key][c * 8]) = u;
  }
  __syncthreads();
  const int d = threadIdx.x >> 1, g = threadIdx.x & 1;
  unsigned w[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    float f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = __uint_as_float(((unsigned)tile[tile_key(g, q * 4 + e)][d]) << 16) * inv;
    w[q] = pack_fp8x4(f[0], f[1], f[2], f[3]);
  }
  uint4* dst = reinterpret_cast<uint4*>(vt + (((int64_t)head * ntiles + t) * D + d) * KVB + g * 32);
  dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
  dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
struct Params {
  const unsigned char* q; int64_t ldq;     // e4m3 [Sq, H*128]
  const unsigned char* k; int64_t ldk;     // e4m3 [Skv, H*128]
  const unsigned char* vt;                 // e4m3 [H][ntiles][128][64]
  const float* amax;                       // f32 [3][H]: q, k, v
  int64_t Sq, Skv;
  int heads, nqb, ntiles;
  // Keys / values as PIECES (sequence parallel, e4m3 on the wire: every rank quantised its own rows of the chunk and shipped
  // kq rows + Vt tiles): piece i holds `piece_rows` keys as tpp = ceil(piece_rows / 64) tiles; its kq rows start at
  // k + i * piece_stride, its Vt tiles [H][tpp][128][64] at vt + i * piece_stride.  One piece = the plain layout.
  int64_t piece_stride;
  int piece_rows, tpp;
  float thr;
  attc::Params c;   // o / ldo, carried state (acc, ldacc, ml, state_in, state_out), Sq, heads: what load_state / store_result read
  // ---- arrival-gated pieces (VAR bit 256, icv_attention_fp8_fwd_pieces_gated): the pieces are walked in the order seq_piece[0..n) (this
  // rank's own blob first, then the peers in the order the exchange delivers them); position i may be read once
  // (int)(flags[seq_flag[i]] - seq_value[i]) >= 0 (seq_flag < 0: there when the launch starts); piece `own_index` lives at own_delta bytes
  // from its slot in the gathered chunk (the rank's own blob is read where it was quantised, not copied).  A flag that does not come
  // within timeout_ticks sets *err = 0x80000000 | position and the launch finishes on whatever the slot holds (csrc/attn7p.hip's rule).
  const unsigned* flags;
  unsigned* err;
  unsigned long long timeout_ticks;
  int64_t own_delta;
  int own_index, n_seq;
  int seq_piece[ICV_ATTN_MAX_PIECES];
  int seq_flag[ICV_ATTN_MAX_PIECES];
  unsigned seq_value[ICV_ATTN_MAX_PIECES];
};

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

// the same with the address split into a wave-uniform base (SGPR pair) and a per-lane 32-bit byte offset that does not change from
// tile to tile: no vector arithmetic per request (round 6: attn8 is bounded by VALU issue, profiles/r06/rocprofv3_summary_i2v720_fp8.md)
__device__ __forceinline__ void dma16s(const void* base, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  // the "s" constraint alone does not move a value the compiler chose to keep in vector registers: make it scalar (folds away when it is)
  const uint64_t b = (uint64_t)(uintptr_t)base;
  const uint64_t sbase = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)b);
  (void)keep;
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory", "m0");
}

__device__ __forceinline__ i32x8 read32(const char* p0, const char* p1) {
  const i32x4 lo = *reinterpret_cast<const i32x4*>(p0);
  const i32x4 hi = *reinterpret_cast<const i32x4*>(p1);
  return (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// VAR bit flags: 512 = one barrier per two key tiles on an 8-stage ring (with 32), 256 = arrival-gated pieces (with 32), 32 = software-pipelined key loop (S(t+1) and O += V(t-1)P(t-1) issued around tile t's softmax; implies 8), 64 = its row sums on
// packed adds, 1 = wave groups one tile apart, 4 = s_setprio(1) around MFMA clusters, 8 = lean vector work (LDS-DMA addresses as
// scalar base + constant lane offset, row sums on packed fp32 adds)
template <int VAR>
__global__ __launch_bounds__(512) void attn8_kernel(Params p) {
  constexpr bool STAGGER = VAR & 1, SETPRIO = VAR & 4, PIPE = VAR & 32, LEAN = (VAR & 8) || PIPE, PKADD = VAR & 64, PFD2 = VAR & 128, GATE = VAR & 256, TWO = VAR & 512;
  constexpr int NSTG = TWO ? 8 : NSTAGE;       // TWO (with 32): 8-stage ring, ONE vmcnt(0) + barrier per TWO key tiles
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int l31 = lane & 31;
  const float p_lim = __builtin_amdgcn_exp2f(p.thr);

  // (head, query block) with the XCD-aware remap of attn_common.h
  int head, qb;
  {
    const int nwg = p.heads * p.nqb;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int qn = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (qn + 1) : r * (qn + 1) + (xcd - r) * qn) + local;
    head = wg / p.nqb;
    qb = wg - head * p.nqb;
  }
  const int64_t q0 = (int64_t)qb * QB + wave * 32;
  int64_t qr_c = q0 + l31;
  qr_c = qr_c < p.Sq ? qr_c : p.Sq - 1;

  // E8M0 block scales of the three operands (one power of two per head)
  const int sQ = 127 + scale_exp(p.amax[head]), sK = 127 + scale_exp(p.amax[p.heads + head]);
  const int sV = 127 + scale_exp(p.amax[2 * p.heads + head]);

  // Q fragments: lane (q = lane & 31, g = hi) holds bytes [64 s + 32 g, +32) of its row, s = 0, 1
  i32x8 qf[2];
  {
    const unsigned char* qp = p.q + qr_c * p.ldq + (int64_t)head * D + hi * 32;
#pragma unroll
    for (int s = 0; s < 2; ++s) qf[s] = read32(reinterpret_cast<const char*>(qp + s * 64), reinterpret_cast<const char*>(qp + s * 64 + 16));
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // retire ordinary loads before any LDS-DMA is in flight (see experiments/attn4.hip)

  f32x16 ot[4];
  float m_run, l_run;
  attc::load_state(p.c, qr_c, head, hi, ot, m_run, l_run);      // empty, or the state carried from the previous key chunk
  // the carried state's loads retire HERE: left pending, the compiler's wait for them lands on the accumulators' first use inside
  // the key loop - a vmcnt(0) per tile that also waits for the LDS-DMA requests just issued (found in the ISA, round 6)
  if (!(VAR & 16)) __builtin_amdgcn_s_waitcnt(0x0F70);     // VAR bit 16: round 5's behaviour, kept for the A/B (tools/attn_fp8_bench.py)
  float m_base = m_run < -1.0e29f ? 0.f : m_run;               // reference baked into cinit (0 while there is none yet)
  f32x16 cinit;
#pragma unroll
  for (int r = 0; r < 16; ++r) cinit[r] = -m_base;

  // ---- LDS-DMA lane mapping ----
  // K tile: wave w covers key rows 8w..8w+7 (8 lanes x 16 B per row); physical chunk pc holds logical pc ^ ((row >> 1) & 7)
  const int krow = wave * 8 + (lane >> 3);
  const int kcol = ((lane & 7) ^ ((krow >> 1) & 7)) * 16;
  // V^T tile: wave w covers bytes [1024 w, +1024) = d rows 16w..16w+15 (4 lanes x 16 B per row); pc holds pc ^ ((d >> 2) & 3)
  const int vrow = wave * 16 + (lane >> 2);
  const int vcol = ((lane & 3) ^ ((vrow >> 2) & 3)) * 16;
  const int nt = p.ntiles;
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const unsigned char* kh = p.k + (int64_t)head * D;
  // tile -> (piece, tile inside the piece), advanced incrementally: A8_DMA_TILE is called once per tile index, in order
  int d_pc = 0, d_tl = 0;
  // LEAN: per-lane byte offsets inside a tile, fixed for the whole launch (the ragged last tile of a piece clamps its rows)
  const int rag_rows = p.piece_rows - (p.tpp - 1) * KVB;
  const unsigned koff_full = (unsigned)(krow * (int)p.ldk + kcol);
  const unsigned koff_rag = (unsigned)((krow < rag_rows ? krow : rag_rows - 1) * (int)p.ldk + kcol);
  const unsigned voff = (unsigned)(vrow * KVB + vcol);
  // LEAN / PIPE: the next tile's two source addresses live in scalar registers and advance by a constant (a piece boundary
  // recomputes them): ~10 scalar instructions per tile instead of ~50
  const unsigned char* kp_ = kh;
  const unsigned char* vp_ = p.vt + (int64_t)head * p.tpp * (D * KVB);
  // GATE: byte offset of the piece at sequence position POS_ from piece 0, and the bounded wait for its arrival flag.  Every wave waits
  // for itself (it issues its own share of a tile's requests); a wave that waits holds up nothing but the barrier two tiles on.
#define A8_PIECE_OFF(POS_) \
  (GATE ? (int64_t)p.seq_piece[(POS_)] * p.piece_stride + (p.seq_piece[(POS_)] == p.own_index ? p.own_delta : (int64_t)0) : (int64_t)(POS_) * p.piece_stride)
#define A8_GATE(POS_)                                                                                                     \
  if (GATE) {                                                                                                             \
    const int fi_ = p.seq_flag[(POS_)];                                                                                   \
    if (fi_ >= 0) {                                                                                                       \
      const unsigned want_ = p.seq_value[(POS_)];                                                                         \
      const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime();                                                    \
      while ((int)(__hip_atomic_load(p.flags + fi_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want_) < 0) {          \
        __builtin_amdgcn_s_sleep(16);                                                                                     \
        if (p.err && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;                   \
        if (p.timeout_ticks && __builtin_amdgcn_s_memrealtime() - t0_ > p.timeout_ticks) {                                \
          if (p.err) atomicCAS(p.err, 0u, 0x80000000u | (unsigned)(POS_));                                                \
          break;                                                                                                          \
        }                                                                                                                 \
      }                                                                                                                   \
    }                                                                                                                     \
  }
  if (GATE) {
    A8_GATE(0);
    const int64_t off0 = A8_PIECE_OFF(0);
    kp_ += off0;
    vp_ += off0;
  }
#define A8_DMA_TILE(T_)                                                                              \
  if (PIPE) {                                                                                        \
    const unsigned l0_ = lds_base + (unsigned)(((T_) & (NSTG - 1)) * STAGE_BYTES + wave * 1024);     \
    dma16s(kp_, d_tl == p.tpp - 1 ? koff_rag : koff_full, l0_);                                      \
    dma16s(vp_, voff, l0_ + KT_BYTES);                                                               \
    if ((T_) < nt - 1) {                                                                             \
      if (d_tl + 1 == p.tpp) {                                                                       \
        d_tl = 0;                                                                                    \
        d_pc = d_pc + 1;                                                                             \
        A8_GATE(d_pc);                                                                               \
        const int64_t po_ = A8_PIECE_OFF(d_pc);                                                      \
        kp_ = kh + po_;                                                                              \
        vp_ = p.vt + po_ + (int64_t)head * p.tpp * (D * KVB);                                        \
      } else {                                                                                       \
        d_tl = d_tl + 1;                                                                             \
        kp_ += (int64_t)KVB * p.ldk;                                                                 \
        vp_ += D * KVB;                                                                              \
      }                                                                                              \
    }                                                                                                \
  } else if (LEAN) {                                                                                 \
    const unsigned l0_ = lds_base + (unsigned)(((T_) & (NSTAGE - 1)) * STAGE_BYTES + wave * 1024);   \
    const int64_t pb_ = (int64_t)d_pc * p.piece_stride;                                              \
    dma16s(kh + pb_ + (int64_t)d_tl * KVB * p.ldk, d_tl == p.tpp - 1 ? koff_rag : koff_full, l0_);   \
    dma16s(p.vt + pb_ + ((int64_t)head * p.tpp + d_tl) * (D * KVB), voff, l0_ + KT_BYTES);           \
    /* the same advance as below in mask arithmetic: a boolean turned into an integer goes through a vector register */ \
    const int go_ = ((T_) - (nt - 1)) >> 31;            /* -1 while there is a next tile */              \
    const int in_ = (d_tl + 1 - p.tpp) >> 31;           /* -1 while the next tile is in the same piece */ \
    d_pc += (1 + in_) & go_;                                                                         \
    d_tl = (((d_tl + 1) & in_) & go_) | (d_tl & ~go_);                                               \
  } else {                                                                                           \
    int kr_ = d_tl * KVB + krow;                                                                     \
    kr_ = kr_ < p.piece_rows ? kr_ : p.piece_rows - 1;                                               \
    const unsigned l0_ = lds_base + (unsigned)(((T_) & (NSTAGE - 1)) * STAGE_BYTES + wave * 1024);   \
    const int64_t pb_ = (int64_t)d_pc * p.piece_stride;                                              \
    dma16(kh + pb_ + (int64_t)kr_ * p.ldk + kcol, l0_);                                              \
    dma16(p.vt + pb_ + (((int64_t)head * p.tpp + d_tl) * D + vrow) * KVB + vcol, l0_ + KT_BYTES);    \
    if ((T_) < nt - 1) {           /* past the last tile the last one is re-loaded (uniform DMA counts) */ \
      if (++d_tl == p.tpp) { d_tl = 0; ++d_pc; }                                                     \
    }                                                                                                \
  }
#define A8_VMCNT2() asm volatile("s_waitcnt vmcnt(2)" ::: "memory")
#define A8_BARRIER()                                          \
  do {                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                             \
    asm volatile("" ::: "memory");                            \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)

  const int grp = STAGGER ? (wave >> 2) : 0;
  if constexpr (!PIPE) {               // (the pipelined loop has its own prologue below)
    A8_DMA_TILE(0);
    A8_DMA_TILE(1);
    A8_VMCNT2();
    A8_BARRIER();
    if (grp == 1) {
      A8_DMA_TILE(2);
      A8_VMCNT2();
      A8_BARRIER();
    }
  }
  const int ahead = 2 + grp;

  // fragment read offsets: K row l31 (+32 per key block), chunks 4s + 2hi, +1;  V^T row d0*32 + l31, chunks 2hi, 2hi+1
  const int k_sw = (l31 >> 1) & 7;          // (key >> 1) & 7 is the same for key and key + 32
  const int k_row = l31 * 128;
  const int v_sw0 = (l31 >> 2) & 3;          // ((d0*32 + l31) >> 2) & 3 == (l31 >> 2) & 3

  int c_tl = 0;        // tile inside its piece of the tile being computed
  i32x8 pf = {0, 0, 0, 0, 0, 0, 0, 0};

  if constexpr (PIPE) {
    // ---- software-pipelined key loop (round 6) ----
    // One iteration = tile t's softmax (vector work) issued BETWEEN the matrix instructions of its neighbours: S(t+1) = K(t+1) Q^T and
    // O += V(t-1) P(t-1).  A wave's MFMAs run asynchronously to its own VALU stream, so the exponentials no longer wait on the matrix
    // pipe and the matrix pipe no longer waits for them (the un-pipelined loop relies on the OTHER wave of the SIMD being in the
    // opposite phase - which a per-tile barrier prevents: both start every tile together).  Ring use: tile t+2 is requested at the top
    // of iteration t into the stage K(t-2) left two iterations ago and V(t-2) left in the previous one; it has to have landed by the
    // end of the iteration (vmcnt(0): one iteration of latency cover - measured sufficient, the rows come from L2).
#define A8_QK(STG_, DST_)                                                                                                       \
    _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                                \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                                           \
      const int c0 = 4 * s + 2 * hi;                                                                                             \
      const char* rowp = smem + (STG_) * STAGE_BYTES + kb * 4096 + k_row;                                                        \
      const i32x8 kf = read32(rowp + ((c0 ^ k_sw) << 4), rowp + (((c0 + 1) ^ k_sw) << 4));                                       \
      DST_[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf[s], s == 0 ? cinit : DST_[kb], 0, 0, 0, sK, 0, sQ);      \
    }
#define A8_PV(STG_, PF_)                                                                                                        \
    _Pragma("unroll") for (int d0 = 0; d0 < 4; ++d0) {                                                                           \
      const char* rowp = smem + (STG_) * STAGE_BYTES + KT_BYTES + (d0 * 32 + l31) * 64;                                          \
      const i32x8 vf = read32(rowp + (((2 * hi) ^ v_sw0) << 4), rowp + (((2 * hi + 1) ^ v_sw0) << 4));                           \
      ot[d0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, PF_, ot[d0], 0, 0, 0, sV, 0, 127);                            \
    }
    // V of "tile -1" (the last stage) = zeros: the first iteration's O += V(-1) P(-1) adds 0 x 0 instead of branching around it
    *reinterpret_cast<i32x4*>(smem + (NSTG - 1) * STAGE_BYTES + KT_BYTES + tid * 16) = (i32x4){0, 0, 0, 0};
    A8_DMA_TILE(0);
    A8_DMA_TILE(1);
    if (TWO) A8_DMA_TILE(2);          // TWO: a pair (t, t+1) reads K(t+1), K(t+2), V(t-1), V(t) - tiles <= t + 2 are there when it starts
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    A8_BARRIER();
    f32x16 stx[2][2];
    i32x8 pfx[2] = {pf, pf};
    A8_QK(0, stx[0]);
    for (int t4 = 0; t4 < nt; t4 += NSTG) {
#pragma unroll
      for (int ti = 0; ti < NSTG; ++ti) {
        const int t = t4 + ti;
        if (t >= nt) break;
        f32x16(&st)[2] = stx[ti & 1];
        f32x16(&sn)[2] = stx[(ti + 1) & 1];
        const int key0 = c_tl * KVB;
        {
          const int in_ = (c_tl + 1 - p.tpp) >> 31;
          c_tl = (c_tl + 1) & in_;
        }
        if (key0 + KVB > p.piece_rows) {               // the ragged last tile of a piece (S(t) was finished an iteration ago)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
              if (key >= p.piece_rows) st[kb][r] = NEG_BIG;
            }
        }
        if (!TWO) {
          A8_DMA_TILE(t + 2);
        } else if ((ti & 1) == 0) {      // requested at the top of the pair into the stages V(t-5), V(t-4) left two pairs ago
          A8_DMA_TILE(t + 3);
          A8_DMA_TILE(t + 4);
        }
        const bool no_ref = m_run < -1.0e29f;
        // ---- the pipelined block: 8 segments = one MFMA each (S(t+1): two first halves, two of O += V(t-1) P(t-1), S(t+1): second
        // halves - four MFMAs after the ones they accumulate on - , the other two of O), the NEXT segment's fragment read, and a tenth of
        // tile t's softmax in the MFMA's shadow: 4 exponentials, their row-sum adds, their e4m3 packing ----
        float pv[2][16];
        float ps = 0.f;
        f32x2 ps2 = {0.f, 0.f};
        i32x8& pn = pfx[ti & 1];
        const i32x8 pp = pfx[(ti + 1) & 1];
        const char* const kst = smem + ((ti + 1) & (NSTG - 1)) * STAGE_BYTES + k_row;
        const char* const vst = smem + ((ti + NSTG - 1) & (NSTG - 1)) * STAGE_BYTES + KT_BYTES + l31 * 64;
        auto frag = [&](int g) -> i32x8 {
          if (g == 0 || g == 1 || g == 4 || g == 5) {           // K fragment: key block kb = g & 1, k-step s = g >> 2
            const int c0 = 4 * (g >> 2) + 2 * hi;
            const char* rowp = kst + (g & 1) * 4096;
            return read32(rowp + ((c0 ^ k_sw) << 4), rowp + (((c0 + 1) ^ k_sw) << 4));
          }
          const int d0 = (g & 1) + ((g >> 2) << 1);             // V^T fragment: d block 0, 1 (g = 2, 3) and 2, 3 (g = 6, 7)
          const char* rowp = vst + d0 * 32 * 64;
          return read32(rowp + (((2 * hi) ^ v_sw0) << 4), rowp + (((2 * hi + 1) ^ v_sw0) << 4));
        };
        i32x8 fr = frag(0), fr1 = fr;
        if (PFD2) fr1 = frag(1);
        if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          i32x8 nfr = fr;
          if (g + (PFD2 ? 2 : 1) < 8) nfr = frag(g + (PFD2 ? 2 : 1));
          if (g == 0 || g == 1) sn[g] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fr, qf[0], cinit, 0, 0, 0, sK, 0, sQ);
          else if (g == 4 || g == 5) sn[g - 4] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fr, qf[1], sn[g - 4], 0, 0, 0, sK, 0, sQ);
          else {
            const int d0 = (g & 1) + ((g >> 2) << 1);
            ot[d0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fr, pp, ot[d0], 0, 0, 0, sV, 0, 127);
          }
          const int kb = g >> 2, r0 = (g & 3) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) pv[kb][r0 + j] = __builtin_amdgcn_exp2f(st[kb][r0 + j]);
          if (PKADD) {
            ps2 += (f32x2){pv[kb][r0], pv[kb][r0 + 1]};
            ps2 += (f32x2){pv[kb][r0 + 2], pv[kb][r0 + 3]};
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) ps += pv[kb][r0 + j];
          }
          pn[g] = pack_fp8x4_over(pv[kb][r0], pv[kb][r0 + 1], pv[kb][r0 + 2], pv[kb][r0 + 3], pn[g]);
          if (PFD2) { fr = fr1; fr1 = nfr; } else fr = nfr;
          __builtin_amdgcn_sched_barrier(0);
        }
        if (SETPRIO) __builtin_amdgcn_s_setprio(0);
        if (PKADD) ps = ps2[0] + ps2[1];
        if (__any(!(ps <= p_lim) || no_ref)) {         // rare: re-base on the tile's true maximum (after O += V(t-1) P(t-1) was issued)
          float mloc = st[0][0];
#pragma unroll
          for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[0][r]);
#pragma unroll
          for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[1][r]);
          mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) + m_base;
          const float m_new = fmaxf(m_run, mloc);
          const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
          m_run = m_new;
          l_run *= alpha;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
          const float dm = m_new - m_base;
          m_base = m_new;
#pragma unroll
          for (int r = 0; r < 16; ++r) cinit[r] = -m_new;
          ps = 0.f;
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              pv[kb][r] = __builtin_amdgcn_exp2f(st[kb][r] - dm);
              ps += pv[kb][r];
              sn[kb][r] -= dm;                          // S(t+1) was started against the old reference
            }
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 4) pn[kb * 4 + (r >> 2)] = pack_fp8x4_over(pv[kb][r], pv[kb][r + 1], pv[kb][r + 2], pv[kb][r + 3], pn[kb * 4 + (r >> 2)]);
        }
        l_run += ps;
        if (!TWO || (ti & 1)) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          A8_BARRIER();
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (TWO, odd tile count: the last pair's requests were not waited for)
    {                                                   // the last tile's O += V P
      const int lt = (nt - 1) & (NSTG - 1);
      i32x8 pl;
#pragma unroll
      for (int i = 0; i < 8; ++i) pl[i] = ((nt - 1) & 1) ? pfx[1][i] : pfx[0][i];
      A8_PV(lt, pl);
    }
    attc::store_result(p.c, q0 + l31, head, hi, ot, m_run, l_run);
    return;
#undef A8_QK
#undef A8_PV
  }
  // LEAN: the ring position is a compile-time constant in each of NSTAGE unrolled bodies, so every ds_read address is a lane
  // register (fixed for the launch) plus an immediate: no vector address arithmetic per tile
  for (int t4 = 0; t4 < nt; t4 += (LEAN ? NSTAGE : 1))
#pragma unroll
  for (int ti = 0; ti < (LEAN ? NSTAGE : 1); ++ti) {
    const int t = t4 + ti;
    if (LEAN && t >= nt) break;
    const char* ks = smem + (LEAN ? ti : (t & (NSTAGE - 1))) * STAGE_BYTES;
    const char* vs = ks + KT_BYTES;
    const int key0 = c_tl * KVB;                   // first key of this tile inside its piece
    if (++c_tl == p.tpp) c_tl = 0;
    A8_DMA_TILE(t + ahead);

    // ---- S^T = Kq Qq^T (4 MFMAs), scales in the MFMA, reference in the C operand of the first step ----
    f32x16 st[2];
    const bool no_ref = m_run < -1.0e29f;
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int c0 = 4 * s + 2 * hi;
        const char* rowp = ks + kb * 4096 + k_row;
        const i32x8 kf = read32(rowp + ((c0 ^ k_sw) << 4), rowp + (((c0 + 1) ^ k_sw) << 4));
        st[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf[s], s == 0 ? cinit : st[kb], 0, 0, 0, sK, 0, sQ);
      }
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    if (key0 + KVB > p.piece_rows) {               // the ragged last tile of a piece: its padding keys do not exist
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= p.piece_rows) st[kb][r] = NEG_BIG;
        }
    }
    // ---- lazy-max softmax at unit scale (attn2.hip / attn7.hip) over the whole 64-key tile: P = exp2(S) against
    // the reference in the accumulators; its partial row sum bounds every P, so only when it exceeds 2^thr (or there is
    // no reference yet) the tile's true max is taken, O / l rescaled, the scores re-based and P recomputed ----
    float pv[2][16];
    float ps = 0.f;
    if (LEAN) {                       // two running partial sums per lane: v_pk_add_f32 adds a pair of P per instruction
      f32x2 ps2 = {0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          pv[kb][r] = __builtin_amdgcn_exp2f(st[kb][r]);
          pv[kb][r + 1] = __builtin_amdgcn_exp2f(st[kb][r + 1]);
          ps2 += (f32x2){pv[kb][r], pv[kb][r + 1]};
        }
      ps = ps2[0] + ps2[1];
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pv[kb][r] = __builtin_amdgcn_exp2f(st[kb][r]);
          ps += pv[kb][r];
        }
    }
    if (__any(!(ps <= p_lim) || no_ref)) {
      float mloc = st[0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[1][r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) + m_base;       // st = s - m_base
      const float m_new = fmaxf(m_run, mloc);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
      const float dm = m_new - m_base;
      m_base = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) cinit[r] = -m_new;
      ps = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pv[kb][r] = __builtin_amdgcn_exp2f(st[kb][r] - dm);
          ps += pv[kb][r];
        }
    }
    l_run += ps;
    // P^T operand: the lane's 32 keys in k-slot order j = kb*16 + r (tile_key), 4 e4m3 per dword
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 4)
        pf[kb * 4 + (r >> 2)] = LEAN ? pack_fp8x4_over(pv[kb][r], pv[kb][r + 1], pv[kb][r + 2], pv[kb][r + 3], pf[kb * 4 + (r >> 2)])
                                     : (int)pack_fp8x4(pv[kb][r], pv[kb][r + 1], pv[kb][r + 2], pv[kb][r + 3]);
    // ---- O^T += Vt P^T : one MFMA per 32-row d block ----
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int d0 = 0; d0 < 4; ++d0) {
      const char* rowp = vs + (d0 * 32 + l31) * 64;
      const i32x8 vf = read32(rowp + (((2 * hi) ^ v_sw0) << 4), rowp + (((2 * hi + 1) ^ v_sw0) << 4));
      ot[d0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf, ot[d0], 0, 0, 0, sV, 0, 127);
    }
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);

    A8_VMCNT2();
    A8_BARRIER();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (grp == 0 && STAGGER) A8_BARRIER();

  attc::store_result(p.c, q0 + l31, head, hi, ot, m_run, l_run);
}

template <int VAR>
int launch(const Params& p, hipStream_t st) {
  static icv_dev_flags attr_set = {};
  constexpr int lds = ((VAR & 512) ? 8 : NSTAGE) * STAGE_BYTES;
  if (int rc = icv_ensure_dynamic_lds((const void*)attn8_kernel<VAR>, lds, &attr_set, "attn8")) return rc;
  const int64_t nwg = (int64_t)p.heads * p.nqb;
  hipLaunchKernelGGL(attn8_kernel<VAR>, dim3((unsigned)nwg), dim3(512), lds, st, p);
  return icv_check_launch("icv_attention_fp8_fwd");
}

}  // namespace att8

extern "C" int64_t icv_attention_fp8_vt_bytes(int64_t Skv, int64_t heads) {
  return heads * ((Skv + att8::KVB - 1) / att8::KVB) * (int64_t)(att8::D * att8::KVB);
}

extern "C" int icv_attention_fp8_prepare(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                         int64_t Sq, int64_t Skv, int64_t heads, void* qq, int64_t ldqq, void* kq,
                                         int64_t ldkq, void* vt, float* amax, void* stream) {
  ICV_REQUIRE(amax && (q || k), "icv_attention_fp8_prepare: null pointer");
  ICV_REQUIRE((q == nullptr) == (qq == nullptr), "icv_attention_fp8_prepare: q and qq go together");
  ICV_REQUIRE((k == nullptr) == (kq == nullptr) && (k == nullptr) == (v == nullptr) && (k == nullptr) == (vt == nullptr),
              "icv_attention_fp8_prepare: k, v, kq and vt go together");
  ICV_REQUIRE(heads > 0 && (q == nullptr || Sq > 0) && (k == nullptr || Skv > 0), "icv_attention_fp8_prepare: empty problem");
  ICV_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldqq % 16 == 0 && ldkq % 16 == 0, "icv_attention_fp8_prepare: leading dims must keep 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  int* ab = reinterpret_cast<int*>(amax);
  if (q) {
    ICV_REQUIRE(hipMemsetAsync(amax, 0, sizeof(float) * heads, st) == hipSuccess, "icv_attention_fp8_prepare: memset failed");
    const dim3 grid((unsigned)heads, (unsigned)((Sq + 255) / 256));
    hipLaunchKernelGGL(att8::amax_kernel, grid, dim3(256), 0, st, (const bf16_t*)q, ldq, Sq, ab);
    hipLaunchKernelGGL(att8::quant_rows_kernel, grid, dim3(256), 0, st, (const bf16_t*)q, ldq, Sq, amax, (unsigned char*)qq, ldqq);
  }
  if (k) {
    ICV_REQUIRE(hipMemsetAsync(amax + heads, 0, sizeof(float) * 2 * heads, st) == hipSuccess, "icv_attention_fp8_prepare: memset failed");
    const dim3 grid((unsigned)heads, (unsigned)((Skv + 255) / 256));
    const int nt = (int)((Skv + att8::KVB - 1) / att8::KVB);
    hipLaunchKernelGGL(att8::amax_kernel, grid, dim3(256), 0, st, (const bf16_t*)k, ldk, Skv, ab + heads);
    hipLaunchKernelGGL(att8::amax_kernel, grid, dim3(256), 0, st, (const bf16_t*)v, ldv, Skv, ab + 2 * heads);
    hipLaunchKernelGGL(att8::quant_rows_kernel, grid, dim3(256), 0, st, (const bf16_t*)k, ldk, Skv, amax + heads, (unsigned char*)kq, ldkq);
    hipLaunchKernelGGL(att8::quant_vt_kernel, dim3((unsigned)nt, (unsigned)heads), dim3(256), 0, st, (const bf16_t*)v, ldv, Skv, amax + 2 * heads, (unsigned char*)vt, nt);
  }
  return icv_check_launch("icv_attention_fp8_prepare");
}

static int attn8_run(const void* qq, int64_t ldqq, const void* kq, int64_t ldkq, const void* vt, const float* amax, void* o,
                     int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in, int state_out, int64_t Sq, int64_t piece_rows,
                     int64_t n_pieces, int64_t piece_stride, int64_t heads, hipStream_t st, const att8::Params* gate = nullptr) {
  att8::Params p;
  p.flags = nullptr; p.err = nullptr; p.timeout_ticks = 0; p.own_delta = 0; p.own_index = -1; p.n_seq = 0;
  if (gate) {
    p.flags = gate->flags; p.err = gate->err; p.timeout_ticks = gate->timeout_ticks; p.own_delta = gate->own_delta; p.own_index = gate->own_index;
    p.n_seq = gate->n_seq;
    for (int i = 0; i < gate->n_seq; ++i) { p.seq_piece[i] = gate->seq_piece[i]; p.seq_flag[i] = gate->seq_flag[i]; p.seq_value[i] = gate->seq_value[i]; }
  }
  const int64_t Skv = piece_rows * n_pieces;
  p.q = (const unsigned char*)qq; p.ldq = ldqq; p.k = (const unsigned char*)kq; p.ldk = ldkq; p.vt = (const unsigned char*)vt;
  p.amax = amax; p.Sq = Sq; p.Skv = Skv; p.heads = (int)heads;
  p.nqb = (int)((Sq + att8::QB - 1) / att8::QB);
  p.tpp = (int)((piece_rows + att8::KVB - 1) / att8::KVB);
  p.piece_rows = (int)piece_rows;
  p.piece_stride = piece_stride;
  p.ntiles = (int)(p.tpp * n_pieces);
  p.thr = (float)icv_get_option_int("attn_defer_max_log2", 8);
  p.c = attc::Params{};
  p.c.o = (bf16_t*)o; p.c.ldo = ldo; p.c.acc = acc; p.c.ldacc = ldacc; p.c.ml = ml; p.c.state_in = state_in; p.c.state_out = state_out;
  p.c.Sq = Sq; p.c.Skv = Skv; p.c.heads = (int)heads; p.c.nqb = p.nqb; p.c.sc = 1.0f; p.c.thr = p.thr;
  ICV_REQUIRE((int64_t)p.heads * p.nqb < (1LL << 31) && piece_rows < (1LL << 30), "icv_attention_fp8_fwd: grid too large");
  // default (-1) = 164: the software-pipelined loop, fragment reads two segments ahead, s_setprio(1) over the pipelined block
  // (+10...12 % over round 5's loop = variant 0 at S = 37 440 / 86 400, +4 % at 512 text keys, bit-identical outputs:
  // profiles/r06/attn8_pipelined_ab.txt)
  if (gate) return att8::launch<164 | 256>(p, st);
  int variant = icv_get_option_int("attn8_variant", -1);
  if (variant < 0) variant = 164;
  switch (variant) {
    case 0: return att8::launch<0>(p, st);
    case 1: return att8::launch<1>(p, st);
    case 4: return att8::launch<4>(p, st);
    case 5: return att8::launch<5>(p, st);
    case 8: return att8::launch<8>(p, st);
    case 9: return att8::launch<9>(p, st);
    case 16: return att8::launch<16>(p, st);
    case 32: return att8::launch<32>(p, st);
    case 96: return att8::launch<96>(p, st);
    case 36: return att8::launch<36>(p, st);
    case 160: return att8::launch<160>(p, st);
    case 164: return att8::launch<164>(p, st);
    case 676: return att8::launch<676>(p, st);
  }
  icv_set_error("icv_attention_fp8_fwd: unknown attn8_variant");
  return 1;
}

extern "C" int icv_attention_fp8_fwd(const void* qq, int64_t ldqq, const void* kq, int64_t ldkq, const void* vt,
                                     const float* amax, void* o, int64_t ldo, int64_t Sq, int64_t Skv, int64_t heads,
                                     void* stream) {
  ICV_REQUIRE(qq && kq && vt && amax && o, "icv_attention_fp8_fwd: null pointer");
  ICV_REQUIRE(Sq > 0 && Skv > 0 && heads > 0, "icv_attention_fp8_fwd: empty problem");
  ICV_REQUIRE(ldqq % 16 == 0 && ldkq % 16 == 0 && ldo % 4 == 0, "icv_attention_fp8_fwd: leading dims must keep 16-byte row alignment");
  return attn8_run(qq, ldqq, kq, ldkq, vt, amax, o, ldo, nullptr, 0, nullptr, 0, 0, Sq, Skv, 1, 0, heads, (hipStream_t)stream);
}

extern "C" int icv_attention_fp8_fwd_chunk(const void* qq, int64_t ldqq, const void* kq, int64_t ldkq, const void* vt,
                                           const float* amax, void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml,
                                           int64_t Sq, int64_t Skv, int64_t heads, int first, int last, void* stream) {
  ICV_REQUIRE(qq && kq && vt && amax, "icv_attention_fp8_fwd_chunk: null pointer");
  ICV_REQUIRE(Sq > 0 && Skv > 0 && heads > 0, "icv_attention_fp8_fwd_chunk: empty problem");
  ICV_REQUIRE(ldqq % 16 == 0 && ldkq % 16 == 0, "icv_attention_fp8_fwd_chunk: leading dims must keep 16-byte row alignment");
  ICV_REQUIRE((first && last) || (acc && ml && ldacc % 4 == 0), "icv_attention_fp8_fwd_chunk: carried state buffers required unless first && last");
  ICV_REQUIRE(!last || (o && ldo % 4 == 0), "icv_attention_fp8_fwd_chunk: output required for the last chunk");
  return attn8_run(qq, ldqq, kq, ldkq, vt, amax, o, ldo, acc, ldacc, ml, first ? 0 : 1, last ? 0 : 1, Sq, Skv, 1, 0, heads, (hipStream_t)stream);
}

// ---- e4m3 K|V ON THE WIRE (sequence parallel, BASELINE.json config #5) ------------------------------------------------------
// Each rank quantises its OWN rows of a K|V chunk once and ships e4m3 bytes (half the xGMI traffic of bf16 rows, 1/world of the
// quantise work of "gather bf16, quantise the gathered chunk on every rank").  For that the scales must be known before the
// exchange: icv_attention_fp8_kv_amax gives this rank's per-head abs-max of its whole K and V shard, the host max-reduces the
// 2 x H floats over the sequence-parallel group (one tiny collective per layer), and every rank then quantises with the SAME
// per-head scale — the scale of the unsharded launch, so the e4m3 values are exactly the single-GPU ones.
// A rank's piece of a chunk ("blob") = [ kq: rows_pad x (H*128) e4m3 row-major | vt: [H][rows_pad / 64][128][64] ],
// rows_pad = rows rounded up to 64; blob bytes = 2 * rows_pad * H * 128 (icv_attention_fp8_blob_bytes).  The gathered chunk is
// `n_pieces` blobs back to back (rank-major) and icv_attention_fp8_fwd_pieces consumes it in place.
extern "C" int64_t icv_attention_fp8_blob_bytes(int64_t rows, int64_t heads) {
  const int64_t rp = (rows + att8::KVB - 1) / att8::KVB * att8::KVB;
  return 2 * rp * heads * att8::D;
}

extern "C" int icv_attention_fp8_kv_amax(const void* k, int64_t ldk, const void* v, int64_t ldv, int64_t rows, int64_t heads, float* amax,
                                         void* stream) {
  ICV_REQUIRE(k && v && amax && rows > 0 && heads > 0, "icv_attention_fp8_kv_amax: null pointer / empty problem");
  ICV_REQUIRE(ldk % 8 == 0 && ldv % 8 == 0, "icv_attention_fp8_kv_amax: leading dims must keep 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  ICV_REQUIRE(hipMemsetAsync(amax + heads, 0, sizeof(float) * 2 * heads, st) == hipSuccess, "icv_attention_fp8_kv_amax: memset failed");
  const dim3 grid((unsigned)heads, (unsigned)((rows + 255) / 256));
  int* ab = reinterpret_cast<int*>(amax);
  hipLaunchKernelGGL(att8::amax_kernel, grid, dim3(256), 0, st, (const bf16_t*)k, ldk, rows, ab + heads);
  hipLaunchKernelGGL(att8::amax_kernel, grid, dim3(256), 0, st, (const bf16_t*)v, ldv, rows, ab + 2 * heads);
  return icv_check_launch("icv_attention_fp8_kv_amax");
}

extern "C" int icv_attention_fp8_quantize_kv(const void* k, int64_t ldk, const void* v, int64_t ldv, int64_t rows, int64_t heads,
                                             const float* amax, void* blob, void* stream) {
  ICV_REQUIRE(k && v && amax && blob && rows > 0 && heads > 0, "icv_attention_fp8_quantize_kv: null pointer / empty problem");
  ICV_REQUIRE(ldk % 8 == 0 && ldv % 8 == 0, "icv_attention_fp8_quantize_kv: leading dims must keep 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  const int nt = (int)((rows + att8::KVB - 1) / att8::KVB);
  const int64_t ldkq = heads * att8::D;
  unsigned char* kq = (unsigned char*)blob;
  unsigned char* vt = kq + (int64_t)nt * att8::KVB * ldkq;
  const dim3 grid((unsigned)heads, (unsigned)((rows + 255) / 256));
  hipLaunchKernelGGL(att8::quant_rows_kernel, grid, dim3(256), 0, st, (const bf16_t*)k, ldk, rows, amax + heads, kq, ldkq);
  hipLaunchKernelGGL(att8::quant_vt_kernel, dim3((unsigned)nt, (unsigned)heads), dim3(256), 0, st, (const bf16_t*)v, ldv, rows, amax + 2 * heads, vt, nt);
  return icv_check_launch("icv_attention_fp8_quantize_kv");
}

extern "C" int icv_attention_fp8_fwd_pieces(const void* qq, int64_t ldqq, const void* blobs, int64_t piece_rows, int64_t n_pieces,
                                            const float* amax, void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int64_t Sq,
                                            int64_t heads, int first, int last, void* stream) {
  ICV_REQUIRE(qq && blobs && amax, "icv_attention_fp8_fwd_pieces: null pointer");
  ICV_REQUIRE(Sq > 0 && piece_rows > 0 && n_pieces > 0 && heads > 0, "icv_attention_fp8_fwd_pieces: empty problem");
  ICV_REQUIRE(ldqq % 16 == 0, "icv_attention_fp8_fwd_pieces: leading dims must keep 16-byte row alignment");
  ICV_REQUIRE((first && last) || (acc && ml && ldacc % 4 == 0), "icv_attention_fp8_fwd_pieces: carried state buffers required unless first && last");
  ICV_REQUIRE(!last || (o && ldo % 4 == 0), "icv_attention_fp8_fwd_pieces: output required for the last chunk");
  const int64_t rp = (piece_rows + att8::KVB - 1) / att8::KVB * att8::KVB;
  const int64_t ldkq = heads * att8::D;
  const unsigned char* kq = (const unsigned char*)blobs;
  return attn8_run(qq, ldqq, kq, ldkq, kq + rp * ldkq, amax, o, ldo, acc, ldacc, ml, first ? 0 : 1, last ? 0 : 1, Sq, piece_rows, n_pieces,
                   icv_attention_fp8_blob_bytes(piece_rows, heads), heads, (hipStream_t)stream);
}

// The same over pieces that may still be ARRIVING (the sequence-parallel schedule of SURVEY.md §8e for the e4m3 wire format: "process K/V
// chunks in arrival order (own shard first)"): the launch does not wait for the chunk's exchange on the host; it walks the blobs in the
// order seq_piece[0..n_pieces) and every position gates, inside the kernel, on its arrival flag (seq_flag[i] < 0: present at launch) -
// the words the copy-engine transport raises per peer (icv_ipc_arrival) or the one a side stream writes behind a collective
// (icv_flag_write).  `own_blob` (may be NULL): the blob of piece `own_index` is read THERE instead of from its slot in `blobs`.
// seq_* are host arrays.  Time-out / error word: icv_attention_fwd_pieces' rule.
extern "C" int icv_attention_fp8_fwd_pieces_gated(const void* qq, int64_t ldqq, const void* blobs, int64_t piece_rows, int64_t n_pieces,
                                                  const void* own_blob, int64_t own_index, const int32_t* seq_piece, const int32_t* seq_flag,
                                                  const uint32_t* seq_value, const uint32_t* flags, uint32_t* err, int64_t timeout_us,
                                                  const float* amax, void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int64_t Sq,
                                                  int64_t heads, int first, int last, void* stream) {
  ICV_REQUIRE(qq && blobs && amax && seq_piece && seq_flag && seq_value, "icv_attention_fp8_fwd_pieces_gated: null pointer");
  ICV_REQUIRE(Sq > 0 && piece_rows > 0 && n_pieces > 0 && n_pieces <= ICV_ATTN_MAX_PIECES && heads > 0,
              "icv_attention_fp8_fwd_pieces_gated: empty problem or more than %d pieces", ICV_ATTN_MAX_PIECES);
  ICV_REQUIRE(ldqq % 16 == 0, "icv_attention_fp8_fwd_pieces_gated: leading dims must keep 16-byte row alignment");
  ICV_REQUIRE((first && last) || (acc && ml && ldacc % 4 == 0), "icv_attention_fp8_fwd_pieces_gated: carried state buffers required unless first && last");
  ICV_REQUIRE(!last || (o && ldo % 4 == 0), "icv_attention_fp8_fwd_pieces_gated: output required for the last chunk");
  ICV_REQUIRE(!own_blob || (own_index >= 0 && own_index < n_pieces && (uintptr_t)own_blob % 16 == 0), "icv_attention_fp8_fwd_pieces_gated: bad own blob");
  const int64_t rp = (piece_rows + att8::KVB - 1) / att8::KVB * att8::KVB;
  const int64_t ldkq = heads * att8::D;
  const int64_t stride = icv_attention_fp8_blob_bytes(piece_rows, heads);
  const unsigned char* kq = (const unsigned char*)blobs;
  att8::Params g;
  g.flags = flags; g.err = err; g.timeout_ticks = timeout_us > 0 ? (unsigned long long)timeout_us * 100ull : 0ull;
  g.own_index = own_blob ? (int)own_index : -1;
  g.own_delta = own_blob ? (int64_t)((const unsigned char*)own_blob - (kq + own_index * stride)) : 0;
  g.n_seq = (int)n_pieces;
  unsigned long long seen = 0;
  for (int64_t i = 0; i < n_pieces; ++i) {
    ICV_REQUIRE(seq_piece[i] >= 0 && seq_piece[i] < n_pieces && !((seen >> seq_piece[i]) & 1ull),
                "icv_attention_fp8_fwd_pieces_gated: seq_piece is not a permutation of the pieces");
    ICV_REQUIRE(seq_flag[i] < 0 || flags, "icv_attention_fp8_fwd_pieces_gated: position %lld waits for flag %d but no flag array was given", (long long)i,
                seq_flag[i]);
    seen |= 1ull << seq_piece[i];
    g.seq_piece[i] = seq_piece[i]; g.seq_flag[i] = seq_flag[i]; g.seq_value[i] = seq_value[i];
  }
  return attn8_run(qq, ldqq, kq, ldkq, kq + rp * ldkq, amax, o, ldo, acc, ldacc, ml, first ? 0 : 1, last ? 0 : 1, Sq, piece_rows, n_pieces, stride, heads,
                   (hipStream_t)stream, &g);
}
