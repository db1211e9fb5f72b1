// K6 for the sequence-parallel schedule ("process K/V chunks in arrival order (own shard first) with online-softmax
// merging"): ONE launch per layer whose work-groups walk a list of K|V PIECES - this rank's own rows first, then every (peer, row
// chunk) in the order the exchange delivers them - and gate on an ARRIVAL FLAG per piece instead of the host launching one
// carried-state kernel per chunk.  What that removes, per layer and forward: C - 1 launches with their per-XCD tails, the
// C - 1 round trips of the carried (O, m, l) state through HBM, and the all-peers barrier in front of every chunk
// (profiles/r05/attn_round_occupancy.md; chunks 4 -> 2 alone was worth 2 % of a sequence-parallel step).
//
// The inner loop is attn7.hip's default schedule (variant 148/132: 8 waves x 32 query rows, 64-key tiles, LDS-DMA ring used as two
// halves of two tiles, ONE vmcnt(0) + barrier per 128 keys, lazy max, unit scale, s_setprio) - see there for the fragment / swizzle
// conventions.  New here:
//   * the key axis is a list of pieces {k, v, rows, flag, value} carried in the kernel arguments; the softmax state (O, m, l and
//     the reference baked into the MFMA's C operand) stays in registers across pieces;
//   * the ring runs THROUGH a piece boundary when the next piece is already there: wave 0 samples the next piece's flag once per
//     128-key interval (one uncached dword load whose latency hides behind the interval's own vmcnt(0)), publishes the verdict
//     through an LDS word in front of the interval's barrier, and the last interval of a piece requests the next piece's first two
//     tiles instead of its own (dead) past-the-end tiles.  The verdict is taken by ONE lane and read by all waves after a barrier:
//     every wave issues its share of a tile's DMA, so a per-wave opinion about "ready" would tear a tile;
//   * a piece that has NOT arrived when its predecessor ends costs a bubble: wave 0 spins on the flag (s_sleep between polls, bounded
//     by a time-out that sets an error word and lets the kernel finish with garbage rather than hang the queue; once the word is set no
//     waiter of this or a later launch waits out the deadline again), then the ring is refilled;
//   * no cache maintenance is needed for the late rows: the rows of a piece are first read after its flag was seen, the launch's own
//     acquire invalidated whatever an earlier launch left in L2 / L1, and pieces are whole rows (no cache line straddles two pieces),
//     so no stale line of a piece can exist; the flag itself is polled with system-scope (uncached) loads.
// Replaces: the chunked icv_attention_fwd_chunk sequence of the sequence-parallel self-attention (the fork's / xDiT-USP's gather-then-
// flash_attention [EXT]); the reference itself has no such path (one GPU, [R infinicube/inference/guidance_buffer_generation.py:759-766]).
#include "attn_common.h"

namespace att7p {

using attc::D;
using attc::NEG_BIG;
using attc::lds_read_tr16;
constexpr int KVB = 64;
constexpr int TILE_BYTES = KVB * D * 2;      // 16 KiB (K or V)
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // 32 KiB
constexpr int NST = 4, NW = 8, NI = 16 / NW, QB = NW * 32;
constexpr int LDS_BYTES = NST * STAGE_BYTES + 64;   // ring + the verdict words
constexpr int MAX_PIECES = ICV_ATTN_MAX_PIECES;

struct Piece {
  const bf16_t* k;
  const bf16_t* v;
  int rows;
  int flag;          // index into Params::flags; < 0: the rows are there when the launch starts
  unsigned value;    // arrived when (int)(flags[flag] - value) >= 0
  int pad;
};

struct Params {
  attc::Params a;                      // q / o / strides / heads / nqb / scale / trace (k, v, Skv unused)
  const unsigned* flags;
  unsigned* err;                       // first time-out wins: 0x80000000 | piece index
  unsigned long long timeout_ticks;    // 100 MHz s_memrealtime ticks; 0 = wait for ever
  unsigned long long* ptrace;          // diagnostics: [work-group][piece] tick at which the piece's first tile was started
  int n_pieces;
  Piece piece[MAX_PIECES];
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

// uncached (system-scope) dword load WITHOUT a compiler-inserted wait: the caller consumes the value after its own s_waitcnt vmcnt(0)
__device__ __forceinline__ unsigned load_flag_async(const unsigned* f) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(f) : "memory");
  return v;
}

template <bool UNIT>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2))) void attn7p_kernel(Params pp) {
  const attc::Params& p = pp.a;
  const float p_lim = __builtin_amdgcn_exp2f(p.thr);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  volatile unsigned* verdict = reinterpret_cast<volatile unsigned*>(smem + NST * STAGE_BYTES);   // [2]: double-buffered by interval parity
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
  int64_t qr_c = q0 + l31;
  qr_c = qr_c < p.Sq ? qr_c : p.Sq - 1;

  // ---- softmax state (registers, across all pieces) ----
  f32x16 ot[4];
  float m_run, l_run;
  attc::load_state(p, qr_c, head, hi, ot, m_run, l_run);
  float m_base = m_run < -1.0e29f ? 0.f : m_run;
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
  __builtin_amdgcn_s_waitcnt(0x0F70);     // retire the VGPR-destination loads before any LDS-DMA is in flight (cdna guide §5 trap (b))

  // ---- LDS-DMA lane mapping (attn7.hip) ----
  const int pc = lane & 15;
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  int dkey[NI], kcol[NI], vcol[NI];
  unsigned ko[NI], vo[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    dkey[j] = (wave * NI + j) * 4 + (lane >> 4);
    kcol[j] = (pc ^ (dkey[j] & 15)) * 8;
    vcol[j] = (pc ^ ((dkey[j] & 3) << 2)) * 8;
    ko[j] = (unsigned)(((int64_t)dkey[j] * p.ldk + kcol[j]) * 2);
    vo[j] = (unsigned)(((int64_t)dkey[j] * p.ldv + vcol[j]) * 2);
  }

  // ---- the piece the DMA side reads (wave-uniform) ----
  const bf16_t* kh_d = nullptr;
  const bf16_t* vh_d = nullptr;
  int skv_d = 0, nt_d = 0;
#define P_SET_DMA_PIECE(J_)                                              \
  {                                                                      \
    const Piece& pc_ = pp.piece[(J_)];                                   \
    kh_d = pc_.k + (int64_t)head * D;                                    \
    vh_d = pc_.v + (int64_t)head * D;                                    \
    skv_d = __builtin_amdgcn_readfirstlane(pc_.rows);                    \
    nt_d = (skv_d + KVB - 1) / KVB;                                      \
  }
  // tile T_ of the DMA piece (clamped to its last tile: a request past the end re-reads that tile into a dead stage) -> ring stage STG_
#define P_DMA_TILE(T_, STG_)                                                                         \
  {                                                                                                  \
    const int tt_ = (T_) < nt_d ? (T_) : nt_d - 1;                                                   \
    const unsigned l0_ = lds_base + (unsigned)(((STG_) & (NST - 1)) * STAGE_BYTES + (wave * NI) * 1024); \
    if ((tt_ + 1) * KVB <= skv_d) {                                                                  \
      const bf16_t* kt_ = kh_d + (int64_t)tt_ * KVB * p.ldk;                                         \
      const bf16_t* vt_ = vh_d + (int64_t)tt_ * KVB * p.ldv;                                         \
      _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_) dma16s(kt_, ko[j_], l0_ + j_ * 1024);        \
      _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_) dma16s(vt_, vo[j_], l0_ + TILE_BYTES + j_ * 1024); \
    } else {                                                                                         \
      _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_) {                                            \
        int64_t r_ = (int64_t)tt_ * KVB + dkey[j_];                                                  \
        r_ = r_ < skv_d ? r_ : skv_d - 1;                                                            \
        dma16(kh_d + r_ * p.ldk + kcol[j_], l0_ + j_ * 1024);                                        \
        dma16(vh_d + r_ * p.ldv + vcol[j_], l0_ + TILE_BYTES + j_ * 1024);                           \
      }                                                                                              \
    }                                                                                                \
  }
#define P_BARRIER()                                           \
  do {                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                             \
    asm volatile("" ::: "memory");                            \
    __builtin_amdgcn_sched_barrier(0);                        \
  } while (0)
  // blocking wait for piece J_ (a bubble): ONE lane polls, everybody else parks at the barrier; bounded by the time-out
#define P_WAIT_PIECE(J_)                                                                                          \
  {                                                                                                               \
    const int fi_ = __builtin_amdgcn_readfirstlane(pp.piece[(J_)].flag);                                          \
    if (fi_ >= 0) {                                                                                               \
      if (tid == 0) {                                                                                             \
        const unsigned want_ = pp.piece[(J_)].value;                                                              \
        const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime();                                          \
        while ((int)(__hip_atomic_load(pp.flags + fi_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want_) < 0) { \
          __builtin_amdgcn_s_sleep(16);                                                                           \
          /* fail fast: once ANY work-group of any launch has given up, nobody waits out the deadline again */    \
          if (pp.err && __hip_atomic_load(pp.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;       \
          if (pp.timeout_ticks && __builtin_amdgcn_s_memrealtime() - t0_ > pp.timeout_ticks) {                    \
            if (pp.err) atomicCAS(pp.err, 0u, 0x80000000u | (unsigned)(J_));                                      \
            break;                                                                                                \
          }                                                                                                       \
        }                                                                                                         \
      }                                                                                                           \
      P_BARRIER();                                                                                                \
    }                                                                                                             \
  }

  const int k_row_off = l31 * 256;
  const int k_sw = l31 & 15;
  const int g = lane >> 4, t16 = lane & 15;
  const int v_key_lo = 4 * hi + (t16 >> 2);
  const int v_byte_lo = (g & 1) * 32 + (t16 & 3) * 8;
  const int v_sw = (t16 >> 2) << 6;

  // one 64-key tile from ring stage `stg`: keys [key0, key0 + 64) of a piece with `skv` rows (attn7.hip's tile, variant 148 / 132)
  auto tile = [&](const int stg, const int key0, const int skv) __attribute__((always_inline)) {
    const char* ks = smem + (stg & (NST - 1)) * STAGE_BYTES;
    const char* vs = ks + TILE_BYTES;
    f32x16 st[2];
    const bool no_ref = UNIT && m_run < -1.0e29f;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int c = ds * 2 + hi;
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + kb * 8192 + k_row_off + ((c ^ k_sw) << 4));
        st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], ds == 0 ? (UNIT ? cinit : zero16) : st[kb], 0, 0, 0);
      }
    if (key0 + KVB > skv) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= skv) st[kb][r] = NEG_BIG;
        }
    }
    float mb = -m_run * p.sc;
    float psum = 0.f;
    __builtin_amdgcn_s_setprio(1);
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
      if (__any(!(ps <= p_lim) || no_ref)) {      // lazy max (attn2.hip): the partial row sum bounds every P of the block
        float mloc = st[kb][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, st[kb][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        if (UNIT) mloc += m_base;
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
        if (UNIT) {
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
    __builtin_amdgcn_s_setprio(0);
    l_run += psum;
  };

  // ---- walk the pieces ----
  const int np = __builtin_amdgcn_readfirstlane(pp.n_pieces);
  unsigned gt = 0;                 // ring position of the current piece's tile 0 (always even)
  unsigned iv = 0;                 // interval counter (verdict word parity)
  P_WAIT_PIECE(0);
  P_SET_DMA_PIECE(0);
  P_DMA_TILE(0, gt);
  P_DMA_TILE(1, gt + 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  P_BARRIER();
  for (int pj = 0; pj < np; ++pj) {
    const int skv = __builtin_amdgcn_readfirstlane(pp.piece[pj].rows);
    const int nt = (skv + KVB - 1) / KVB;
    const unsigned g_next = gt + (unsigned)((nt + 1) & ~1);
    const bool has_next = pj + 1 < np;
    const int nflag = has_next ? __builtin_amdgcn_readfirstlane(pp.piece[has_next ? pj + 1 : pj].flag) : -1;
    const unsigned nvalue = has_next ? pp.piece[has_next ? pj + 1 : pj].value : 0u;
    bool next_ready = has_next && nflag < 0;      // wave-uniform
    bool next_issued = false;
    if (pp.ptrace && tid == 0) pp.ptrace[(size_t)blockIdx.x * np + pj] = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < nt; t += 2) {
      const bool last_iv = t + 2 >= nt;
      if (!last_iv) {                                    // the other half of the ring: last read in the previous interval
        P_DMA_TILE(t + 2, gt + t + 2);
        P_DMA_TILE(t + 3, gt + t + 3);
      } else if (has_next && next_ready) {               // run through the boundary: the next piece's first interval
        P_SET_DMA_PIECE(pj + 1);
        P_DMA_TILE(0, g_next);
        P_DMA_TILE(1, g_next + 1);
        next_issued = true;
      }
      const bool poll = has_next && !next_ready;         // wave-uniform
      unsigned fv = 0;
      if (poll && wave == 0) fv = load_flag_async(pp.flags + nflag);
      tile((int)(gt + t), t * KVB, skv);
      if (t + 1 < nt) tile((int)(gt + t + 1), (t + 1) * KVB, skv);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this interval's DMA requests and the flag sample
      if (poll && tid == 0) verdict[iv & 1] = (int)(fv - nvalue) >= 0 ? 1u : 0u;
      P_BARRIER();
      if (poll) next_ready = __builtin_amdgcn_readfirstlane(verdict[iv & 1]) != 0;
      ++iv;
    }
    if (!has_next) break;
    if (!next_issued) {                                  // bubble: the next piece was not known to be there in time
      if (!next_ready) P_WAIT_PIECE(pj + 1);
      P_SET_DMA_PIECE(pj + 1);
      P_DMA_TILE(0, g_next);
      P_DMA_TILE(1, g_next + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      P_BARRIER();
    }
    gt = g_next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  attc::store_result(p, q0 + l31, head, hi, ot, m_run, l_run);
  if (p.trace && tid == 0 && (int)blockIdx.x < p.trace_cap) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* t = p.trace + (size_t)blockIdx.x * 4;
    t[0] = t_start; t[1] = __builtin_amdgcn_s_memrealtime(); t[2] = hwid; t[3] = xcc;
  }
#undef P_SET_DMA_PIECE
#undef P_DMA_TILE
#undef P_BARRIER
#undef P_WAIT_PIECE
}

template <bool UNIT>
int launch(const Params& pp, hipStream_t st) {
  static icv_dev_flags attr_set = {};
  if (int rc = icv_ensure_dynamic_lds((const void*)attn7p_kernel<UNIT>, LDS_BYTES, &attr_set, "attn7p")) return rc;
  const int64_t nwg = (int64_t)pp.a.heads * pp.a.nqb;
  hipLaunchKernelGGL((attn7p_kernel<UNIT>), dim3((unsigned)nwg), dim3(NW * 64), LDS_BYTES, st, pp);
  return icv_check_launch("icv_attention_fwd_pieces");
}

}  // namespace att7p

extern "C" int icv_attention_fwd_pieces(const void* q, int64_t ldq, const icv_kv_piece* pieces, int64_t n_pieces, int64_t ldk, int64_t ldv,
                                        void* o, int64_t ldo, int64_t Sq, int64_t heads, float scale, const uint32_t* flags, uint32_t* err,
                                        int64_t timeout_us, void* trace, void* stream) {
  ICV_REQUIRE(q && pieces && o, "icv_attention_fwd_pieces: null pointer");
  ICV_REQUIRE(Sq > 0 && heads > 0 && n_pieces > 0, "icv_attention_fwd_pieces: empty problem (Sq=%lld heads=%lld pieces=%lld)", (long long)Sq,
              (long long)heads, (long long)n_pieces);
  ICV_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "icv_attention_fwd_pieces: leading dims must keep 16-byte row alignment");
  att7p::Params pp;
  int64_t total = 0;
  int n = 0;
  for (int64_t i = 0; i < n_pieces; ++i) {
    const icv_kv_piece& s = pieces[i];
    ICV_REQUIRE(s.rows >= 0 && s.rows < (1LL << 31) / 64, "icv_attention_fwd_pieces: piece %lld has %lld rows", (long long)i, (long long)s.rows);
    if (s.rows == 0) continue;
    ICV_REQUIRE(n < att7p::MAX_PIECES, "icv_attention_fwd_pieces: more than %d non-empty pieces", att7p::MAX_PIECES);
    ICV_REQUIRE(s.k && s.v, "icv_attention_fwd_pieces: piece %lld has a null pointer", (long long)i);
    ICV_REQUIRE(s.flag < 0 || flags, "icv_attention_fwd_pieces: piece %lld waits for flag %d but no flag array was given", (long long)i, s.flag);
    // the LDS-DMA addresses keep 32-bit per-lane byte offsets from a tile base: a tile spans 64 rows
    ICV_REQUIRE(64 * ldk * 2 < (1LL << 32) && 64 * ldv * 2 < (1LL << 32), "icv_attention_fwd_pieces: row stride too large");
    att7p::Piece& d = pp.piece[n++];
    d.k = (const bf16_t*)s.k; d.v = (const bf16_t*)s.v; d.rows = (int)s.rows; d.flag = s.flag; d.value = s.value; d.pad = 0;
    total += s.rows;
  }
  ICV_REQUIRE(n > 0, "icv_attention_fwd_pieces: every piece is empty");
  attc::fill_params(pp.a, q, ldq, nullptr, ldk, nullptr, ldv, o, ldo, nullptr, 0, nullptr, 0, 0, Sq, total, heads, scale, 256);
  pp.a.trace = icv_attention_trace_buffer(&pp.a.trace_cap);
  pp.flags = flags;
  pp.err = err;
  pp.timeout_ticks = timeout_us > 0 ? (unsigned long long)timeout_us * 100ull : 0ull;
  pp.ptrace = (unsigned long long*)trace;
  pp.n_pieces = n;
  hipStream_t st = (hipStream_t)stream;
  return (pp.a.sc == 1.0f && icv_get_option_int("attn_unit_scale", 1)) ? att7p::launch<true>(pp, st) : att7p::launch<false>(pp, st);
}

// The plain launch and the carried-state chunk launch as ONE piece of this kernel: the same tiles in the same order through the same
// arithmetic as attn7.hip's default variant (bit-identical: tests/test_attn_pieces_gpu.py), measured 1.5-2.7 % faster at the 14B shapes
// (23.36 vs 24.01 ms at S = 37 440, interleaved, profiles/r06/attn7p_vs_attn7.txt) - so attention.hip routes its default there.
int icv_attn7p_single(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, float* acc,
                      int64_t ldacc, float* ml, int state_in, int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, hipStream_t st) {
  ICV_REQUIRE(Skv > 0 && Skv < (1LL << 31) / 64 && 64 * ldk * 2 < (1LL << 32) && 64 * ldv * 2 < (1LL << 32), "icv_attention: key axis / row stride too large");
  att7p::Params pp;
  attc::fill_params(pp.a, q, ldq, nullptr, ldk, nullptr, ldv, o, ldo, acc, ldacc, ml, state_in, state_out, Sq, Skv, heads, scale, 256);
  pp.a.trace = icv_attention_trace_buffer(&pp.a.trace_cap);
  pp.flags = nullptr; pp.err = nullptr; pp.timeout_ticks = 0; pp.ptrace = nullptr;
  pp.n_pieces = 1;
  att7p::Piece& d = pp.piece[0];
  d.k = (const bf16_t*)k; d.v = (const bf16_t*)v; d.rows = (int)Skv; d.flag = -1; d.value = 0; d.pad = 0;
  return (pp.a.sc == 1.0f && icv_get_option_int("attn_unit_scale", 1)) ? att7p::launch<true>(pp, st) : att7p::launch<false>(pp, st);
}

namespace {
__global__ void flag_write_kernel(uint32_t* flag, uint32_t value, unsigned long long delay_ticks) {
  if (delay_ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(32);
  }
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

extern "C" int icv_flag_write(uint32_t* flags, int64_t index, uint32_t value, int64_t delay_us, void* stream) {
  ICV_REQUIRE(flags && index >= 0, "icv_flag_write: bad argument");
  ICV_REQUIRE(delay_us >= 0 && delay_us <= 10000000, "icv_flag_write: delay_us out of range");
  hipLaunchKernelGGL(flag_write_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flags + index, value, (unsigned long long)delay_us * 100ull);
  return icv_check_launch("icv_flag_write");
}
