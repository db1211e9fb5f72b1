// 256x256x64 bf16 MFMA GEMM, 8 waves, 4 phases per K-tile, LDS-DMA with counted vmcnt, two wave
// groups staggered by one barrier (one group issues MFMAs while the other reads LDS / issues DMA).
// Same math/epilogues as gemm.hip; used when N % 256 == 0 and the problem is large.
//
// Geometry: block tile 256(M) x 256(N) x 64(K); waves = 2 (wr) x 4 (wc); wave tile 128 x 64 =
// a-halves {a0,a1} (64 rows each) x b-halves {b0,b1} (32 cols each); v_mfma_f32_16x16x32_bf16,
// operands swapped (W fragment = A operand) so a lane owns 4 consecutive n (vector epilogue).
// LDS: 2 stages x 4 units x 16 KiB = 128 KiB.  A "unit" is what ONE phase needs from ALL waves:
//   A0 = rows {wr*128 + [0,64)},  A1 = rows {wr*128 + 64 + [0,64)},
//   B0 = cols {wc*64 + [0,32)},   B1 = cols {wc*64 + 32 + [0,32)}      (128 rows x 128 B each)
// unit image: [128 rows][8 chunks of 16 B], physical chunk = logical ^ ((row>>1)&7) applied on the
// DMA *source* address and mirrored on ds_read_b128 (conflict-free, see gemm.hip).
//
// Two main-loop schedules (template SCH bit 0; A/B switch "gemm256_sched"):
//   * SCH&1 (DEFAULT): two phases of 32 MFMAs per K-tile — half the barriers; measured +3.4..5 % on every DiT shape
//     (14B QKV 1286 -> 1334, FFN1 1270 -> 1322 TF/s same box; see the loop for its RAW / WAR invariants);
//   * the original four phases of 16 MFMAs, described here:
// Per K-tile t (stage s = t&1), every phase = L-segment ; barrier ; 16 MFMA ; barrier :
//   ph1: read a0,b0 | DMA B1(t+1)->s^1 | vmcnt(8) |  a0 x b0
//   ph2: read b1    | DMA A1(t+1)->s^1 | vmcnt(8) |  a0 x b1
//   ph3: read a1    | DMA A0(t+2)->s   |          |  a1 x b1
//   ph4: --         | DMA B0(t+2)->s   | vmcnt(8) |  a1 x b0   (b0 kept in VGPRs)
// Invariants (wave group 1 runs one barrier behind group 0):
//   RAW: a unit is read one phase AFTER the phase whose vmcnt retired it (every wave's share has
//        landed and a barrier separates the wait from the read).  vmcnt(8) = 4 younger units in flight.
//   WAR: a slot is re-staged >= 2 phases after its last ds_read (A0: read ph1, DMA ph3).
//   Tail tiles re-load the last tile (clamped k) so the counted waits stay uniform.
#include "icv_common.h"

#ifndef G256_SCHED_DEFAULT
#define G256_SCHED_DEFAULT 3
#endif

namespace g256 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int UNIT_BYTES = 128 * 128;         // 16 KiB
constexpr int STAGE_BYTES = 4 * UNIT_BYTES;   // A0 A1 B0 B1
constexpr int LDS_BYTES = 2 * STAGE_BYTES;    // 128 KiB
constexpr int U_A0 = 0, U_A1 = 1, U_B0 = 2, U_B1 = 3;

struct Params {
  const bf16_t* A; int64_t lda;
  const bf16_t* W; int64_t ldw;
  const float* bias;
  int64_t M, N, K;
  void* out; int64_t ldo; int64_t nsplit; int64_t split_stride;
  const float* resid; int64_t ldr;
  const float* gate;
  int tiles_m, tiles_n;
  int gm;   // M-tiles per group of the grouped tile order
  int ablate;   // timing experiments ("gemm256_ablate"): 1 = no DMA inside the main loop, 2 = every DMA re-reads K-tile 0 (L2-resident source); results are then WRONG
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

#define G256_BARRIER()                      \
  do {                                      \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_s_barrier();           \
    asm volatile("" ::: "memory");          \
  } while (0)
#define G256_VMCNT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")

// AUX = cache policy bits of the DMA load (0 = default; 2 = nt: do not retain in L2 - round 3's A/B, SCH bits 16 / 32)
template <int AUX = 0>
__device__ __forceinline__ void dma_unit(const char* __restrict__ base, const unsigned (&off)[2],
                                         int64_t kbyte, char* lds_unit, int wave) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const char* src = base + (int64_t)off[q] + kbyte;
    char* dst = lds_unit + q * 8192 + wave * 1024;  // wave-uniform; HW adds lane*16
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, AUX);
  }
}

// MF = 16: v_mfma_f32_16x16x32_bf16 fragments (8x4 per wave);  MF = 32: v_mfma_f32_32x32x16_bf16 (4x2 per wave)
// SCH bit 0: two phases of 32 MFMAs per K-tile instead of four of 16 (half the barriers; see the loop);
//     bit 1: RESID epilogue issues the residual loads of 16 fragments before consuming any (the A/B fragment
//            registers are dead by then), instead of one load -> wait -> store round trip per fragment.
template <int EPI, int MF, int SCH>
__global__ __launch_bounds__(512) void gemm256_kernel(Params p) {
  constexpr bool TWO_PHASE = (SCH & 1) != 0, BATCH_EPI = (SCH & 2) != 0;
  constexpr bool SERP = (SCH & 8) != 0;       // MFMA order inside a phase: serpentine over the (a, b) fragment grid
  constexpr bool EARLY_B1 = (SCH & 4) != 0;   // two-phase only: B1 of the next tile is requested with A0/B0 (a full tile ahead), not half a tile
  static_assert(!EARLY_B1 || TWO_PHASE, "EARLY_B1 is a variant of the two-phase schedule");
  // round 3 A/B (profiles/r03/gemm_cache_policy_ab.txt): non-temporal DMA loads of the activation (bit 16) / weight (bit 32) stream
  constexpr int AUX_A = (SCH & 16) ? 2 : 0, AUX_B = (SCH & 32) ? 2 : 0;
  // round 4 A/B (SCH bit 6, "k-split"): the four 16 KiB units of a stage are cut by K-HALF instead of by row half - AK0 / AK1 =
  // all 256 A rows x k [0,32) / [32,64), BK0 / BK1 likewise - and a phase is ONE k-step of all 32 accumulators: both phases read
  // 12 fragments per wave (8 a + 4 b) instead of 16 + 8, so the two L-segments of a K-tile load the LDS port equally
  // (profiles/r04/gemm_ksplit_ab.txt).  Same DMA distances, same WAR / RAW argument as the two-phase schedule below.
  constexpr bool KSPLIT = (SCH & 64) != 0;
  static_assert(!KSPLIT || (MF == 16 && TWO_PHASE && !EARLY_B1), "k-split is a variant of the two-phase 16x16x32 schedule");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- block -> tile (bijective XCD remap + grouped order) ----
  const int nwg = p.tiles_m * p.tiles_n;
  int wg;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int GM = p.gm;
  const int group_size = GM * p.tiles_n;
  const int g = wg / group_size;
  const int first_m = g * GM;
  const int gm = min(p.tiles_m - first_m, GM);
  const int tm = first_m + (wg % group_size) % gm;
  const int tn = (wg % group_size) / gm;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  // ---- per-thread DMA source offsets (bytes from A / W base, k = 0), 2 passes per unit ----
  unsigned offA[2][2], offB[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int u = q * 64 + (tid >> 3);            // unit row 0..127
    const int pc = tid & 7;
    const int c = pc ^ ((u >> 1) & 7);            // logical 16-B chunk held by physical chunk pc
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int64_t ra = m0 + (u >> 6) * 128 + h * 64 + (u & 63);
      ra = ra < p.M ? ra : p.M - 1;
      offA[h][q] = (unsigned)((ra * p.lda + c * 8) * 2);
      int64_t rb = n0 + (u >> 5) * 64 + h * 32 + (u & 31);
      rb = rb < p.N ? rb : p.N - 1;
      offB[h][q] = (unsigned)((rb * p.ldw + c * 8) * 2);
    }
  }
  // k-split units: unit row u = block row (A: m0 + u, W: n0 + u), 64-byte rows (32 k), 4 chunks of 16 B, physical chunk =
  // logical ^ ((u >> 1) & 3) (conflict-free by PMC; selectable below)
  unsigned offKA[2], offKB[2];
  // chunk swizzle of a 64-byte-row unit: selectable for the bank-conflict A/B ("gemm256_ablate" bits 2-3; results stay correct)
  const int ksw = (p.ablate >> 2) & 3;
  auto kswz = [ksw](int r) -> int {
    // measured (SQ_LDS_BANK_CONFLICT, profiles/r04/gemm_ksplit_ab.txt): (r>>1)&3 and the bit-1 | bit-3 form are conflict-free,
    // (r>>2)&3 and r&3 conflict two-way on every ds_read_b128
    return ksw == 0 ? ((r >> 1) & 3) : ksw == 1 ? ((r >> 2) & 3) : ksw == 2 ? (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) : (r & 3);
  };
  if (KSPLIT) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = q * 128 + (tid >> 2);
      const int c = (tid & 3) ^ kswz(u);
      int64_t ra = m0 + u;
      ra = ra < p.M ? ra : p.M - 1;
      offKA[q] = (unsigned)((ra * p.lda + c * 8) * 2);
      int64_t rb = n0 + u;
      rb = rb < p.N ? rb : p.N - 1;
      offKB[q] = (unsigned)((rb * p.ldw + c * 8) * 2);
    }
  }
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Wb = reinterpret_cast<const char*>(p.W);
  const int nt = (int)(p.K / BK);
  const bool k0_only = p.ablate & 2;
  auto kbyte = [&](int t) -> int64_t { return k0_only ? 0 : (int64_t)(t < nt ? t : nt - 1) * (BK * 2); };

  f32x4 acc[8][4];     // MF == 16
  f32x16 acc32[4][2];  // MF == 32
  if (MF == 16) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
  }

  // ---- fragment read addressing: lane (fr = row in 16-row frag, kq = 8-wide k chunk) ----
  const int fr = (MF == 16) ? (lane & 15) : (lane & 31), kq = (MF == 16) ? (lane >> 4) : (lane >> 5);
  // a-frag i (0..3) of a-half: unit row = wr*64 + i*16 + fr ; b-frag j (0..1): unit row = wc*32 + j*16 + fr
  // MF 16: k-step = 32 (4 chunks, kq = 0..3), 2 k-steps;  MF 32: k-step = 16 (2 chunks, kq = 0..1), 4 k-steps
  constexpr int NKS = (MF == 16) ? 2 : 4, CPK = (MF == 16) ? 4 : 2;
  int a_off[NKS], b_off[NKS];  // byte offset within a unit per k-step, minus the i/j row term
  const int ar = wr * 64 + fr, br = wc * 32 + fr;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    // (row + 16*i) keeps ((row>>1)&7): (16*i)>>1 = 8*i does not touch bits 0..2 -> same swizzle
    a_off[ks] = ar * 128 + (((ks * CPK + kq) ^ ((ar >> 1) & 7)) << 4);
    b_off[ks] = br * 128 + (((ks * CPK + kq) ^ ((br >> 1) & 7)) << 4);
  }
  constexpr int NAF = (MF == 16) ? 4 : 2, NBF = (MF == 16) ? 2 : 1, FROWS = MF * 128;  // frags per half, bytes per frag row block

  if (KSPLIT) {
    constexpr int U_AK0 = 0, U_AK1 = 1, U_BK0 = 2, U_BK1 = 3;
    // fragment addressing: a-fragment i8 (0..7) = block rows wr*128 + (i8>>2)*64 + (i8&3)*16 + fr, b-fragment j4 (0..3) = block cols
    // wc*64 + (j4>>1)*32 + (j4&1)*16 + fr (the epilogue's mapping); lane (fr, kq) reads the 16-byte chunk kq of its row
    const int kar = wr * 128 + fr, kbr = wc * 64 + fr;
    // (fragment rows add multiples of 16 to kar / kbr: none of the swizzle candidates below looks at bits >= 4)
    const int ka_off = kar * 64 + ((kq ^ kswz(kar & 15)) << 4);
    const int kb_off = kbr * 64 + ((kq ^ kswz(kbr & 15)) << 4);
    bf16x8 kaf[8], kbf[4];
    dma_unit<AUX_A>(Ab, offKA, kbyte(0), smem + U_AK0 * UNIT_BYTES, wave);
    dma_unit<AUX_B>(Wb, offKB, kbyte(0), smem + U_BK0 * UNIT_BYTES, wave);
    dma_unit<AUX_A>(Ab, offKA, kbyte(0) + 64, smem + U_AK1 * UNIT_BYTES, wave);
    dma_unit<AUX_B>(Wb, offKB, kbyte(0) + 64, smem + U_BK1 * UNIT_BYTES, wave);
    dma_unit<AUX_A>(Ab, offKA, kbyte(1), smem + STAGE_BYTES + U_AK0 * UNIT_BYTES, wave);
    dma_unit<AUX_B>(Wb, offKB, kbyte(1), smem + STAGE_BYTES + U_BK0 * UNIT_BYTES, wave);
    G256_VMCNT8();            // AK0(0), BK0(0) landed (4 younger units in flight)
    G256_BARRIER();
    if (wr == 1) G256_BARRIER();  // stagger: group 1 runs one barrier behind group 0
#define G256K_READ(UA, UB)                                                                                         \
    {                                                                                                              \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                \
        kbf[j] = *reinterpret_cast<const bf16x8*>(cur + (UB) * UNIT_BYTES + kb_off + ((j >> 1) * 32 + (j & 1) * 16) * 64); \
      _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                \
        kaf[i] = *reinterpret_cast<const bf16x8*>(cur + (UA) * UNIT_BYTES + ka_off + ((i >> 2) * 64 + (i & 3) * 16) * 64); \
    }
#define G256K_MFMA()                                                                                               \
    {                                                                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      __builtin_amdgcn_s_setprio(1);                                                                               \
      _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                \
      _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                           \
        const int j = (i & 1) ? 3 - jj : jj;          /* boustrophedon: consecutive MFMAs share an operand */      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kbf[j], kaf[i], acc[i][j], 0, 0, 0);                   \
      }                                                                                                            \
      __builtin_amdgcn_s_setprio(0);                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
    // Per K-tile t (stage s = t&1), each phase = L-segment ; barrier ; 32 MFMA ; barrier :
    //   P1: read AK0, BK0 | DMA AK1(t+1), BK1(t+1) -> s^1 | vmcnt(8) : AK1, BK1(t) landed (issued in P1(t-1); younger: P2(t-1)'s and these)
    //   P2: read AK1, BK1 | DMA AK0(t+2), BK0(t+2) -> s   | vmcnt(8) : AK0, BK0(t+1) landed (issued in P2(t-1))
    // WAR: a unit is re-staged in the phase after its last read, by which time every wave's reads of it were retired
    // (lgkmcnt(0)) before a barrier the re-staging group has passed - the argument of the two-phase schedule, for all four units.
    for (int t = 0; t < nt; ++t) {
      char* cur = smem + (t & 1) * STAGE_BYTES;
      char* oth = smem + ((t + 1) & 1) * STAGE_BYTES;
      G256K_READ(U_AK0, U_BK0);
      if (!(p.ablate & 1)) dma_unit<AUX_A>(Ab, offKA, kbyte(t + 1) + 64, oth + U_AK1 * UNIT_BYTES, wave);
      if (!(p.ablate & 1)) dma_unit<AUX_B>(Wb, offKB, kbyte(t + 1) + 64, oth + U_BK1 * UNIT_BYTES, wave);
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      G256_BARRIER();
      G256K_MFMA();
      G256_BARRIER();
      G256K_READ(U_AK1, U_BK1);
      if (!(p.ablate & 1)) dma_unit<AUX_A>(Ab, offKA, kbyte(t + 2), cur + U_AK0 * UNIT_BYTES, wave);
      if (!(p.ablate & 1)) dma_unit<AUX_B>(Wb, offKB, kbyte(t + 2), cur + U_BK0 * UNIT_BYTES, wave);
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      G256_BARRIER();
      G256K_MFMA();
      G256_BARRIER();
    }
#undef G256K_READ
#undef G256K_MFMA
  } else {
  // ---- prologue: tile 0 complete + A0,B0 of tile 1 ----
  dma_unit<AUX_A>(Ab, offA[0], kbyte(0), smem + U_A0 * UNIT_BYTES, wave);
  dma_unit<AUX_B>(Wb, offB[0], kbyte(0), smem + U_B0 * UNIT_BYTES, wave);
  dma_unit<AUX_B>(Wb, offB[1], kbyte(0), smem + U_B1 * UNIT_BYTES, wave);
  dma_unit<AUX_A>(Ab, offA[1], kbyte(0), smem + U_A1 * UNIT_BYTES, wave);
  dma_unit<AUX_A>(Ab, offA[0], kbyte(1), smem + STAGE_BYTES + U_A0 * UNIT_BYTES, wave);
  dma_unit<AUX_B>(Wb, offB[0], kbyte(1), smem + STAGE_BYTES + U_B0 * UNIT_BYTES, wave);
  if (EARLY_B1) dma_unit<AUX_B>(Wb, offB[1], kbyte(1), smem + STAGE_BYTES + U_B1 * UNIT_BYTES, wave);
  if (EARLY_B1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // A0(0), B0(0), B1(0) landed (4 younger units in flight)
  else if (TWO_PHASE) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // A0(0), B0(0), B1(0) landed (3 younger units in flight)
  else G256_VMCNT8();     // A0(0), B0(0) landed (4 younger units in flight)
  G256_BARRIER();
  if (wr == 1) G256_BARRIER();  // stagger: group 1 runs one barrier behind group 0

  bf16x8 af[NAF][NKS], b0f[NBF][NKS], b1f[NBF][NKS];

#define G256_MFMA(AH, BF, BH)                                                                   \
  {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    __builtin_amdgcn_s_setprio(1);                                                              \
    _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks)                                          \
    _Pragma("unroll") for (int i = 0; i < NAF; ++i)                                             \
    _Pragma("unroll") for (int jj = 0; jj < NBF; ++jj) {                                        \
      /* SERP: boustrophedon over (i, j) so that every consecutive MFMA pair shares one operand */ \
      const int j = (SERP && ((i + ks * NAF) & 1)) ? NBF - 1 - jj : jj;                         \
      if (MF == 16)                                                                             \
        acc[(AH) * 4 + i][(BH) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(              \
            BF[j][ks], af[i][ks], acc[(AH) * 4 + i][(BH) * 2 + j], 0, 0, 0);                    \
      else                                                                                      \
        acc32[(AH) * 2 + i][(BH)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                    \
            BF[j][ks], af[i][ks], acc32[(AH) * 2 + i][(BH)], 0, 0, 0);                          \
    }                                                                                           \
    __builtin_amdgcn_s_setprio(0);                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  }

  if (TWO_PHASE) {
    // Two phases per K-tile t (stage s = t&1), each = L-segment ; barrier ; 32 MFMA ; barrier :
    //   P1: read a0,b0,b1 | DMA B1(t+1),A1(t+1) -> s^1 | vmcnt(8) |  a0 x b0, a0 x b1
    //   P2: read a1       | DMA A0(t+2),B0(t+2) -> s   | vmcnt(6) |  a1 x b0, a1 x b1
    // RAW: A1(t) is retired by P1's vmcnt(8) (4 younger units stay in flight) and read in P2; B1(t+1) (and the older
    //      A0/B0(t+1)) by P2's vmcnt(6) and read in P1(t+1): always one phase after the wait + barrier.
    // WAR: A0/B0 of a stage are re-staged in the phase after their last read (P1 -> P2) — every L-segment therefore
    //      retires its own ds_reads (lgkmcnt(0)) before its barrier, so no read is still pending when the other wave
    //      group issues the DMA two barriers later; A1/B1 are re-staged three / four barriers after their last read.
    for (int t = 0; t < nt; ++t) {
      char* cur = smem + (t & 1) * STAGE_BYTES;
      char* oth = smem + ((t + 1) & 1) * STAGE_BYTES;
      // ---------------- phase 1: a0 x (b0, b1) ----------------
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int j = 0; j < NBF; ++j) {
          b0f[j][ks] = *reinterpret_cast<const bf16x8*>(cur + U_B0 * UNIT_BYTES + b_off[ks] + j * FROWS);
          b1f[j][ks] = *reinterpret_cast<const bf16x8*>(cur + U_B1 * UNIT_BYTES + b_off[ks] + j * FROWS);
        }
#pragma unroll
        for (int i = 0; i < NAF; ++i)
          af[i][ks] = *reinterpret_cast<const bf16x8*>(cur + U_A0 * UNIT_BYTES + a_off[ks] + i * FROWS);
      }
      if (!EARLY_B1 && !(p.ablate & 1)) dma_unit<AUX_B>(Wb, offB[1], kbyte(t + 1), oth + U_B1 * UNIT_BYTES, wave);
      if (!(p.ablate & 1)) dma_unit<AUX_A>(Ab, offA[1], kbyte(t + 1), oth + U_A1 * UNIT_BYTES, wave);
      // A1(t) must have landed.  Younger: A0,B0(t+1) + this phase's B1,A1(t+1) = 8 instructions; EARLY_B1: A0,B0,B1(t+1) + A1(t+1) = 8
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      G256_BARRIER();
      G256_MFMA(0, b0f, 0);
      G256_MFMA(0, b1f, 1);
      G256_BARRIER();
      // ---------------- phase 2: a1 x (b0, b1) ----------------
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int i = 0; i < NAF; ++i)
          af[i][ks] = *reinterpret_cast<const bf16x8*>(cur + U_A1 * UNIT_BYTES + a_off[ks] + i * FROWS);
      if (!(p.ablate & 1)) dma_unit<AUX_A>(Ab, offA[0], kbyte(t + 2), cur + U_A0 * UNIT_BYTES, wave);
      if (!(p.ablate & 1)) dma_unit<AUX_B>(Wb, offB[0], kbyte(t + 2), cur + U_B0 * UNIT_BYTES, wave);
      if (EARLY_B1) {
        // B1 of this stage was last read in P1(t): re-stage it now, a full tile before P1(t+2) reads it.  A0,B0,B1(t+1)
        // must have landed; younger: A1(t+1) + A0,B0,B1(t+2) = 8 instructions
        if (!(p.ablate & 1)) dma_unit<AUX_B>(Wb, offB[1], kbyte(t + 2), cur + U_B1 * UNIT_BYTES, wave);
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      } else
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      G256_BARRIER();
      G256_MFMA(1, b0f, 0);
      G256_MFMA(1, b1f, 1);
      G256_BARRIER();
    }
  } else
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * STAGE_BYTES;
    char* oth = smem + ((t + 1) & 1) * STAGE_BYTES;
    // ---------------- phase 1: a0 x b0 ----------------
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int j = 0; j < NBF; ++j)
        b0f[j][ks] = *reinterpret_cast<const bf16x8*>(cur + U_B0 * UNIT_BYTES + b_off[ks] + j * FROWS);
#pragma unroll
      for (int i = 0; i < NAF; ++i)
        af[i][ks] = *reinterpret_cast<const bf16x8*>(cur + U_A0 * UNIT_BYTES + a_off[ks] + i * FROWS);
    }
    dma_unit<AUX_B>(Wb, offB[1], kbyte(t + 1), oth + U_B1 * UNIT_BYTES, wave);
    G256_VMCNT8();
    G256_BARRIER();
    G256_MFMA(0, b0f, 0);
    G256_BARRIER();
    // ---------------- phase 2: a0 x b1 ----------------
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int j = 0; j < NBF; ++j)
        b1f[j][ks] = *reinterpret_cast<const bf16x8*>(cur + U_B1 * UNIT_BYTES + b_off[ks] + j * FROWS);
    dma_unit<AUX_A>(Ab, offA[1], kbyte(t + 1), oth + U_A1 * UNIT_BYTES, wave);
    G256_VMCNT8();
    G256_BARRIER();
    G256_MFMA(0, b1f, 1);
    G256_BARRIER();
    // ---------------- phase 3: a1 x b1 ----------------
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int i = 0; i < NAF; ++i)
        af[i][ks] = *reinterpret_cast<const bf16x8*>(cur + U_A1 * UNIT_BYTES + a_off[ks] + i * FROWS);
    dma_unit<AUX_A>(Ab, offA[0], kbyte(t + 2), cur + U_A0 * UNIT_BYTES, wave);
    G256_BARRIER();
    G256_MFMA(1, b1f, 1);
    G256_BARRIER();
    // ---------------- phase 4: a1 x b0 ----------------
    dma_unit<AUX_B>(Wb, offB[0], kbyte(t + 2), cur + U_B0 * UNIT_BYTES, wave);
    G256_VMCNT8();
    G256_BARRIER();
    G256_MFMA(1, b0f, 0);
    G256_BARRIER();
  }
#undef G256_MFMA
  }   // !KSPLIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain tail DMA before the LDS is released
  if (wr == 0) G256_BARRIER();                       // re-balance the stagger

  // ---- epilogue: a lane owns ONE row m and runs of 4 consecutive n (swapped MFMA operands) ----
#define G256_EMIT(M_, N_, V0_, V1_, V2_, V3_)                                                          \
  {                                                                                                    \
    const int64_t m = (M_), n = (N_);                                                                  \
    if (m < p.M && n < p.N) {                                                                          \
      float v0 = (V0_), v1 = (V1_), v2 = (V2_), v3 = (V3_);                                            \
      if (p.bias) {                                                                                    \
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);                                 \
        v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;                                                    \
      }                                                                                                \
      const int64_t off = icv_out_offset(m, n, p.ldo, p.N, p.nsplit, p.split_stride);                  \
      if (EPI == ICV_EPI_BF16 || EPI == ICV_EPI_GELU_BF16) {                                           \
        if (EPI == ICV_EPI_GELU_BF16) {                                                                \
          v0 = gelu_tanh(v0); v1 = gelu_tanh(v1); v2 = gelu_tanh(v2); v3 = gelu_tanh(v3);              \
        }                                                                                              \
        *reinterpret_cast<uint2*>((bf16_t*)p.out + off) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)); \
      } else if (EPI == ICV_EPI_RESID_F32) {                                                           \
        const float4 r = *reinterpret_cast<const float4*>(p.resid + m * p.ldr + n);                    \
        float4 o;                                                                                      \
        if (p.gate) {                                                                                  \
          const float4 gt = *reinterpret_cast<const float4*>(p.gate + n);                              \
          o = make_float4(r.x + gt.x * v0, r.y + gt.y * v1, r.z + gt.z * v2, r.w + gt.w * v3);         \
        } else {                                                                                       \
          o = make_float4(r.x + v0, r.y + v1, r.z + v2, r.w + v3);                                     \
        }                                                                                              \
        *reinterpret_cast<float4*>((float*)p.out + off) = o;                                           \
      } else {                                                                                         \
        *reinterpret_cast<float4*>((float*)p.out + off) = make_float4(v0, v1, v2, v3);                 \
      }                                                                                                \
    }                                                                                                  \
  }
  if (MF == 16 && EPI == ICV_EPI_RESID_F32 && BATCH_EPI) {
    // x[m, n] = resid[m, n] + gate[n] * (acc + bias[n]) with the 16 residual loads of a half (4 x 4 fragments) in
    // flight together: 64 VGPRs, the size of the dead A/B fragments.  N % 256 == 0 here, so every n is in range;
    // rows past M (partial last m-tile) load a clamped row and skip the store.
    float4 bs[4], gt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + kq * 4;
      bs[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      gt[j] = p.gate ? *reinterpret_cast<const float4*>(p.gate + n) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float4 rs[4][4];
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int64_t m = m0 + wr * 128 + hf * 64 + ii * 16 + fr;
        const int64_t mc = m < p.M ? m : p.M - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          rs[ii][j] = *reinterpret_cast<const float4*>(p.resid + mc * p.ldr + n0 + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + kq * 4);
      }
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int64_t m = m0 + wr * 128 + hf * 64 + ii * 16 + fr;
        if (m < p.M) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + kq * 4;
            const f32x4 a = acc[hf * 4 + ii][j];
            const float4 r = rs[ii][j];
            *reinterpret_cast<float4*>((float*)p.out + m * p.ldo + n) =
                make_float4(r.x + gt[j].x * (a[0] + bs[j].x), r.y + gt[j].y * (a[1] + bs[j].y),
                            r.z + gt[j].z * (a[2] + bs[j].z), r.w + gt[j].w * (a[3] + bs[j].w));
          }
        }
      }
    }
  } else if (MF == 16) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        G256_EMIT(m0 + wr * 128 + (i >> 2) * 64 + (i & 3) * 16 + fr, n0 + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + kq * 4,
                  acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3])
  } else {
    // 32x32 C layout: col = lane&31 (-> m), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (-> n)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          G256_EMIT(m0 + wr * 128 + (i >> 1) * 64 + (i & 1) * 32 + fr, n0 + wc * 64 + j * 32 + rr * 8 + kq * 4,
                    acc32[i][j][rr * 4 + 0], acc32[i][j][rr * 4 + 1], acc32[i][j][rr * 4 + 2], acc32[i][j][rr * 4 + 3])
  }
#undef G256_EMIT
}

template <int EPI, int MF, int SCH>
int launch(const Params& p, hipStream_t st) {
  static icv_dev_flags attr_set = {};
  if (int rc = icv_ensure_dynamic_lds((const void*)gemm256_kernel<EPI, MF, SCH>, LDS_BYTES, &attr_set, "gemm256")) return rc;
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  hipLaunchKernelGGL((gemm256_kernel<EPI, MF, SCH>), dim3((unsigned)nwg), dim3(512), LDS_BYTES, st, p);
  return icv_check_launch("icv_gemm_bf16(256)");
}

}  // namespace g256

// Called by icv_gemm_bf16 (gemm.hip) when the shape qualifies.
int icv_gemm256_dispatch(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                         int64_t M, int64_t N, int64_t K, int epilogue, void* out, int64_t ldo,
                         int64_t nsplit, int64_t split_stride, const float* resid, int64_t ldr,
                         const float* gate, hipStream_t st) {
  g256::Params p;
  p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.out = out; p.ldo = ldo; p.nsplit = nsplit; p.split_stride = split_stride;
  p.resid = resid; p.ldr = ldr; p.gate = gate;
  p.tiles_m = (int)((M + g256::BM - 1) / g256::BM);
  p.tiles_n = (int)((N + g256::BN - 1) / g256::BN);
  p.gm = icv_get_option_int("gemm256_gm", 4);
  p.ablate = icv_get_option_int("gemm256_ablate", 0);
  const bool m32 = icv_get_option_int("gemm256_mfma", 16) == 32;
  // schedule variant (A/B switch "gemm256_sched"): bit 0 = two 32-MFMA phases per K-tile, bit 1 = batched residual loads,
  // bit 2 (with both: 7) = B1 of the next tile requested a full tile ahead
  const int sch = icv_get_option_int("gemm256_sched", G256_SCHED_DEFAULT) & 127;
#define G256_CASE(E_)                                                                              \
  case E_:                                                                                         \
    if (m32) return (sch & 1) ? g256::launch<E_, 32, 1>(p, st) : g256::launch<E_, 32, 0>(p, st);   \
    switch (sch) {                                                                                 \
      case 0: return g256::launch<E_, 16, 0>(p, st);                                               \
      case 1: return g256::launch<E_, 16, 1>(p, st);                                               \
      case 2: return g256::launch<E_, 16, 2>(p, st);                                               \
      case 7: return g256::launch<E_, 16, 7>(p, st);                                               \
      case 11: return g256::launch<E_, 16, 11>(p, st);                                             \
      case 19: return g256::launch<E_, 16, 19>(p, st);                                             \
      case 35: return g256::launch<E_, 16, 35>(p, st);                                             \
      case 51: return g256::launch<E_, 16, 51>(p, st);                                             \
      case 67: return g256::launch<E_, 16, 67>(p, st);                                             \
      default: return g256::launch<E_, 16, 3>(p, st);                                              \
    }
  switch (epilogue) {
    G256_CASE(ICV_EPI_BF16)
    G256_CASE(ICV_EPI_GELU_BF16)
    G256_CASE(ICV_EPI_RESID_F32)
    G256_CASE(ICV_EPI_F32)
  }
#undef G256_CASE
  icv_set_error("icv_gemm_bf16: unknown epilogue %d", epilogue);
  return 1;
}
