// §8f-4: the Wan-VAE's convolutions as SHIFTED-ROW GEMMs on the matrix cores (no MIOpen: no kernel search on first use, the
// same kernel in every process, nothing transposed or padded on the fly).
//
// Layout idea (MI355X-first: 288 GB of HBM pay for halos).  An activation lives as a PADDED NDHWC volume [Tp, Hp, Wp, C] —
// the causal time padding and the spatial padding are REAL zero rows of the buffer — flattened to a row matrix X[rows, C].
// A convolution tap (dt, dh, dw) then reads, for output row m, input row m + off with ONE constant
// off = (dt*Hp + dh)*Wp + dw for the whole volume, so
//        out[m, :] = bias + sum_taps  X[m + off_tap, :] . W_tap^T        for m in [m0, m1)
// is a GEMM whose K axis runs over (tap, channel) and whose A-operand row base shifts by a constant per tap: no im2col, no
// per-element index arithmetic, no bounds checks in the loop.  Outputs are produced for EVERY row of the range, halo
// positions included (a few % of extra rows: (Hp*Wp)/(H*W) = 1.013 at 240 x 416); what lands on halo positions is garbage by
// construction and is zeroed by whoever produces the next convolution's input (icv_rmsnorm_act_rows_masked, the resamplers).
// 3x3x3 causal, (3,1,1), 1x3x3 per-frame and 1x1x1 convolutions of any stride-1 geometry are all the same kernel with a
// different tap table; the stride-2 convolutions of the encoder are computed at stride 1 and subsampled by their consumer.
//
// Kernel: gemm256.hip's pipeline re-cut for narrow N.  Block tile 256 (rows) x 32*NB (output channels), NB = 1 | 3 | 6 ->
// 32 / 96 / 192 channels per block (the VAE's widths are 96, 192, 384 = 1, 1, 2 tiles of 96 / 192; 768 = 4 x 192), 8 waves as
// 4 (rows) x 2 (cols), wave tile 64 x 16*NB = 4 x NB fragments of v_mfma_f32_16x16x32_bf16 with swapped operands (a lane
// owns 4 consecutive channels of one row: 8-byte epilogue stores).  K-tile = 64 = two HALVES of 32 channels; each half has
// its own (tap, channel offset), so channel counts that are multiples of 32 but not of 64 (96) need no padding: a K-tile
// may straddle two taps, the lanes that stage its upper 64 bytes simply use the other tap's row shift.
// LDS: 2 stages x (A0 16 KiB + A1 16 KiB + B NBP*8 KiB), LDS-DMA (global_load_lds 16 B, bank swizzle on the source
// address), two phases per K-tile with counted vmcnt and the two wave groups one barrier apart — schedule, RAW / WAR
// argument and swizzle are gemm256.hip's, with ONE B unit read in phase 1 and kept in registers:
//   P1(t): read a0, b   | DMA A1(t+1) -> s^1          | vmcnt(4+NBP): A1(t) landed        | a0 x b
//   P2(t): read a1      | DMA A0(t+2), B(t+2) -> s    | vmcnt(4+NBP): A0, B(t+1) landed   | a1 x b
#include "icv_common.h"

namespace cv {

constexpr int BM = 256, BK = 64;
constexpr int AUNIT = 128 * 128;      // 16 KiB: 128 rows x 64 k
constexpr int MAX_TAPS = 27;

struct Params {
  const char* X; int64_t ldx;          // bf16 rows, ldx in elements
  const char* W; int64_t ldw;          // bf16 [N, K] K-contiguous, K = halves * 32 (zero-padded to a multiple of 64)
  const float* bias;
  int64_t m0, m1;                      // output row range
  int64_t N;
  bf16_t* out; int64_t ldo;
  const bf16_t* resid; int64_t ldr;    // optional bf16 [rows, N]: added to the result (the residual branch of a block)
  int nt;                              // K-tiles
  int halves_per_tap;                  // cin / 32
  int ntaps;
  unsigned long long magic;            // ceil(2^32 / halves_per_tap) (= 2^32 for one half per tap): tap = (g * magic) >> 32 for g < 2^16
  int64_t tap_bytes[MAX_TAPS];         // row shift of tap i in BYTES (off_i * ldx * 2)
  int tiles_m, tiles_n;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

#define CV_BARRIER()                        \
  do {                                      \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_s_barrier();           \
    asm volatile("" ::: "memory");          \
  } while (0)

template <int NB>
__global__ __launch_bounds__(512) void conv_shift_kernel(Params p) {
  constexpr int BN = 32 * NB;
  constexpr int NBP = (BN + 63) / 64;            // DMA passes (64 rows each) of the B unit
  constexpr int BUNIT = NBP * 8192;
  constexpr int STAGE = 2 * AUNIT + BUNIT;
  constexpr int WAITN = 4 + NBP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // block -> tile: contiguous chunk of tiles per XCD (neighbouring row tiles share their halo rows in that XCD's L2)
  const int nwg = p.tiles_m * p.tiles_n;
  int wg;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
  const int64_t row0 = p.m0 + (int64_t)tm * BM;
  const int64_t n0 = (int64_t)tn * BN;

  // per-thread DMA source offsets (bytes, k = 0 of the thread's half) for the 2 passes of A0 / A1 and the NBP passes of B
  unsigned offA[2][2], offB[NBP];
  const int pc = tid & 7;
  bool upper;          // this lane stages a chunk of the K-tile's upper half (k 32..63)
  {
    // the logical chunk depends on the unit row through the swizzle; (row >> 1) & 7 is the same for pass 0 and pass 1
    // (rows differ by 64), so `upper` is one flag per lane
    const int u0 = tid >> 3;
    upper = ((pc ^ ((u0 >> 1) & 7)) >> 2) != 0;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int u = q * 64 + (tid >> 3);            // unit row 0..127
    const int c = pc ^ ((u >> 1) & 7);            // logical 16-B chunk held by physical chunk pc
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int64_t r = row0 + (u >> 5) * 64 + h * 32 + (u & 31);
      r = r < p.m1 ? r : p.m1 - 1;
      offA[h][q] = (unsigned)(r * p.ldx * 2 + (c & 3) * 16);
    }
  }
#pragma unroll
  for (int q = 0; q < NBP; ++q) {
    const int u = q * 64 + (tid >> 3);
    const int c = pc ^ ((u >> 1) & 7);
    int64_t r = n0 + u;
    r = r < p.N ? r : p.N - 1;
    offB[q] = (unsigned)(r * p.ldw * 2 + c * 16);
  }
  // byte shift of a K-tile's half: tap row shift + channel offset (halves past the last tap re-read the last real half:
  // their weights are zero, and the data they multiply is finite)
  const int last_half = p.ntaps * p.halves_per_tap - 1;
  auto half_bytes = [&](int g) -> int64_t {
    g = g < last_half ? g : last_half;
    const int tap = (int)(((unsigned long long)(unsigned)g * p.magic) >> 32);          // g < 2^16, magic <= 2^32: no overflow
    const int c0 = (g - tap * p.halves_per_tap) * 32;
    return p.tap_bytes[tap] + (int64_t)c0 * 2;
  };
  const int nt = p.nt;
  // the A shift of K-tile t as THIS lane needs it (its chunk lies in the lower or the upper half of the tile).  The tap table is
  // read with scalar loads, whose wait (lgkmcnt) would also drain the LDS reads of the segment that issues them: the value for
  // tile t+3 is therefore formed right behind the segment's own `s_waitcnt lgkmcnt(0)`, a whole tile before its DMA needs it.
  auto a_shift = [&](int t) -> int64_t {
    t = t < nt ? t : nt - 1;
    const int64_t lo = half_bytes(2 * t), hi = half_bytes(2 * t + 1);
    return upper ? hi : lo;
  };
  auto dma_a = [&](int h, int64_t kb, char* unit) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const char* src = p.X + (int64_t)offA[h][q] + kb;
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(unit + q * 8192 + wave * 1024), 16, 0, 0);
    }
  };
  auto dma_b = [&](int t, char* unit) {
    t = t < nt ? t : nt - 1;
#pragma unroll
    for (int q = 0; q < NBP; ++q) {
      const char* src = p.W + (int64_t)offB[q] + (int64_t)t * (BK * 2);
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(unit + q * 8192 + wave * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[4][NB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read addressing: lane (fr = row in the 16-row fragment, kq = 16-byte chunk of the 32-wide k-step)
  const int fr = lane & 15, kq = lane >> 4;
  const int ar = wr * 32 + fr, br = wc * 16 * NB + fr;      // unit rows of fragment 0
  int a_off[2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_off[ks] = ar * 128 + (((ks * 4 + kq) ^ ((ar >> 1) & 7)) << 4);     // + 16 rows per fragment keeps (row >> 1) & 7
    b_off[ks] = br * 128 + (((ks * 4 + kq) ^ ((br >> 1) & 7)) << 4);
  }

  char* const st0 = smem;
  char* const st1 = smem + STAGE;
  // prologue: A0(0), B(0), A1(0) -> stage 0; A0(1), B(1) -> stage 1
  int64_t kb1 = a_shift(1), kb2 = a_shift(2);       // shifts of tile t+1 / t+2 while tile t is computed
  {
    const int64_t kb0 = a_shift(0);
    dma_a(0, kb0, st0);
    dma_b(0, st0 + 2 * AUNIT);
    dma_a(1, kb0, st0 + AUNIT);
    dma_a(0, kb1, st1);
    dma_b(1, st1 + 2 * AUNIT);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");      // A0(0), B(0) landed
  CV_BARRIER();
  if (wr >= 2) CV_BARRIER();                                          // stagger: waves 4..7 run one barrier behind

  bf16x8 af[2][2], bf[NB][2];
#define CV_MFMA(AH)                                                                                         \
  {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
    _Pragma("unroll") for (int jj = 0; jj < NB; ++jj) {                                                     \
      const int j = (i & 1) ? NB - 1 - jj : jj;                                                             \
      acc[(AH) * 2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][ks], af[i][ks], acc[(AH) * 2 + i][j], 0, 0, 0); \
    }                                                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  }
  for (int t = 0; t < nt; ++t) {
    char* cur = (t & 1) ? st1 : st0;
    char* oth = (t & 1) ? st0 : st1;
    // ---------------- phase 1: a0 x b ----------------
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < NB; ++j) bf[j][ks] = *reinterpret_cast<const bf16x8*>(cur + 2 * AUNIT + b_off[ks] + j * 2048);
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i][ks] = *reinterpret_cast<const bf16x8*>(cur + a_off[ks] + i * 2048);
    }
    dma_a(1, kb1, oth + AUNIT);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAITN) : "memory");
    CV_BARRIER();
    CV_MFMA(0);
    CV_BARRIER();
    // ---------------- phase 2: a1 x b ----------------
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i][ks] = *reinterpret_cast<const bf16x8*>(cur + AUNIT + a_off[ks] + i * 2048);
    dma_a(0, kb2, cur);
    dma_b(t + 2, cur + 2 * AUNIT);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAITN) : "memory");
    kb1 = kb2;
    kb2 = a_shift(t + 3);
    CV_BARRIER();
    CV_MFMA(1);
    CV_BARRIER();
  }
#undef CV_MFMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the tail DMA before the LDS is released
  if (wr < 2) CV_BARRIER();                           // re-balance the stagger

  // ---- epilogue: a lane owns ONE row and runs of 4 consecutive channels.  Bias once per column run; the residual rows of
  // a half (2 x NB fragments) are requested together before any is consumed (the fragment registers are dead by now): a
  // load -> wait -> store round trip per fragment would cost more than a third of a 96-channel tile's main loop ----
  float4 bs[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int64_t n = n0 + wc * 16 * NB + j * 16 + kq * 4;
    bs[j] = (p.bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    uint2 rs[2][NB];
    if (p.resid) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int64_t m = row0 + wr * 64 + hf * 32 + ii * 16 + fr;
        const int64_t mc = m < p.m1 ? m : p.m1 - 1;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int64_t n = n0 + wc * 16 * NB + j * 16 + kq * 4;
          const int64_t nc = n < p.N ? n : p.N - 4;
          rs[ii][j] = *reinterpret_cast<const uint2*>(p.resid + mc * p.ldr + nc);
        }
      }
    }
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int64_t m = row0 + wr * 64 + hf * 32 + ii * 16 + fr;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int64_t n = n0 + wc * 16 * NB + j * 16 + kq * 4;
        f32x4 v = acc[hf * 2 + ii][j];
        v[0] += bs[j].x; v[1] += bs[j].y; v[2] += bs[j].z; v[3] += bs[j].w;
        if (p.resid) {
          const uint2 r = rs[ii][j];
          v[0] += bf16lo_to_f32(r.x); v[1] += bf16hi_to_f32(r.x); v[2] += bf16lo_to_f32(r.y); v[3] += bf16hi_to_f32(r.y);
        }
        if (m < p.m1 && n < p.N)
          *reinterpret_cast<uint2*>(p.out + m * p.ldo + n) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      }
    }
  }
}

template <int NB>
int launch(const Params& p, hipStream_t st) {
  constexpr int BN = 32 * NB, NBP = (BN + 63) / 64, LDS = 2 * (2 * AUNIT + NBP * 8192);
  static icv_dev_flags attr_set = {};
  if (int rc = icv_ensure_dynamic_lds((const void*)conv_shift_kernel<NB>, LDS, &attr_set, "conv_shift")) return rc;
  hipLaunchKernelGGL((conv_shift_kernel<NB>), dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(512), LDS, st, p);
  return icv_check_launch("icv_conv3d_ndhwc");
}

}  // namespace cv

extern "C" int icv_conv3d_ndhwc(const void* x, int64_t ldx, int64_t x_rows_before, int64_t x_rows_after, const void* w, const float* bias,
                                const int64_t* tap_row_offsets, int64_t ntaps, int64_t cin, int64_t m0, int64_t m1, int64_t cout, void* out,
                                int64_t ldo, const void* resid, int64_t ldr, void* stream) {
  ICV_REQUIRE(x && w && out && tap_row_offsets, "icv_conv3d_ndhwc: null argument");
  ICV_REQUIRE(ntaps >= 1 && ntaps <= cv::MAX_TAPS, "icv_conv3d_ndhwc: 1..%d taps, got %lld", cv::MAX_TAPS, (long long)ntaps);
  ICV_REQUIRE(cin >= 32 && cin % 32 == 0 && cin <= 32 * 4096, "icv_conv3d_ndhwc: cin = %lld must be a multiple of 32 (pad the channels with zeros)", (long long)cin);
  ICV_REQUIRE(cout >= 4 && cout % 4 == 0, "icv_conv3d_ndhwc: cout = %lld must be a multiple of 4 (pad the filters with zeros)", (long long)cout);
  ICV_REQUIRE(m1 > m0 && m0 >= 0, "icv_conv3d_ndhwc: empty row range [%lld, %lld)", (long long)m0, (long long)m1);
  ICV_REQUIRE(ldx >= cin && ldx % 8 == 0 && ldo >= cout && ldo % 4 == 0 && (!resid || (ldr >= cout && ldr % 4 == 0)), "icv_conv3d_ndhwc: row strides");
  const int64_t halves = ntaps * (cin / 32);
  const int64_t K = (halves + 1) / 2 * 64;             // weight row length: zero-padded to whole K-tiles
  ICV_REQUIRE(halves < 65536, "icv_conv3d_ndhwc: K too large");
  cv::Params p;
  p.X = (const char*)x; p.ldx = ldx; p.W = (const char*)w; p.ldw = K; p.bias = bias;
  p.m0 = m0; p.m1 = m1; p.N = cout; p.out = (bf16_t*)out; p.ldo = ldo; p.resid = (const bf16_t*)resid; p.ldr = ldr;
  p.nt = (int)(K / 64); p.halves_per_tap = (int)(cin / 32); p.ntaps = (int)ntaps;
  p.magic = ((1ull << 32) + (uint64_t)p.halves_per_tap - 1) / (uint64_t)p.halves_per_tap;
  for (int i = 0; i < ntaps; ++i) {
    const int64_t off = tap_row_offsets[i];
    // every row the kernel touches must exist: the caller states how many addressable rows precede row 0 / follow row m1 - 1
    ICV_REQUIRE(m0 + off >= -x_rows_before && (m1 - 1) + off < m1 + x_rows_after, "icv_conv3d_ndhwc: tap %d (row offset %lld) leaves the buffer: rows [%lld, %lld) with %lld rows before row 0 and %lld after row %lld",
                i, (long long)off, (long long)m0, (long long)m1, (long long)x_rows_before, (long long)x_rows_after, (long long)(m1 - 1));
    p.tap_bytes[i] = off * ldx * 2;
  }
  ICV_REQUIRE((double)cout * (double)K * 2.0 < 4294967296.0, "icv_conv3d_ndhwc: the weight matrix spans >= 4 GiB");
  const int nb = cout <= 32 ? 1 : cout <= 96 ? 3 : (cout % 192 != 0 && cout % 96 == 0) ? 3 : 6;
  const int bn = 32 * nb;
  p.tiles_n = (int)((cout + bn - 1) / bn);
  hipStream_t st = (hipStream_t)stream;
  // The kernel keeps a lane's A-row position as a 32-bit byte offset from the operand base (the tap shifts are 64-bit), so ONE launch
  // covers rows whose offsets from ITS base stay below 4 GiB.  A longer volume (a full-resolution 192-channel decoder volume of a
  // 240 x 416 tile reaches 4 GiB at 109 frames; an untiled 480p volume exceeds it outright - ADVICE r5) is cut into row ranges, each
  // launched with the bases of x / out / resid moved to its first row: the arithmetic per output row is unchanged (same taps, same K
  // order), so the result is bit-identical to a hypothetical single launch.
  const int64_t max_rows = (int64_t)((4294967296.0 - 1.0) / ((double)ldx * 2.0)) / cv::BM * cv::BM;
  ICV_REQUIRE(max_rows >= cv::BM, "icv_conv3d_ndhwc: a row of %lld elements is too long", (long long)ldx);
  // [m0, m1) from base row 0 fits one launch when m1 <= max_rows (the common case); otherwise ranges of max_rows rows from m0 on
  for (int64_t a = (m1 <= max_rows ? 0 : m0); a < m1;) {
    const int64_t b = (m1 - a <= max_rows) ? m1 : a + max_rows;
    cv::Params q = p;
    q.X = p.X + a * ldx * 2;
    q.out = p.out + a * ldo;
    q.resid = p.resid ? p.resid + a * ldr : nullptr;
    q.m0 = (a == 0 ? m0 : 0);
    q.m1 = b - a;
    q.tiles_m = (int)((q.m1 - q.m0 + cv::BM - 1) / cv::BM);
    ICV_REQUIRE((int64_t)q.tiles_m * q.tiles_n <= 0x7fffffffLL, "icv_conv3d_ndhwc: too many tiles");
    int rc = 0;
    switch (nb) {
      case 1: rc = cv::launch<1>(q, st); break;
      case 3: rc = cv::launch<3>(q, st); break;
      default: rc = cv::launch<6>(q, st); break;
    }
    if (rc) return rc;
    a = b;
  }
  return 0;
}
