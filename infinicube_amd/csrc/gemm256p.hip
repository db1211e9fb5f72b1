// gemm256's default schedule as a PERSISTENT kernel (round 5; VERDICT r4 item 6 "one more bounded, different attempt").
// One work-group per CU loops over tiles; the prologue DMAs of tile i+1 (6 of the 8 units of its first two K-tiles) are issued
// BEFORE the epilogue of tile i, so the block launch, the first operand fetch and the pipeline fill of every tile but the first
// run under the previous tile's stores.  Per 256x256 tile of the 14B shapes the main loop is ~ 122 us of a ~ 127 us tile slot;
// what is left outside it is what this variant hides.
// Same tile geometry, LDS image, swizzle, two-phase main loop, counted vmcnt and wave-group stagger as gemm256.hip (MF = 16, SCH = 3),
// bit-identical results (same MFMA order per accumulator).  Tile order: XCD x owns the same contiguous range of the grouped tile
// order as in the one-tile-per-block launch; its work-groups take the next tile of that range from a per-XCD counter when they START
// their current tile (DYN), so tiles start in the order in which CUs free up - as with the hardware's own dispatch - and the set of
// tiles in flight per XCD (the L2 picture) is unchanged.  With a static stride instead (DYN = false, option gemm256 = 5) the XCD's
// work-groups drift apart and lose their shared panels: -6 % on QKV / FFN1.
// Measured, same box, interleaved (profiles/r05/gemm_persistent_ab.txt): bf16 / GELU epilogues +1.5...2.9 % on the 14B shapes,
// +3...5 % on the 1.3B ones, +1.3...2.4 % at the sp4 shard; gated-residual epilogue: a tie (+0.1...0.3 % once its register spills
// were removed; -0.5...2.6 % before).  So icv_gemm_bf16 uses this kernel for the bf16 / GELU epilogues when a launch has at least two tiles per CU, and
// gemm256.hip otherwise (option gemm256_persist = 0 switches it off; gemm256 = 5 / 6 force it for every epilogue).
//
// (A hipGraph captured on stream S keeps S's counter block: replay it on S, as the loop does.  Replaying it on ANOTHER stream
// concurrently with eager GEMMs on S would make two launches share one block - serialise such a replay against S, or capture it on the
// stream it will run on.)
// The work counters are STATELESS between launches: a 64-byte block (8 per-XCD counters + an exit counter) that every launch finds
// zeroed and whose last-exiting work-group zeroes again - safe under hipGraph replay; one block per (device, stream) from a pool that
// is allocated outside stream capture, so launches on different streams never share counters.
#include <map>
#include <mutex>
#include <utility>

#include "icv_common.h"

namespace g256p {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int UNIT_BYTES = 128 * 128;
constexpr int STAGE_BYTES = 4 * UNIT_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;
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
  int gm;
  // DYN: the next tile of an XCD's range is taken from a per-XCD counter when a work-group starts its current tile.  ctr[0..7] = the
  // counters, ctr[8] = work-groups that have left; all zero at launch, zeroed again by the last work-group to leave.
  unsigned* ctr;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

#define P_BARRIER()                         \
  do {                                      \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_s_barrier();           \
    asm volatile("" ::: "memory");          \
  } while (0)

__device__ __forceinline__ void dma_unit(const char* __restrict__ base, const unsigned (&off)[2], int64_t kbyte, char* lds_unit, int wave) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const char* src = base + (int64_t)off[q] + kbyte;
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(lds_unit + q * 8192 + wave * 1024), 16, 0, 0);
  }
}

template <int EPI, bool DYN>
__global__ __launch_bounds__(512) void gemm256p_kernel(Params p) {
  __shared__ int next_li;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, kq = lane >> 4;

  // this work-group's chain of tiles: XCD x = bid & 7 owns [start, start + len) of the grouped order; local index l strides by L
  const int nwg = p.tiles_m * p.tiles_n;
  const int xcd = blockIdx.x & 7, l0 = blockIdx.x >> 3, L = (gridDim.x + 7 - xcd) >> 3;      // work-groups of this launch on XCD x
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int len = q8 + (xcd < r8 ? 1 : 0);
  const int GM = p.gm, group_size = GM * p.tiles_n;
  auto tile_of = [&](int wg, int64_t& m0, int64_t& n0) {
    const int g = wg / group_size;
    const int first_m = g * GM;
    const int gm = min(p.tiles_m - first_m, GM);
    m0 = (int64_t)(first_m + (wg % group_size) % gm) * BM;
    n0 = (int64_t)((wg % group_size) / gm) * BN;
  };
  auto offsets = [&](int64_t m0, int64_t n0, unsigned (&oA)[2][2], unsigned (&oB)[2][2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = q * 64 + (tid >> 3);
      const int pc = tid & 7;
      const int c = pc ^ ((u >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int64_t ra = m0 + (u >> 6) * 128 + h * 64 + (u & 63);
        ra = ra < p.M ? ra : p.M - 1;
        oA[h][q] = (unsigned)((ra * p.lda + c * 8) * 2);
        int64_t rb = n0 + (u >> 5) * 64 + h * 32 + (u & 31);
        rb = rb < p.N ? rb : p.N - 1;
        oB[h][q] = (unsigned)((rb * p.ldw + c * 8) * 2);
      }
    }
  };
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Wb = reinterpret_cast<const char*>(p.W);
  const int nt = (int)(p.K / BK);
  auto kbyte = [&](int t) -> int64_t { return (int64_t)(t < nt ? t : nt - 1) * (BK * 2); };
  auto prologue = [&](const unsigned (&oA)[2][2], const unsigned (&oB)[2][2]) {      // tile's K-tile 0 complete + A0, B0 of K-tile 1
    dma_unit(Ab, oA[0], kbyte(0), smem + U_A0 * UNIT_BYTES, wave);
    dma_unit(Wb, oB[0], kbyte(0), smem + U_B0 * UNIT_BYTES, wave);
    dma_unit(Wb, oB[1], kbyte(0), smem + U_B1 * UNIT_BYTES, wave);
    dma_unit(Ab, oA[1], kbyte(0), smem + U_A1 * UNIT_BYTES, wave);
    dma_unit(Ab, oA[0], kbyte(1), smem + STAGE_BYTES + U_A0 * UNIT_BYTES, wave);
    dma_unit(Wb, oB[0], kbyte(1), smem + STAGE_BYTES + U_B0 * UNIT_BYTES, wave);
  };

  const int ar = wr * 64 + fr, br = wc * 32 + fr;
  int a_off[2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_off[ks] = ar * 128 + (((ks * 4 + kq) ^ ((ar >> 1) & 7)) << 4);
    b_off[ks] = br * 128 + (((ks * 4 + kq) ^ ((br >> 1) & 7)) << 4);
  }
  constexpr int FROWS = 16 * 128;

  auto fetch = [&]() -> int {          // DYN: index inside this XCD's range of the next tile to start (>= len: none left); block-uniform
    if (tid == 0) next_li = (int)atomicAdd(p.ctr + xcd, 1u);
    __syncthreads();
    const int v = next_li;
    __syncthreads();
    return v;
  };
  auto leave = [&]() {                 // DYN: this work-group fetches no more; the last one to leave resets the block for the next launch
    if (DYN && tid == 0) {
      __threadfence();
      if (atomicAdd(p.ctr + 8, 1u) == gridDim.x - 1) {
        __threadfence();
#pragma unroll
        for (int x = 0; x < 9; ++x) atomicExch(p.ctr + x, 0u);
      }
    }
  };
  int li = DYN ? fetch() : l0;
  if (li >= len) {
    leave();
    return;
  }
  int64_t m0, n0;
  unsigned offA[2][2], offB[2][2];
  tile_of(start + li, m0, n0);
  offsets(m0, n0, offA, offB);
  prologue(offA, offB);
  bool first = true;

  for (;;) {
    const int li_next = DYN ? fetch() : li + L;        // taken NOW: tiles start in the order in which work-groups become free
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A0(0), B0(0), B1(0) must have landed.  First tile: exactly the 12 prologue instructions are in flight -> vmcnt(6); later tiles:
    // the prologue was issued in front of the previous epilogue's loads and stores, which are younger -> everything
    if (first) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    first = false;
    P_BARRIER();
    if (wr == 1) P_BARRIER();          // stagger: group 1 runs one barrier behind group 0

    bf16x8 af[4][2], b0f[2][2], b1f[2][2];
#define P_MFMA(AH, BF, BH)                                                                      \
  {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    __builtin_amdgcn_s_setprio(1);                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                               \
      acc[(AH) * 4 + i][(BH) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[j][ks], af[i][ks], acc[(AH) * 4 + i][(BH) * 2 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  }
    for (int t = 0; t < nt; ++t) {
      char* cur = smem + (t & 1) * STAGE_BYTES;
      char* oth = smem + ((t + 1) & 1) * STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          b0f[j][ks] = *reinterpret_cast<const bf16x8*>(cur + U_B0 * UNIT_BYTES + b_off[ks] + j * FROWS);
          b1f[j][ks] = *reinterpret_cast<const bf16x8*>(cur + U_B1 * UNIT_BYTES + b_off[ks] + j * FROWS);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i][ks] = *reinterpret_cast<const bf16x8*>(cur + U_A0 * UNIT_BYTES + a_off[ks] + i * FROWS);
      }
      dma_unit(Wb, offB[1], kbyte(t + 1), oth + U_B1 * UNIT_BYTES, wave);
      dma_unit(Ab, offA[1], kbyte(t + 1), oth + U_A1 * UNIT_BYTES, wave);
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      P_BARRIER();
      P_MFMA(0, b0f, 0);
      P_MFMA(0, b1f, 1);
      P_BARRIER();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i][ks] = *reinterpret_cast<const bf16x8*>(cur + U_A1 * UNIT_BYTES + a_off[ks] + i * FROWS);
      dma_unit(Ab, offA[0], kbyte(t + 2), cur + U_A0 * UNIT_BYTES, wave);
      dma_unit(Wb, offB[0], kbyte(t + 2), cur + U_B0 * UNIT_BYTES, wave);
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      P_BARRIER();
      P_MFMA(1, b0f, 0);
      P_MFMA(1, b1f, 1);
      P_BARRIER();
    }
#undef P_MFMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail tiles' re-loads still target the LDS
    if (wr == 0) P_BARRIER();                           // re-balance the stagger: every wave is past its last LDS read

    // ---- the NEXT tile's first operands start moving now, under this tile's epilogue ----
    const int64_t cm0 = m0, cn0 = n0;
    const bool has_next = li_next < len;
    // gated-residual epilogue: the next tile's offsets and prologue wait until the first half of the epilogue is done (its 16 residual
    // loads + 64 accumulator registers + the offsets do not fit in 256 VGPRs together: computing the offsets here spilled 25 registers);
    // the other epilogues only store, so their prologue starts right away
    if (has_next && EPI != ICV_EPI_RESID_F32) {
      tile_of(start + li_next, m0, n0);
      offsets(m0, n0, offA, offB);
      prologue(offA, offB);
    }

    // ---- epilogue of the tile just computed (gemm256.hip's, for the four epilogues) ----
    float4 bs[4], gt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = cn0 + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + kq * 4;
      bs[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      gt[j] = (EPI == ICV_EPI_RESID_F32 && p.gate) ? *reinterpret_cast<const float4*>(p.gate + n) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if (EPI == ICV_EPI_RESID_F32 && hf == 1 && has_next) {      // the first half's accumulators are dead: room for the offsets
        tile_of(start + li_next, m0, n0);
        offsets(m0, n0, offA, offB);
        prologue(offA, offB);
      }
      // gated residual: the residual rows are loaded in batches of 8 float4 (two 16-row fragments) ahead of their use - gemm256.hip batches
      // 16, but this kernel carries the loop state of the persistent schedule on top and 16 spill 25 registers
#pragma unroll
      for (int qh = 0; qh < 2; ++qh) {
        float4 rs[2][4];
        if (EPI == ICV_EPI_RESID_F32) {
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2) {
            const int64_t m = cm0 + wr * 128 + hf * 64 + (qh * 2 + i2) * 16 + fr;
            const int64_t mc = m < p.M ? m : p.M - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              rs[i2][j] = *reinterpret_cast<const float4*>(p.resid + mc * p.ldr + cn0 + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + kq * 4);
          }
        }
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          const int ii = qh * 2 + i2;
          const int64_t m = cm0 + wr * 128 + hf * 64 + ii * 16 + fr;
          if (m >= p.M) continue;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t n = cn0 + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + kq * 4;
            const f32x4 a = acc[hf * 4 + ii][j];
            float v0 = a[0] + bs[j].x, v1 = a[1] + bs[j].y, v2 = a[2] + bs[j].z, v3 = a[3] + bs[j].w;
            const int64_t off = icv_out_offset(m, n, p.ldo, p.N, p.nsplit, p.split_stride);
            if (EPI == ICV_EPI_BF16 || EPI == ICV_EPI_GELU_BF16) {
              if (EPI == ICV_EPI_GELU_BF16) { v0 = gelu_tanh(v0); v1 = gelu_tanh(v1); v2 = gelu_tanh(v2); v3 = gelu_tanh(v3); }
              *reinterpret_cast<uint2*>((bf16_t*)p.out + off) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
            } else if (EPI == ICV_EPI_RESID_F32) {
              const float4 r = rs[i2][j];
              *reinterpret_cast<float4*>((float*)p.out + off) = make_float4(r.x + gt[j].x * v0, r.y + gt[j].y * v1, r.z + gt[j].z * v2, r.w + gt[j].w * v3);
            } else {
              *reinterpret_cast<float4*>((float*)p.out + off) = make_float4(v0, v1, v2, v3);
            }
          }
        }
      }
    }
    if (li_next >= len) break;
    li = li_next;
  }
  leave();
}

template <int EPI, bool DYN>
int launch1(const Params& p, unsigned grid, hipStream_t st) {
  static icv_dev_flags attr_set = {};
  if (int rc = icv_ensure_dynamic_lds((const void*)gemm256p_kernel<EPI, DYN>, LDS_BYTES, &attr_set, "gemm256p")) return rc;
  hipLaunchKernelGGL((gemm256p_kernel<EPI, DYN>), dim3(grid), dim3(512), LDS_BYTES, st, p);
  return icv_check_launch("icv_gemm_bf16(256p)");
}

// ---- counter blocks: one per (device, stream), from a per-device pool allocated outside stream capture ----
constexpr int kBlockWords = 16, kPoolBlocks = 256;      // 16 KiB per device; a stream handle keeps its block for the life of the process
struct CounterPool {
  unsigned* base = nullptr;
  int used = 0;
  bool failed = false;
};
std::mutex g_mu;
std::map<int, CounterPool> g_pools;
std::map<std::pair<int, hipStream_t>, unsigned*> g_blocks;

// nullptr = no block available right now (pool not yet allocated while `st` is capturing, pool exhausted, allocation failed):
// the caller launches the one-tile-per-block kernel instead - never an error.
unsigned* counters_for(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_blocks.find({dev, st});
  if (it != g_blocks.end()) return it->second;
  CounterPool& pool = g_pools[dev];
  if (!pool.base) {
    if (pool.failed) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return nullptr;                                        // no allocation inside a capture; a later eager launch makes the pool
    }
    const size_t bytes = (size_t)kPoolBlocks * kBlockWords * sizeof(unsigned);
    if (hipMalloc((void**)&pool.base, bytes) != hipSuccess || hipMemset(pool.base, 0, bytes) != hipSuccess) {
      (void)hipGetLastError();
      pool.base = nullptr;
      pool.failed = true;
      return nullptr;
    }
  }
  if (pool.used >= kPoolBlocks) return nullptr;
  unsigned* blk = pool.base + (size_t)(pool.used++) * kBlockWords;
  g_blocks[{dev, st}] = blk;
  return blk;
}

template <int EPI>
int launch(Params& p, int n_cu, bool dyn, hipStream_t st) {
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  const unsigned grid = (unsigned)(nwg < n_cu ? nwg : n_cu);
  // the tile space is cut into 8 ranges keyed by blockIdx.x & 7 (one per XCD): a launch with fewer than 8 work-groups but more tiles
  // than work-groups (gemm256p_cus < 8, a device reporting < 8 CUs) would leave whole ranges without a work-group and their tiles
  // unwritten - such a launch is not this kernel's business: the one-tile-per-block kernel takes it
  if (grid < 8 && nwg > (int64_t)grid) return -1;
  if (!dyn) return launch1<EPI, false>(p, grid, st);
  p.ctr = counters_for(st);
  if (!p.ctr) return -1;                                     // caller falls back to gemm256.hip
  return launch1<EPI, true>(p, grid, st);
}

}  // namespace g256p

int icv_gemm256p_cus() {      // CUs of the CURRENT device (cached per device: a process may drive several)
  static int n_cu[ICV_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ICV_MAX_DEVICES) return 256;
  if (!n_cu[dev]) {
    hipDeviceProp_t prop;
    n_cu[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n_cu[dev];
}

// mode: 5 = static stride, 6 = per-XCD work counter.  Returns -1 (no error text) when mode 6 has no counter block for `st` right
// now: the caller launches the one-tile-per-block kernel.
int icv_gemm256p_dispatch(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int64_t M, int64_t N, int64_t K, int epilogue,
                          void* out, int64_t ldo, int64_t nsplit, int64_t split_stride, const float* resid, int64_t ldr, const float* gate,
                          int mode, hipStream_t st) {
  g256p::Params p;
  p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.out = out; p.ldo = ldo; p.nsplit = nsplit; p.split_stride = split_stride;
  p.resid = resid; p.ldr = ldr; p.gate = gate;
  p.tiles_m = (int)((M + g256p::BM - 1) / g256p::BM);
  p.tiles_n = (int)((N + g256p::BN - 1) / g256p::BN);
  p.gm = icv_get_option_int("gemm256_gm", 4);
  const int n_cu = icv_gemm256p_cus();
  const int cus = icv_get_option_int("gemm256p_cus", n_cu);
  const bool dyn = mode == 6;
  p.ctr = nullptr;
  switch (epilogue) {
    case ICV_EPI_BF16: return g256p::launch<ICV_EPI_BF16>(p, cus, dyn, st);
    case ICV_EPI_GELU_BF16: return g256p::launch<ICV_EPI_GELU_BF16>(p, cus, dyn, st);
    case ICV_EPI_RESID_F32: return g256p::launch<ICV_EPI_RESID_F32>(p, cus, dyn, st);
    case ICV_EPI_F32: return g256p::launch<ICV_EPI_F32>(p, cus, dyn, st);
  }
  icv_set_error("icv_gemm_bf16: unknown epilogue %d", epilogue);
  return 1;
}
