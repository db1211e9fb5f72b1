// bf16 MFMA GEMM with fused epilogues for the DiT projections.
//   C[M,N] = A[M,K] x W[N,K]^T, both operands K-contiguous (torch Linear layout) -> both MFMA
//   fragments are 16-byte K-slices, no transposes anywhere.
//
// v1 structure ("step-3" of the CDNA4 guide): 128x128x64 block tile, 4 waves (2x2), wave tile
// 64x64 = 4x4 fragments of v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//   * HBM -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR round trip.  The DMA destination
//     is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and
//     mirrored on the ds_read_b128 side (same involution both sides).
//   * LDS image per operand tile: [128 rows][8 chunks of 16 B]; physical chunk pc of row r holds
//     logical chunk pc ^ ((r>>1)&7)  -> the 16 lanes of a ds_read_b128 service group hit 16
//     distinct 16-B slots of the 256-B bank row (conflict-free).
//   * Operands are swapped in the MFMA (W fragment as A-operand, A fragment as B-operand), so a
//     lane ends up with 4 CONSECUTIVE n for one m: epilogue loads/stores are 8/16-byte vectors.
//   * 1-D grid, XCD-aware remap (block b runs on XCD b%8 -> give each XCD a contiguous chunk of
//     tiles) + grouped (8 m-tiles x all n-tiles) ordering so co-resident blocks share panels in L2.
#include <math.h>

#include "icv_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;

struct GemmParams {
  const bf16_t* A; int64_t lda;
  const bf16_t* W; int64_t ldw;
  const float* bias;
  int64_t M, N, K;
  void* out; int64_t ldo; int64_t nsplit; int64_t split_stride;
  const float* resid; int64_t ldr;
  const float* gate;
  int tiles_m, tiles_n;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// stage one 128x64 bf16 tile (rows row0.. of a [rows_total, ld] matrix, k-offset k0) into LDS
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ base, int64_t ld,
                                           int64_t row0, int64_t rows_total, int64_t k0,
                                           char* lds_tile, int tid) {
  const int wave = tid >> 6;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = p * 32 + (tid >> 3);
    const int pc = tid & 7;
    const int c = pc ^ ((r >> 1) & 7);
    int64_t gr = row0 + r;
    gr = gr < rows_total ? gr : rows_total - 1;  // clamp: tail rows read valid memory, never stored
    const bf16_t* src = base + gr * ld + k0 + c * 8;
    char* dst = lds_tile + p * 4096 + wave * 1024;  // wave-uniform; HW adds lane*16
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- block -> tile: bijective XCD remap, then grouped ordering ----
  const int nwg = p.tiles_m * p.tiles_n;
  int wg;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  constexpr int GM = 8;
  const int group_size = GM * p.tiles_n;
  const int g = wg / group_size;
  const int first_m = g * GM;
  const int gm = min(p.tiles_m - first_m, GM);
  const int tm = first_m + (wg % group_size) % gm;
  const int tn = (wg % group_size) / gm;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = (int)(p.K / BK);
  stage_tile(p.A, p.lda, m0, p.M, 0, smem, tid);
  stage_tile(p.W, p.ldw, n0, p.N, 0, smem + TILE_BYTES, tid);
  __syncthreads();

  // per-lane fragment addressing (constant over the K loop)
  const int fr = lane & 15;   // row within a 16-row fragment
  const int kq = lane >> 4;   // which 8-wide k-chunk of a 32-wide k-step

  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * STAGE_BYTES;
    if (t + 1 < nt) {
      char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
      stage_tile(p.A, p.lda, m0, p.M, (int64_t)(t + 1) * BK, nxt, tid);
      stage_tile(p.W, p.ldw, n0, p.N, (int64_t)(t + 1) * BK, nxt + TILE_BYTES, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + fr;
        const int pc = (ks * 4 + kq) ^ ((r >> 1) & 7);
        af[i] = *reinterpret_cast<const bf16x8*>(cur + r * 128 + pc * 16);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wn * 64 + j * 16 + fr;
        const int pc = (ks * 4 + kq) ^ ((r >> 1) & 7);
        wf[j] = *reinterpret_cast<const bf16x8*>(cur + TILE_BYTES + r * 128 + pc * 16);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();  // next tile landed (vmcnt(0) inside) and everyone is done with `cur`
  }

  // ---- epilogue: lane owns m = fr, n = kq*4 + 0..3 of each 16x16 fragment ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + i * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + kq * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      const int64_t off = icv_out_offset(m, n, p.ldo, p.N, p.nsplit, p.split_stride);
      if (EPI == ICV_EPI_BF16 || EPI == ICV_EPI_GELU_BF16) {
        if (EPI == ICV_EPI_GELU_BF16) {
          v[0] = gelu_tanh(v[0]); v[1] = gelu_tanh(v[1]); v[2] = gelu_tanh(v[2]); v[3] = gelu_tanh(v[3]);
        }
        *reinterpret_cast<uint2*>((bf16_t*)p.out + off) =
            make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      } else if (EPI == ICV_EPI_RESID_F32) {
        const float4 r = *reinterpret_cast<const float4*>(p.resid + m * p.ldr + n);
        float4 o;
        if (p.gate) {
          const float4 gt = *reinterpret_cast<const float4*>(p.gate + n);
          o = make_float4(r.x + gt.x * v[0], r.y + gt.y * v[1], r.z + gt.z * v[2], r.w + gt.w * v[3]);
        } else {
          o = make_float4(r.x + v[0], r.y + v[1], r.z + v[2], r.w + v[3]);
        }
        *reinterpret_cast<float4*>((float*)p.out + off) = o;
      } else {
        *reinterpret_cast<float4*>((float*)p.out + off) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

}  // namespace

int icv_gemm256w_dispatch(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                          int64_t M, int64_t N, int64_t K, int epilogue, void* out, int64_t ldo,
                          int64_t nsplit, int64_t split_stride, const float* resid, int64_t ldr,
                          const float* gate, hipStream_t st);
int icv_gemm256x_dispatch(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                          int64_t M, int64_t N, int64_t K, int epilogue, void* out, int64_t ldo,
                          int64_t nsplit, int64_t split_stride, const float* resid, int64_t ldr,
                          const float* gate, hipStream_t st);
int icv_gemm256_dispatch(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                         int64_t M, int64_t N, int64_t K, int epilogue, void* out, int64_t ldo,
                         int64_t nsplit, int64_t split_stride, const float* resid, int64_t ldr,
                         const float* gate, hipStream_t st);
int icv_gemm256p_dispatch(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                          int64_t M, int64_t N, int64_t K, int epilogue, void* out, int64_t ldo, int64_t nsplit,
                          int64_t split_stride, const float* resid, int64_t ldr, const float* gate, int mode, hipStream_t st);
int icv_gemm256p_cus();

extern "C" int icv_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw,
                             const float* bias, int64_t M, int64_t N, int64_t K, int epilogue,
                             void* out, int64_t ldo, int64_t nsplit, int64_t split_stride,
                             const float* resid, int64_t ldr, const float* gate, void* stream) {
  ICV_REQUIRE(A && W && out, "icv_gemm_bf16: null pointer");
  ICV_REQUIRE(M > 0 && N > 0 && K >= 64 && K % 64 == 0, "icv_gemm_bf16: K=%lld must be a positive multiple of 64", (long long)K);
  ICV_REQUIRE(N < (1LL << 31), "icv_gemm_bf16: N=%lld too large", (long long)N);
  ICV_REQUIRE(N % 4 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ldo % 4 == 0, "icv_gemm_bf16: N%%4, lda%%8, ldw%%8, ldo%%4 alignment");
  // both kernels keep per-lane A / W source offsets as 32-bit byte offsets from the operand base
  ICV_REQUIRE((double)N * (double)ldw * 2.0 < 4294967296.0, "icv_gemm_bf16: the weight operand spans >= 4 GiB (N*ldw)");
  if ((double)M * (double)lda * 2.0 >= 4294967296.0) {
    // an A operand of >= 4 GiB (e.g. both CFG forwards of a 720p clip as one FFN2 launch: 172 800 rows x 13 824): the same launch over
    // row ranges of < 4 GiB each, cut on the 256-row tile grid (round 6; it used to be an error asking the caller to split)
    const int64_t max_rows = (int64_t)(4294967295.0 / ((double)lda * 2.0)) / 256 * 256;
    ICV_REQUIRE(max_rows >= 256, "icv_gemm_bf16: lda=%lld too large", (long long)lda);
    const int64_t out_elem = (epilogue == ICV_EPI_BF16 || epilogue == ICV_EPI_GELU_BF16) ? 2 : 4;
    for (int64_t m0 = 0; m0 < M; m0 += max_rows) {
      const int64_t rows = M - m0 < max_rows ? M - m0 : max_rows;
      if (int rc = icv_gemm_bf16((const char*)A + m0 * lda * 2, lda, W, ldw, bias, rows, N, K, epilogue, (char*)out + m0 * ldo * out_elem, ldo, nsplit,
                                 split_stride, resid ? resid + m0 * ldr : nullptr, ldr, gate, stream))
        return rc;
    }
    return 0;
  }
  if (nsplit <= 0) nsplit = N;
  ICV_REQUIRE(nsplit % 4 == 0 && N % nsplit == 0, "icv_gemm_bf16: nsplit must divide N and be a multiple of 4");
  ICV_REQUIRE(epilogue != ICV_EPI_RESID_F32 || (resid && ldr % 4 == 0 && nsplit == N), "icv_gemm_bf16: RESID epilogue needs resid, ldr%%4==0, no split");
  // Kernel choice.  The 256x256 8-wave 4-phase kernel (gemm256.hip) is ~1.35x faster per tile-FLOP than the
  // 128x128 kernel below, but runs 1 block per CU: with few tiles the last wave of blocks is mostly empty
  // (e.g. M=4680, N=5120 at 8-way sequence parallel = 380 tiles = 1.48 rounds of 256 CUs).  Pick by
  // quantisation-adjusted throughput; option gemm256 = 0 / 1 forces one kernel, 2 (default) = heuristic.
  if (N % 256 == 0 && M >= 256) {
    const int mode = icv_get_option_int("gemm256", 2);
    if (mode == 3) {  // experiment: 4 waves x (128 x 128), one wave per SIMD (experiments/gemm256w.hip)
#ifdef ICV_EXPERIMENTS
      return icv_gemm256w_dispatch(A, lda, W, ldw, bias, M, N, K, epilogue, out, ldo, nsplit, split_stride,
                                   resid, ldr, gate, (hipStream_t)stream);
#else
      icv_set_error("gemm256 = 3 is an experiment: rebuild libicvideo with ICV_EXPERIMENTS=1");
      return 1;
#endif
    }
    if (mode == 4) {  // experiment: 4 waves x (128 x 128), whole-tile double buffer, one barrier per K-tile (experiments/gemm256x.hip)
#ifdef ICV_EXPERIMENTS
      return icv_gemm256x_dispatch(A, lda, W, ldw, bias, M, N, K, epilogue, out, ldo, nsplit, split_stride,
                                   resid, ldr, gate, (hipStream_t)stream);
#else
      icv_set_error("gemm256 = 4 is an experiment: rebuild libicvideo with ICV_EXPERIMENTS=1");
      return 1;
#endif
    }
    if (mode == 5 || mode == 6) {  // A/B: gemm256's schedule as a persistent kernel for EVERY epilogue (5: static stride, 6: per-XCD work counter)
      const int rc = icv_gemm256p_dispatch(A, lda, W, ldw, bias, M, N, K, epilogue, out, ldo, nsplit, split_stride, resid, ldr, gate, mode,
                                           (hipStream_t)stream);
      if (rc >= 0) return rc;
      return icv_gemm256_dispatch(A, lda, W, ldw, bias, M, N, K, epilogue, out, ldo, nsplit, split_stride, resid, ldr, gate, (hipStream_t)stream);
    }
    bool use256 = mode == 1;
    if (mode == 2) {
      static int n_cu = 0;
      if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
      }
      const double t256 = (double)((M + 255) / 256) * (double)(N / 256);
      const double t128 = (double)((M + 127) / 128) * (double)((N + 127) / 128);
      const double slots256 = n_cu, slots128 = 2.0 * n_cu;     // blocks resident per round
      const double eff256 = t256 / (ceil(t256 / slots256) * slots256) * 1.35;
      const double eff128 = t128 / (ceil(t128 / slots128) * slots128) * 1.00;
      // padding waste of the partial last m-tile
      const double use_m256 = (double)M / (double)(((M + 255) / 256) * 256), use_m128 = (double)M / (double)(((M + 127) / 128) * 128);
      use256 = eff256 * use_m256 >= eff128 * use_m128;
    }
    // the persistent form of the same kernel (gemm256p.hip) where it measures faster: bf16 / GELU epilogues (nothing but stores
    // after the main loop, so the next tile's prologue hides under them: +1.5...2.9 % at 14B, +3...5 % at 1.3B) with at least two
    // tiles per CU; bit-identical results.  gemm256_persist = 0 switches it off.
    if (use256 && mode == 2 && (epilogue == ICV_EPI_BF16 || epilogue == ICV_EPI_GELU_BF16) && icv_get_option_int("gemm256_persist", 1) != 0 &&
        (int64_t)((M + 255) / 256) * (N / 256) >= 2 * (int64_t)icv_gemm256p_cus()) {
      const int rc = icv_gemm256p_dispatch(A, lda, W, ldw, bias, M, N, K, epilogue, out, ldo, nsplit, split_stride, resid, ldr, gate, 6,
                                           (hipStream_t)stream);
      if (rc >= 0) return rc;      // -1: no counter block for this stream right now -> the one-tile-per-block launch below
    }
    if (use256)
      return icv_gemm256_dispatch(A, lda, W, ldw, bias, M, N, K, epilogue, out, ldo, nsplit, split_stride,
                                  resid, ldr, gate, (hipStream_t)stream);
  }
  GemmParams p;
  p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.out = out; p.ldo = ldo; p.nsplit = nsplit; p.split_stride = split_stride;
  p.resid = resid; p.ldr = ldr; p.gate = gate;
  p.tiles_m = (int)((M + BM - 1) / BM);
  p.tiles_n = (int)((N + BN - 1) / BN);
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  ICV_REQUIRE(nwg < (1LL << 31), "icv_gemm_bf16: grid too large");
  dim3 grid((unsigned)nwg), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case ICV_EPI_BF16: hipLaunchKernelGGL(gemm_bf16_kernel<ICV_EPI_BF16>, grid, block, 0, st, p); break;
    case ICV_EPI_GELU_BF16: hipLaunchKernelGGL(gemm_bf16_kernel<ICV_EPI_GELU_BF16>, grid, block, 0, st, p); break;
    case ICV_EPI_RESID_F32: hipLaunchKernelGGL(gemm_bf16_kernel<ICV_EPI_RESID_F32>, grid, block, 0, st, p); break;
    case ICV_EPI_F32: hipLaunchKernelGGL(gemm_bf16_kernel<ICV_EPI_F32>, grid, block, 0, st, p); break;
    default: icv_set_error("icv_gemm_bf16: unknown epilogue %d", epilogue); return 1;
  }
  return icv_check_launch("icv_gemm_bf16");
}
