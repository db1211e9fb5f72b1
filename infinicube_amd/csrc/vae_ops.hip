// §8f-4: the Wan-VAE's channel RMS norm (+ SiLU) on NDHWC activations, one fused HBM pass.
// The VAE stays a stock PyTorch-ROCm module (north star); what surrounds its convolutions does not have to be five
// elementwise passes: upstream's RMS_norm is F.normalize(x, dim = channel) * sqrt(C) * gamma followed by nn.SiLU — as
// stock ops a norm reduction, a divide, two multiplies and the activation, each reading and writing the whole activation
// (measured: 24 % of a tiled decode, profiles/r04/vae_layer_tuning.md).  In NDHWC memory a pixel's C channels are
// contiguous, so this is a row kernel over [pixels, C]: 2 B read + 2 B written per element.
//   y = x / max(|x|_2, eps) * scale * gamma ;  act = 1: y <- y * sigmoid(y)        (fp32 inside, ONE rounding to bf16)
// Block = 256 threads = floor(256 / (C/8)) whole rows; a thread owns 8 consecutive channels (16 B); the row's sum of
// squares is combined through LDS (C/8 is 12 / 24 / 48 for the VAE's 96 / 192 / 384 channels: not a power of two, so no
// wave-shuffle segments).
#include "icv_common.h"

namespace {

__global__ __launch_bounds__(256) void rmsnorm_act_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                               const float* __restrict__ gamma, int64_t rows, int C, int lpr,
                                                               int rpb, float scale, float eps, int act, int Hp, int Wp, int pt) {
  __shared__ float part[256];
  __shared__ float rinv[64];
  const int tid = threadIdx.x;
  const int r_in = tid / lpr, within = tid - r_in * lpr;
  const int64_t row = (int64_t)blockIdx.x * rpb + r_in;
  const bool live = r_in < rpb && row < rows;
  // padded-volume form (Hp > 0): rows are the (t, h, w) positions of a [Tp, Hp, Wp] volume with `pt` leading padding frames and a
  // one-pixel spatial halo; halo / padding rows are WRITTEN AS ZEROS (they are the next convolution's zero padding), whatever
  // the input holds there (csrc/conv.hip leaves garbage on them)
  bool interior = true;
  if (Hp > 0 && live) {
    const int64_t f = row / ((int64_t)Hp * Wp);
    const int rem = (int)(row - f * ((int64_t)Hp * Wp));
    const int h = rem / Wp, w = rem - h * Wp;
    interior = f >= pt && h >= 1 && h <= Hp - 2 && w >= 1 && w <= Wp - 2;
  }
  float v[8];
  float ss = 0.f;
  if (live && interior) {
    const uint4 raw = *reinterpret_cast<const uint4*>(x + row * C + within * 8);
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
      ss += v[2 * i] * v[2 * i] + v[2 * i + 1] * v[2 * i + 1];
    }
  }
  part[tid] = ss;
  __syncthreads();
  if (live && interior && within == 0) {
    float s = 0.f;
    for (int i = 0; i < lpr; ++i) s += part[tid + i];
    rinv[r_in] = scale / fmaxf(sqrtf(s), eps);
  }
  __syncthreads();
  if (!live) return;
  if (!interior) {
    *reinterpret_cast<uint4*>(out + row * C + within * 8) = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  const float k = rinv[r_in];
  const float4 g0 = *reinterpret_cast<const float4*>(gamma + within * 8);
  const float4 g1 = *reinterpret_cast<const float4*>(gamma + within * 8 + 4);
  const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  float y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    y[i] = v[i] * k * g[i];
    if (act) y[i] = y[i] / (1.f + __expf(-y[i]));
  }
  *reinterpret_cast<uint4*>(out + row * C + within * 8) =
      make_uint4(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7]));
}

}  // namespace

extern "C" int icv_rmsnorm_act_rows(const void* x, void* out, const float* gamma, int64_t rows, int64_t C, float scale,
                                    float eps, int act, void* stream) {
  ICV_REQUIRE(x && out && gamma, "icv_rmsnorm_act_rows: null argument");
  ICV_REQUIRE(rows >= 0 && C >= 8 && C % 8 == 0 && C / 8 <= 256, "icv_rmsnorm_act_rows: C must be a multiple of 8 in [8, 2048], got %lld", (long long)C);
  ICV_REQUIRE(act == 0 || act == 1, "icv_rmsnorm_act_rows: act = 0 (none) | 1 (SiLU)");
  if (rows == 0) return 0;
  const int lpr = (int)(C / 8), rpb = 256 / lpr > 64 ? 64 : 256 / lpr;
  const int64_t blocks = (rows + rpb - 1) / rpb;
  ICV_REQUIRE(blocks <= 0x7fffffffLL, "icv_rmsnorm_act_rows: too many rows");
  hipLaunchKernelGGL(rmsnorm_act_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, gamma, rows,
                     (int)C, lpr, rpb, scale, eps, act, 0, 0, 0);
  return icv_check_launch("icv_rmsnorm_act_rows");
}

extern "C" int icv_rmsnorm_act_volume(const void* x, void* out, const float* gamma, int64_t Tp, int64_t Hp, int64_t Wp, int64_t pt, int64_t C,
                                      float scale, float eps, int act, void* stream) {
  ICV_REQUIRE(x && out && gamma, "icv_rmsnorm_act_volume: null argument");
  ICV_REQUIRE(C >= 8 && C % 8 == 0 && C / 8 <= 256, "icv_rmsnorm_act_volume: C must be a multiple of 8 in [8, 2048], got %lld", (long long)C);
  ICV_REQUIRE(Tp > pt && pt >= 0 && Hp >= 3 && Wp >= 3 && Hp < (1 << 15) && Wp < (1 << 15), "icv_rmsnorm_act_volume: bad volume [%lld, %lld, %lld], %lld padding frames",
              (long long)Tp, (long long)Hp, (long long)Wp, (long long)pt);
  ICV_REQUIRE(act == 0 || act == 1, "icv_rmsnorm_act_volume: act = 0 (none) | 1 (SiLU)");
  const int64_t rows = Tp * Hp * Wp;
  const int lpr = (int)(C / 8), rpb = 256 / lpr > 64 ? 64 : 256 / lpr;
  const int64_t blocks = (rows + rpb - 1) / rpb;
  ICV_REQUIRE(blocks <= 0x7fffffffLL, "icv_rmsnorm_act_volume: too many rows");
  hipLaunchKernelGGL(rmsnorm_act_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, gamma, rows,
                     (int)C, lpr, rpb, scale, eps, act, (int)Hp, (int)Wp, (int)pt);
  return icv_check_launch("icv_rmsnorm_act_volume");
}
