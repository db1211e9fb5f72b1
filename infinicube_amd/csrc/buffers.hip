// Guidance-buffer producers feeding the hot path: the coordinate buffer of
// `generate_coordinate_buffer_from_memory_global_norm` [R infinicube/utils/buffer_utils.py:180-265] with
// `unproject_depth_torch` [R infinicube/utils/depth_utils.py:402-466] fused in.  HBM-bound, one thread per
// pixel, the [N,H,W,3] fp32 point map is never materialised: points are recomputed from the depth map
// (4 B/pixel read) in the sampling pass and in the normalisation pass, and the final pass can emit the
// uint8 buffer the video pipeline consumes directly (3 B/pixel written instead of 12 + a host round trip).
//   X_cam  = depth * Kinv (x, y, 1)^T        X_cam0 = T_n (X_cam, 1)^T,   T_n = pose_0^-1 pose_n
#include "icv_common.h"

namespace {

struct CoordParams {
  const float* depth;   // [N, H, W]
  const float* tf;      // [N, 16] row-major 4x4 camera-n -> camera-0
  float kinv[9];
  int64_t N, H, W;
};

// (a0*b0 + a1*b1) + a2*b2 with every operation rounded on its own
__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  return __fadd_rn(__fadd_rn(__fmul_rn(a0, b0), __fmul_rn(a1, b1)), __fmul_rn(a2, b2));
}

__device__ __forceinline__ void point_cam0(const CoordParams& p, int64_t pix, float d, float out[3]) {
  const int64_t hw = p.H * p.W;
  const int n = (int)(pix / hw);
  const int64_t r = pix - (int64_t)n * hw;
  const float y = (float)(r / p.W), x = (float)(r % p.W);
  // rays = Kinv (x, y, 1).  Byte outputs must match the reference bit for bit, so every product and sum is rounded
  // separately, left to right, exactly like the reference's torch.matmul on these 3x3 / 4x4 operands (verified against
  // its CPU output: no fused multiply-add anywhere; tests/golden/make_coord_buffer_golden.py) — hence the explicit
  // round-to-nearest intrinsics instead of expressions hipcc may contract (the TU is also built -ffp-contract=off).
  const float rx = dot3(p.kinv[0], x, p.kinv[1], y, p.kinv[2], 1.0f);
  const float ry = dot3(p.kinv[3], x, p.kinv[4], y, p.kinv[5], 1.0f);
  const float rz = dot3(p.kinv[6], x, p.kinv[7], y, p.kinv[8], 1.0f);
  const float cx = __fmul_rn(d, rx), cy = __fmul_rn(d, ry), cz = __fmul_rn(d, rz);
  const float* m = p.tf + (int64_t)n * 16;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    out[i] = __fadd_rn(dot3(m[i * 4 + 0], cx, m[i * 4 + 1], cy, m[i * 4 + 2], cz), __fmul_rn(m[i * 4 + 3], 1.0f));
}

__global__ __launch_bounds__(256) void coord_valid_mask_kernel(CoordParams p, unsigned char* __restrict__ mask, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float d = p.depth[i];
  float pt[3];
  point_cam0(p, i, d, pt);
  mask[i] = (d != 0.f && pt[2] < 1e6f) ? 1 : 0;   // depth 0 = infinitely far (set to 1e7, filtered by z < 1e6)
}

__global__ __launch_bounds__(256) void coord_gather_kernel(CoordParams p, const int64_t* __restrict__ idx, int64_t n,
                                                           float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t pix = idx[i];
  float pt[3];
  point_cam0(p, pix, p.depth[pix], pt);
  out[i * 3 + 0] = pt[0]; out[i * 3 + 1] = pt[1]; out[i * 3 + 2] = pt[2];
}

__global__ __launch_bounds__(256) void coord_normalize_kernel(CoordParams p, float mn0, float mn1, float mn2, float rg0,
                                                              float rg1, float rg2, int has_valid,
                                                              float* __restrict__ out_f32,
                                                              unsigned char* __restrict__ out_u8, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float d = p.depth[i];
  float v[3];
  if (d == 0.f) {
    v[0] = v[1] = v[2] = 1.0f;                      // sky
  } else {
    float pt[3];
    point_cam0(p, i, d, pt);
    const float mn[3] = {mn0, mn1, mn2}, rg[3] = {rg0, rg1, rg2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (has_valid) {
        // (clamp((pt - min) / range * 2 - 1, -1, 1) + 1) / 2, torch's op order [R infinicube/utils/buffer_utils.py:246-252]
        float t = __fsub_rn(__fmul_rn(__fdiv_rn(__fsub_rn(pt[c], mn[c]), rg[c]), 2.0f), 1.0f);
        t = fminf(fmaxf(t, -1.0f), 1.0f);
        v[c] = __fdiv_rn(__fadd_rn(t, 1.0f), 2.0f);
      } else {
        v[c] = __fmul_rn(pt[c], 0.5f);
      }
    }
  }
  if (out_f32) { out_f32[i * 3 + 0] = v[0]; out_f32[i * 3 + 1] = v[1]; out_f32[i * 3 + 2] = v[2]; }
  if (out_u8) {   // the caller's `(buffer * 255).astype(np.uint8)`: truncation toward zero
    out_u8[i * 3 + 0] = (unsigned char)(int)__fmul_rn(v[0], 255.0f);
    out_u8[i * 3 + 1] = (unsigned char)(int)__fmul_rn(v[1], 255.0f);
    out_u8[i * 3 + 2] = (unsigned char)(int)__fmul_rn(v[2], 255.0f);
  }
}

int fill(CoordParams& p, const float* depth, const float* kinv, const float* tf, int64_t N, int64_t H, int64_t W) {
  ICV_REQUIRE(depth && kinv && tf && N > 0 && H > 0 && W > 0, "icv_coord_*: bad arguments");
  p.depth = depth; p.tf = tf; p.N = N; p.H = H; p.W = W;
  for (int i = 0; i < 9; ++i) p.kinv[i] = kinv[i];
  return 0;
}

}  // namespace

extern "C" int icv_coord_valid_mask(const float* depth, const float* kinv_host9, const float* cam_to_cam0,
                                    int64_t N, int64_t H, int64_t W, unsigned char* mask, void* stream) {
  CoordParams p;
  if (int rc = fill(p, depth, kinv_host9, cam_to_cam0, N, H, W)) return rc;
  const int64_t total = N * H * W;
  hipLaunchKernelGGL(coord_valid_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, mask, total);
  return icv_check_launch("icv_coord_valid_mask");
}

extern "C" int icv_coord_gather_points(const float* depth, const float* kinv_host9, const float* cam_to_cam0,
                                       int64_t N, int64_t H, int64_t W, const int64_t* pixel_index, int64_t n,
                                       float* out, void* stream) {
  CoordParams p;
  if (int rc = fill(p, depth, kinv_host9, cam_to_cam0, N, H, W)) return rc;
  ICV_REQUIRE(n > 0 && pixel_index && out, "icv_coord_gather_points: bad arguments");
  hipLaunchKernelGGL(coord_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, pixel_index, n, out);
  return icv_check_launch("icv_coord_gather_points");
}

extern "C" int icv_coord_normalize(const float* depth, const float* kinv_host9, const float* cam_to_cam0,
                                   int64_t N, int64_t H, int64_t W, const float* mins_host3,
                                   const float* ranges_host3, int has_valid, float* out_f32,
                                   unsigned char* out_u8, void* stream) {
  CoordParams p;
  if (int rc = fill(p, depth, kinv_host9, cam_to_cam0, N, H, W)) return rc;
  ICV_REQUIRE(out_f32 || out_u8, "icv_coord_normalize: no output");
  ICV_REQUIRE(!has_valid || (mins_host3 && ranges_host3), "icv_coord_normalize: mins/ranges required");
  const float z[3] = {0.f, 0.f, 0.f}, o[3] = {1.f, 1.f, 1.f};
  const float* mn = has_valid ? mins_host3 : z;
  const float* rg = has_valid ? ranges_host3 : o;
  const int64_t total = N * H * W;
  hipLaunchKernelGGL(coord_normalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p,
                     mn[0], mn[1], mn[2], rg[0], rg[1], rg[2], has_valid, out_f32, out_u8, total);
  return icv_check_launch("icv_coord_normalize");
}

// ---------------------------------------------------------------------------------------------
// semantic / instance colour buffer — two LUT passes over the pixels.
//   semantic_to_color [R infinicube/utils/semantic_utils.py:88-101]: colour = PALETTE[MAPPING[class]]
//   generate_rgb_semantic_buffer [R infinicube/utils/semantic_utils.py:104-131]: instance colour where
//   instance > 0 (table built on the host from the reference's random colormap samples), else semantic.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void semantic_color_kernel(const int* __restrict__ sem, int64_t n,
                                                             const float* __restrict__ lut, int n_classes,
                                                             float* __restrict__ out_f32,
                                                             unsigned char* __restrict__ out_u8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int c = sem[i];
  c = c < 0 ? 0 : (c >= n_classes ? n_classes - 1 : c);
  const float r = lut[c * 3 + 0], g = lut[c * 3 + 1], b = lut[c * 3 + 2];
  if (out_f32) { out_f32[i * 3 + 0] = r; out_f32[i * 3 + 1] = g; out_f32[i * 3 + 2] = b; }
  if (out_u8) {   // the caller's (colour * 255).astype(uint8)
    out_u8[i * 3 + 0] = (unsigned char)(int)(r * 255.0f);
    out_u8[i * 3 + 1] = (unsigned char)(int)(g * 255.0f);
    out_u8[i * 3 + 2] = (unsigned char)(int)(b * 255.0f);
  }
}

__global__ __launch_bounds__(256) void instance_overlay_kernel(const unsigned char* __restrict__ sem_rgb,
                                                               const int* __restrict__ inst, int64_t n,
                                                               const unsigned char* __restrict__ inst_lut,
                                                               unsigned char* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int id = inst[i] & 0xffff;
  const unsigned char* src = id > 0 ? inst_lut + (int64_t)id * 3 : sem_rgb + i * 3;
  out[i * 3 + 0] = src[0]; out[i * 3 + 1] = src[1]; out[i * 3 + 2] = src[2];
}

extern "C" int icv_semantic_to_color(const int* semantics, int64_t n, const float* class_rgb_lut, int n_classes,
                                     float* out_f32, unsigned char* out_u8, void* stream) {
  ICV_REQUIRE(semantics && class_rgb_lut && n > 0 && n_classes > 0 && (out_f32 || out_u8), "icv_semantic_to_color: bad arguments");
  hipLaunchKernelGGL(semantic_color_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     semantics, n, class_rgb_lut, n_classes, out_f32, out_u8);
  return icv_check_launch("icv_semantic_to_color");
}

extern "C" int icv_instance_overlay_u8(const unsigned char* semantics_rgb, const int* instance, int64_t n,
                                       const unsigned char* instance_rgb_lut65536, unsigned char* out, void* stream) {
  ICV_REQUIRE(semantics_rgb && instance && instance_rgb_lut65536 && out && n > 0, "icv_instance_overlay_u8: bad arguments");
  hipLaunchKernelGGL(instance_overlay_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     semantics_rgb, instance, n, instance_rgb_lut65536, out);
  return icv_check_launch("icv_instance_overlay_u8");
}

// ---------------------------------------------------------------------------------------------
// the depth wire format of stage 2, `voxel_depth_100_*.tar` members =
// `(depth * 100).astype(np.uint16)` [R infinicube/inference/guidance_buffer_generation.py:668-670]: one rounded f32
// multiply, truncation toward zero, wrap modulo 2^16 (numpy's float -> uint16 cast goes through a 64-bit integer on
// x86-64).  HBM-bound: 4 B read + 2 B written per pixel, so only the 2-byte image crosses PCIe for PNG encoding.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void depth_to_u16_kernel(const float* __restrict__ depth, unsigned short* __restrict__ out,
                                                           float scale, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 d = *reinterpret_cast<const float4*>(depth + i);
    ushort4 o;
    o.x = (unsigned short)(long long)__fmul_rn(d.x, scale); o.y = (unsigned short)(long long)__fmul_rn(d.y, scale);
    o.z = (unsigned short)(long long)__fmul_rn(d.z, scale); o.w = (unsigned short)(long long)__fmul_rn(d.w, scale);
    *reinterpret_cast<ushort4*>(out + i) = o;
  } else {
    for (int64_t j = i; j < n; ++j) out[j] = (unsigned short)(long long)__fmul_rn(depth[j], scale);
  }
}

extern "C" int icv_depth_to_u16(const float* depth, int64_t n, float scale, unsigned short* out, void* stream) {
  ICV_REQUIRE(depth && out && n > 0, "icv_depth_to_u16: bad arguments");
  ICV_REQUIRE(((uintptr_t)depth % 16 == 0) && ((uintptr_t)out % 8 == 0), "icv_depth_to_u16: buffers must be 16-byte / 8-byte aligned");
  const int64_t threads = (n + 3) / 4;
  hipLaunchKernelGGL(depth_to_u16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, depth, out, scale, n);
  return icv_check_launch("icv_depth_to_u16");
}
