// SURVEY.md §8f row 4 (second half): the buffer renderer's voxel ray-cast — what stage 2 does through fVDB before the hot
// path [R infinicube/utils/fvdb_utils.py:572-605; infinicube/camera/base.py:520-619]:
//     depth    = distance to the first RUN of occupied voxels along the pixel ray that is longer than eps_depth
//                (`segments_along_rays(o, d, 1, eps=1e-1)`), as z-depth (x the camera ray's z);
//     semantic = attribute of the first occupied VOXEL the ray crosses for more than eps_voxel
//     instance   (`voxels_along_rays(o, d, 1, eps=1e-2)` + gather), background value where nothing is hit.
// fVDB (sparse VDB tree + HDDA on CUDA) is an absent dependency, so this is MI355X-first rather than a port: 288 GB of
// HBM make a DENSE int32 index volume over the scene's bounding box affordable (400 m x 400 m x 40 m at 0.2 m = 3.2 GB),
// which turns the tree descent into one 4-byte load per visited voxel; an 8^3-brick occupancy byte map lets a ray cross
// empty space without touching the volume.  One thread per ray (latency-bound random reads: many rays in flight);
// rays of a wave are neighbouring pixels, so they walk neighbouring cells and share cache lines.
//
// Traversal = 3-D DDA over cells.  Every boundary time is a PURE FUNCTION of (axis, cell): t = (face - o) * inv_d, never an
// accumulated sum, so the brick skip (which jumps to the cell the cell-by-cell walk would have reached, found with the
// same comparisons) cannot change a single float; the CPU oracle (oracle/voxel_ref.py) is the plain cell-by-cell walk.
#include "icv_common.h"

namespace {

struct RayParams {
  const int* vol;               // [Dz][Dy][Dx] voxel index or -1
  const unsigned char* bricks;  // [Dz/8][Dy/8][Dx/8] 1 = some voxel of the brick is occupied
  int Dx, Dy, Dz;
  float gx, gy, gz;             // world coordinate of the low corner of cell (0,0,0)
  float ivx, ivy, ivz;          // 1 / voxel size
  const float* rays_cam;        // [H*W, 3] normalised camera rays (the caller's camera_model.get_rays())
  const float* poses;           // [N, 16] camera-to-world, row-major
  int64_t N, HW;
  float eps_depth, eps_voxel;
  const int* attr0; const int* attr1;   // per-voxel attributes (may be NULL)
  int bg0, bg1;
  float* depth; int* out0; int* out1; int* out_idx;   // [N, HW]; any may be NULL
};

__device__ __forceinline__ float face_time(float o, float inv, int step, int c) {
  // time at which the ray leaves cell c along this axis (+inf if it never does)
  return step == 0 ? __builtin_inff() : ((float)(step > 0 ? c + 1 : c) - o) * inv;
}

__global__ __launch_bounds__(256) void voxel_raycast_kernel(RayParams p) {
  const int64_t ray = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (ray >= p.N * p.HW) return;
  const int64_t n = ray / p.HW, pix = ray - n * p.HW;
  const float* m = p.poses + n * 16;
  const float rx = p.rays_cam[pix * 3 + 0], ry = p.rays_cam[pix * 3 + 1], rz = p.rays_cam[pix * 3 + 2];
  // world direction = R r, every product and sum rounded on its own, left to right (the oracle mirrors this)
  float d[3], o[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) d[i] = __fadd_rn(__fadd_rn(__fmul_rn(m[i * 4 + 0], rx), __fmul_rn(m[i * 4 + 1], ry)), __fmul_rn(m[i * 4 + 2], rz));
  // grid coordinates: cell units per axis; t stays the world distance because d is the (unit) world direction
  const float g[3] = {p.gx, p.gy, p.gz}, iv[3] = {p.ivx, p.ivy, p.ivz};
  const int D[3] = {p.Dx, p.Dy, p.Dz};
  float inv[3];
  int step[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o[i] = __fmul_rn(__fsub_rn(m[i * 4 + 3], g[i]), iv[i]);
    const float dg = __fmul_rn(d[i], iv[i]);
    step[i] = dg > 0.f ? 1 : (dg < 0.f ? -1 : 0);
    inv[i] = step[i] ? __fdiv_rn(1.0f, dg) : 0.f;
    d[i] = dg;
  }
  // clip to the volume: t in [t0, t1]
  float t0 = 0.f, t1 = __builtin_inff();
  bool miss = false;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (step[i] == 0) {
      if (o[i] < 0.f || o[i] >= (float)D[i]) miss = true;
    } else {
      const float ta = (0.f - o[i]) * inv[i], tb = ((float)D[i] - o[i]) * inv[i];
      t0 = fmaxf(t0, fminf(ta, tb));
      t1 = fminf(t1, fmaxf(ta, tb));
    }
  }
  float depth = 0.f;
  int hit = -1;
  if (!miss && t0 < t1) {
    int c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int ci = (int)floorf(__fadd_rn(o[i], __fmul_rn(t0, d[i])));
      c[i] = ci < 0 ? 0 : (ci >= D[i] ? D[i] - 1 : ci);
    }
    const int Bx = p.Dx >> 3, By = p.Dy >> 3;
    float t_cur = t0, run_start = 0.f;
    bool in_run = false, depth_done = false, hit_done = false;
    for (int guard = 0; guard < (1 << 20); ++guard) {
      if (c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= D[0] || c[1] >= D[1] || c[2] >= D[2]) break;
      const int64_t bi = ((int64_t)(c[2] >> 3) * By + (c[1] >> 3)) * Bx + (c[0] >> 3);
      if (!p.bricks[bi]) {
        // an empty 8^3 brick: any open run ended when this cell was entered
        if (in_run) {
          in_run = false;
          if (!depth_done && t_cur - run_start > p.eps_depth) { depth = run_start; depth_done = true; }
        }
        if (depth_done && hit_done) break;
        // leave the brick through the face reached first (ties: lowest axis, like the cell walk)
        float tb[3];
        int bb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          bb[i] = step[i] > 0 ? (((c[i] >> 3) + 1) << 3) : ((c[i] >> 3) << 3);
          tb[i] = step[i] == 0 ? __builtin_inff() : ((float)bb[i] - o[i]) * inv[i];
        }
        int a = 0;
        if (tb[1] < tb[a]) a = 1;
        if (tb[2] < tb[a]) a = 2;
        const float ts = tb[a];
        // the other axes: advance while the cell walk would have crossed that face before this one
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (i == a || step[i] == 0) continue;
          for (int k = 0; k < 8; ++k) {
            const float tf = face_time(o[i], inv[i], step[i], c[i]);
            if (tf < ts || (tf == ts && i < a)) c[i] += step[i]; else break;
          }
        }
        c[a] = step[a] > 0 ? bb[a] : bb[a] - 1;
        t_cur = ts;
        continue;
      }
      const int idx = p.vol[((int64_t)c[2] * p.Dy + c[1]) * p.Dx + c[0]];
      const float tx = face_time(o[0], inv[0], step[0], c[0]);
      const float ty = face_time(o[1], inv[1], step[1], c[1]);
      const float tz = face_time(o[2], inv[2], step[2], c[2]);
      int a = 0;
      float t_out = tx;
      if (ty < t_out) { a = 1; t_out = ty; }
      if (tz < t_out) { a = 2; t_out = tz; }
      if (idx >= 0) {
        if (!hit_done && t_out - t_cur > p.eps_voxel) { hit = idx; hit_done = true; }
        if (!in_run) { in_run = true; run_start = t_cur; }
      } else if (in_run) {
        in_run = false;
        if (!depth_done && t_cur - run_start > p.eps_depth) { depth = run_start; depth_done = true; }
      }
      if (depth_done && hit_done) break;
      c[a] += step[a];
      t_cur = t_out;
    }
    if (in_run && !depth_done && t_cur - run_start > p.eps_depth) depth = run_start;   // the run reaches the volume's edge
  }
  if (p.depth) p.depth[ray] = __fmul_rn(depth, rz);      // distance -> z-depth (0 = nothing hit)
  if (p.out_idx) p.out_idx[ray] = hit;
  if (p.out0) p.out0[ray] = hit >= 0 && p.attr0 ? p.attr0[hit] : p.bg0;
  if (p.out1) p.out1[ray] = hit >= 0 && p.attr1 ? p.attr1[hit] : p.bg1;
}

// occupied voxel list -> dense index volume + brick occupancy (volume pre-filled with -1, bricks with 0)
__global__ __launch_bounds__(256) void voxel_scatter_kernel(const int* __restrict__ ijk, int64_t M, int ox, int oy, int oz,
                                                            int Dx, int Dy, int Dz, int* __restrict__ vol,
                                                            unsigned char* __restrict__ bricks) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const int x = ijk[i * 3 + 0] - ox, y = ijk[i * 3 + 1] - oy, z = ijk[i * 3 + 2] - oz;
  if (x < 0 || y < 0 || z < 0 || x >= Dx || y >= Dy || z >= Dz) return;
  vol[((int64_t)z * Dy + y) * Dx + x] = (int)i;
  bricks[((int64_t)(z >> 3) * (Dy >> 3) + (y >> 3)) * (Dx >> 3) + (x >> 3)] = 1;
}

}  // namespace

extern "C" int icv_voxel_scatter(const int* ijk, int64_t M, const int* vol_min3, const int* dims3, int* vol,
                                 unsigned char* bricks, void* stream) {
  ICV_REQUIRE(ijk && vol_min3 && dims3 && vol && bricks && M > 0, "icv_voxel_scatter: bad arguments");
  ICV_REQUIRE(dims3[0] % 8 == 0 && dims3[1] % 8 == 0 && dims3[2] % 8 == 0, "icv_voxel_scatter: volume dims must be multiples of 8");
  hipLaunchKernelGGL(voxel_scatter_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ijk, M,
                     vol_min3[0], vol_min3[1], vol_min3[2], dims3[0], dims3[1], dims3[2], vol, bricks);
  return icv_check_launch("icv_voxel_scatter");
}

extern "C" int icv_voxel_raycast(const int* vol, const unsigned char* bricks, const int* dims3, const float* grid_lo3,
                                 const float* voxel_size3, const float* rays_cam, const float* poses, int64_t N,
                                 int64_t HW, float eps_depth, float eps_voxel, const int* attr0, const int* attr1,
                                 int background0, int background1, float* depth_out, int* attr0_out, int* attr1_out,
                                 int* index_out, void* stream) {
  ICV_REQUIRE(vol && bricks && dims3 && grid_lo3 && voxel_size3 && rays_cam && poses && N > 0 && HW > 0, "icv_voxel_raycast: bad arguments");
  ICV_REQUIRE(dims3[0] % 8 == 0 && dims3[1] % 8 == 0 && dims3[2] % 8 == 0, "icv_voxel_raycast: volume dims must be multiples of 8");
  ICV_REQUIRE(depth_out || attr0_out || attr1_out || index_out, "icv_voxel_raycast: no output");
  RayParams p;
  p.vol = vol; p.bricks = bricks; p.Dx = dims3[0]; p.Dy = dims3[1]; p.Dz = dims3[2];
  p.gx = grid_lo3[0]; p.gy = grid_lo3[1]; p.gz = grid_lo3[2];
  p.ivx = 1.0f / voxel_size3[0]; p.ivy = 1.0f / voxel_size3[1]; p.ivz = 1.0f / voxel_size3[2];
  p.rays_cam = rays_cam; p.poses = poses; p.N = N; p.HW = HW; p.eps_depth = eps_depth; p.eps_voxel = eps_voxel;
  p.attr0 = attr0; p.attr1 = attr1; p.bg0 = background0; p.bg1 = background1;
  p.depth = depth_out; p.out0 = attr0_out; p.out1 = attr1_out; p.out_idx = index_out;
  const int64_t total = N * HW;
  ICV_REQUIRE((total + 255) / 256 < (1LL << 31), "icv_voxel_raycast: too many rays for one launch");
  hipLaunchKernelGGL(voxel_raycast_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
  return icv_check_launch("icv_voxel_raycast");
}
