// Hardware probe: dumps the lane mapping of ds_read_b64_tr_b16 and the C/D layout of
// v_mfma_f32_32x32x16_bf16 so kernel-layout assumptions are checked against the real chip.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_tr.hip -o tools/probe_tr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__global__ void probe(unsigned short* out_tr, float* out_c) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  // every lane supplies its own contiguous 8 bytes: element index lane*4
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + lane * 4));
  for (int j = 0; j < 4; ++j) out_tr[lane * 4 + j] = (unsigned short)r[j];
  // MFMA 32x32x16 C/D layout check with an asymmetric product
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    const int k = (lane >> 5) * 8 + e;
    // C[i][j] = A[i][0]*B[0][j] + A[i][1]*B[1][j] = i + 64*j  (asymmetric, exact in bf16)
    a[e] = (__bf16)((k == 0) ? (float)(lane & 31) : (k == 1 ? 64.f : 0.f));
    b[e] = (__bf16)((k == 0) ? 1.f : (k == 1 ? (float)(lane & 31) : 0.f));
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int e = 0; e < 16; ++e) out_c[lane * 16 + e] = c[e];
}

int main() {
  unsigned short* d_tr; float* d_c;
  hipMalloc(&d_tr, 64 * 4 * 2); hipMalloc(&d_c, 64 * 16 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_tr, d_c);
  unsigned short h_tr[256]; float h_c[1024];
  hipMemcpy(h_tr, d_tr, sizeof(h_tr), hipMemcpyDeviceToHost);
  hipMemcpy(h_c, d_c, sizeof(h_c), hipMemcpyDeviceToHost);
  printf("ds_read_b64_tr_b16: lane -> 4 x (source_lane, source_elem)\n");
  int ok_tr = 1;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      const int src = h_tr[l * 4 + j];
      printf(" (%2d,%d)", src / 4, src % 4);
      const int t = l & 15, g = l >> 4;
      const int exp_lane = g * 16 + 4 * j + (t >> 2), exp_elem = t & 3;
      if (src / 4 != exp_lane || src % 4 != exp_elem) ok_tr = 0;
    }
    printf("\n");
  }
  printf("TR_ASSUMPTION %s (lane t elem j <- lane 4j + t/4, elem t%%4 within each 16-lane group)\n", ok_tr ? "HOLDS" : "FAILS");
  // C[i][j] = i + 64*j
  int ok_c = 1;
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 16; ++e) {
      const float v = h_c[l * 16 + e];
      const int j = l & 31, i = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
      if (v != (float)(i + 64 * j)) ok_c = 0;
    }
  printf("MFMA32_CD_ASSUMPTION %s (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))\n", ok_c ? "HOLDS" : "FAILS");
  return 0;
}
