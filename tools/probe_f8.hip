// Probe of the gfx950 MX-scaled fp8 MFMAs used by gemm_fp8.hip: checks (1) the builtins' operand order,
// (2) that "lane group g = lane / rows holds K bytes [32g, 32g+32) of its row" is a consistent A/B
// K-mapping, (3) that scale operands 0 select the unscaled form and that an E8M0 scale byte of 127
// (2^0) is the identity while 128 doubles.   hipcc --offload-arch=gfx950 -O2 tools/probe_f8.hip -o probe_f8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// C[m][n] = sum_k A[m][k] * B[n][k];  A [16][128], B [16][128] fp8 e4m3 bytes
__global__ void k16(const uint8_t* A, const uint8_t* B, float* C, int mode) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  v8i a = *reinterpret_cast<const v8i*>(A + r * 128 + g * 32);
  v8i b = *reinterpret_cast<const v8i*>(B + r * 128 + g * 32);
  v4f c = {0, 0, 0, 0};
  if (mode == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
  if (mode == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  if (mode == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f80, 0, 0x7f7f7f7f);
  // result layout: first operand indexes the 4-consecutive "row" (= (lane>>4)*4 + i), second the column lane&15
  for (int i = 0; i < 4; ++i) C[(g * 4 + i) * 16 + r] = c[i];
}
// 32x32x64: A [32][64], B [32][64]
__global__ void k32(const uint8_t* A, const uint8_t* B, float* C) {
  const int lane = threadIdx.x, r = lane & 31, g = lane >> 5;
  v8i a = *reinterpret_cast<const v8i*>(A + r * 64 + g * 32);
  v8i b = *reinterpret_cast<const v8i*>(B + r * 64 + g * 32);
  v16f c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * g) * 32 + r] = c[i];
}
static const uint8_t enc[9] = {0xC8, 0xC4, 0xC0, 0xB8, 0x00, 0x38, 0x40, 0x44, 0x48};  // -4..4 in e4m3
int main() {
  srand(1);
  int bad = 0;
  {
    std::vector<uint8_t> A(16 * 128), B(16 * 128);
    std::vector<int> Ai(16 * 128), Bi(16 * 128);
    for (int i = 0; i < 16 * 128; ++i) { Ai[i] = rand() % 9 - 4; Bi[i] = rand() % 9 - 4; A[i] = enc[Ai[i] + 4]; B[i] = enc[Bi[i] + 4]; }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
      hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, mode);
      std::vector<float> C(256);
      hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
      int nb = 0, nbT = 0;
      for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        int ref = 0; for (int k = 0; k < 128; ++k) ref += Ai[m * 128 + k] * Bi[n * 128 + k];
        float want = (float)ref * (mode == 2 ? 2.f : 1.f);
        nb += C[m * 16 + n] != want; nbT += C[n * 16 + m] != want;
      }
      printf("16x16x128 mode %d: C[a_row][b_row] mismatches %d, transposed reading %d\n", mode, nb, nbT);
      bad += (nb != 0);
    }
  }
  {
    std::vector<uint8_t> A(32 * 64), B(32 * 64);
    std::vector<int> Ai(32 * 64), Bi(32 * 64);
    for (int i = 0; i < 32 * 64; ++i) { Ai[i] = rand() % 9 - 4; Bi[i] = rand() % 9 - 4; A[i] = enc[Ai[i] + 4]; B[i] = enc[Bi[i] + 4]; }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 1024 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> C(1024);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int nb = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
      int ref = 0; for (int k = 0; k < 64; ++k) ref += Ai[m * 64 + k] * Bi[n * 64 + k];
      nb += C[m * 32 + n] != (float)ref;
    }
    printf("32x32x64: mismatches %d\n", nb);
    bad += (nb != 0);
  }
  printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
  return bad;
}
