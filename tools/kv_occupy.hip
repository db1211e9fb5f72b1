// Measurement helper for tools/kv_contention.py: a stand-in for what a kernel-based K|V transport does to
// the GPU while attention runs — `k` work-groups that copy bytes in a loop until the host tells them to stop, so the attention
// launch under test runs its WHOLE duration beside exactly k resident copy work-groups.  Two footprints:
//   lds_bytes = 0      : 256 threads, a handful of VGPRs, no LDS — the lightest possible channel
//   lds_bytes = 65536  : the same loop holding 64 KiB of LDS — a channel that cannot share a CU with an attention work-group
//                        (attn7 holds 128 KiB of the CU's 160 KiB), i.e. one that takes its CU away for as long as it runs.
// Not part of libicvideo.so; built by tools/kv_contention.py with hipcc into tools/libkvoccupy.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
volatile int* g_stop = nullptr;          // host-pinned, device-visible

__global__ __launch_bounds__(256) void occupy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t vec_per_block,
                                                     volatile int* stop, unsigned long long* copied, int64_t max_rounds, int use_lds) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) {           // where did this work-group land: copied[1 + 2b] = XCC_ID, copied[2 + 2b] = HW_ID (b < 64)
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (blockIdx.x < 64) { copied[1 + 2 * blockIdx.x] = xcc; copied[2 + 2 * blockIdx.x] = hwid; }
  }
  if (use_lds && threadIdx.x == 0) lds[0] = 0;          // keep the dynamic LDS allocation alive
  const uint4* s = src + (int64_t)blockIdx.x * vec_per_block;
  uint4* d = dst + (int64_t)blockIdx.x * vec_per_block;
  unsigned long long rounds = 0;
  for (int64_t r = 0; r < max_rounds; ++r) {
    for (int64_t i = threadIdx.x; i < vec_per_block; i += 4 * 256) {      // 4 loads in flight per lane, like a channel's unrolled copy
      uint4 a = s[i], b = uint4{}, c = uint4{}, e = uint4{};
      if (i + 256 < vec_per_block) b = s[i + 256];
      if (i + 512 < vec_per_block) c = s[i + 512];
      if (i + 768 < vec_per_block) e = s[i + 768];
      d[i] = a;
      if (i + 256 < vec_per_block) d[i + 256] = b;
      if (i + 512 < vec_per_block) d[i + 512] = c;
      if (i + 768 < vec_per_block) d[i + 768] = e;
    }
    ++rounds;
    if (*stop) break;
  }
  if (threadIdx.x == 0) atomicAdd(copied, rounds * (unsigned long long)vec_per_block * 16ull);
}
}  // namespace

extern "C" int occ_init() {
  if (g_stop) return 0;
  int* p = nullptr;
  if (hipHostMalloc((void**)&p, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return 1;
  *p = 0;
  g_stop = p;
  return 0;
}
// k work-groups, each copying its own `bytes_per_block` slice of src -> dst over and over until occ_stop(); `copied` (device u64)
// accumulates the bytes moved.  max_rounds bounds the run if the host never stops it.
extern "C" int occ_start(int k, const void* src, void* dst, int64_t bytes_per_block, int lds_bytes, void* copied, int64_t max_rounds, void* stream) {
  if (!g_stop || k <= 0) return 1;
  *g_stop = 0;
  if (lds_bytes > 0 && hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 2;
  hipLaunchKernelGGL(occupy_kernel, dim3(k), dim3(256), lds_bytes, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, bytes_per_block / 16, g_stop,
                     (unsigned long long*)copied, max_rounds, lds_bytes > 0 ? 1 : 0);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
extern "C" void occ_stop() { if (g_stop) *g_stop = 1; }

// A stream whose kernels may only run on the CUs whose bit is set in `mask` (hipExtStreamCreateWithCUMask) - tools/cu_mask_probe.py
extern "C" int occ_masked_stream(const unsigned* mask, int words, void** stream) {
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) != hipSuccess) return 1;
  *stream = s;
  return 0;
}
extern "C" void occ_stream_destroy(void* s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }

// ---- probe: a stream wait WITHOUT a wave - hipLaunchHostFunc whose host function polls a host flag (tools/probe_hostfunc.py) ----
#include <atomic>
#include <chrono>
#include <thread>
namespace {
struct HostWait { volatile uint32_t* flag; uint32_t value; std::atomic<int>* entered; };
void host_wait_fn(void* arg) {
  HostWait* w = (HostWait*)arg;
  if (w->entered) w->entered->fetch_add(1);
  const auto t0 = std::chrono::steady_clock::now();
  while ((int32_t)(*w->flag - w->value) < 0) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) break;      // never hang the runtime's thread for good
    std::this_thread::yield();
  }
  delete w;
}
}  // namespace
// `stream` blocks (no wave) until *flag >= value: the host function runs on a runtime thread once the stream's earlier work is done
extern "C" int occ_host_wait(void* stream, volatile uint32_t* flag, uint32_t value, void* entered_counter) {
  HostWait* w = new HostWait{flag, value, (std::atomic<int>*)entered_counter};
  return hipLaunchHostFunc((hipStream_t)stream, host_wait_fn, w) == hipSuccess ? 0 : 1;
}
