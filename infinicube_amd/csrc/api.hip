// libicvideo: error reporting, version, device probing (plain C ABI, see include/icvideo.h).
#include <stdarg.h>
#include <stdio.h>

#include "icv_common.h"

static thread_local char g_err[512] = "";

void icv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int icv_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    icv_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}

int icv_ensure_dynamic_lds(const void* func, int bytes, icv_dev_flags* flags, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  const bool tracked = dev >= 0 && dev < ICV_MAX_DEVICES;
  if (tracked && flags->set[dev]) return 0;
  hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    icv_set_error("%s: hipFuncSetAttribute(%d bytes of LDS) failed on device %d: %s", what, bytes, dev, hipGetErrorString(e));
    return 2;
  }
  if (tracked) flags->set[dev] = true;   // benign race: two threads of one device may both set the attribute once
  return 0;
}

// ---- runtime options (A/B switches for kernel variants; defaults are the shipped configuration) ----
#include <map>
#include <mutex>
#include <string>
static std::map<std::string, int> g_opts;
static std::mutex g_opts_mu;
int icv_get_option_int(const char* name, int dflt) {
  std::lock_guard<std::mutex> lk(g_opts_mu);
  auto it = g_opts.find(name);
  return it == g_opts.end() ? dflt : it->second;
}
extern "C" int icv_set_option(const char* name, int value) {
  // "require_experiments": succeeds only in a library built with ICV_EXPERIMENTS=1 (the A/B kernels under experiments/)
  if (name && std::string(name) == "require_experiments") {
#ifdef ICV_EXPERIMENTS
    return 0;
#else
    icv_set_error("this libicvideo was built without ICV_EXPERIMENTS=1");
    return 1;
#endif
  }
  if (!name) {
    icv_set_error("icv_set_option: null name");
    return 1;
  }
  std::lock_guard<std::mutex> lk(g_opts_mu);
  g_opts[name] = value;
  return 0;
}

extern "C" int icv_abi_version(void) { return ICV_ABI_VERSION; }
extern "C" const char* icv_last_error(void) { return g_err; }

extern "C" int icv_device_info(int device, int64_t out[4]) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    icv_set_error("icv_device_info: %s", hipGetErrorString(e));
    return 2;
  }
  out[0] = prop.multiProcessorCount;
  out[1] = (int64_t)prop.sharedMemPerBlock;
  int arch = 0;
  sscanf(prop.gcnArchName, "gfx%d", &arch);
  out[2] = arch;
  out[3] = prop.warpSize;
  return 0;
}
