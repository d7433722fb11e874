// libegtr_hip.so: status / version plumbing of the C ABI declared in include/egtr_hip.h.
#include "common.h"

namespace {
thread_local hipError_t g_last = hipSuccess;
}

int egtr_check_launch() {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last = e;
    return EGTR_E_LAUNCH;
  }
  return EGTR_OK;
}

int egtr_raise_dynamic_lds(const void* kernel, int bytes, unsigned long long* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return egtr_check_launch();
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(done, __ATOMIC_ACQUIRE) & bit) return EGTR_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return egtr_check_launch();
  __atomic_fetch_or(done, bit, __ATOMIC_RELEASE);
  return EGTR_OK;
}

extern "C" int egtr_abi_version(void) { return EGTR_ABI_VERSION; }

extern "C" const char* egtr_status_string(int status) {
  switch (status) {
    case EGTR_OK: return "ok";
    case EGTR_E_ARG: return "invalid argument (null pointer, non-positive size or misaligned buffer)";
    case EGTR_E_LAUNCH: return "HIP kernel launch failed";
    case EGTR_E_UNSUPPORTED: return "shape / dtype not supported by this entry point";
    default: return "unknown status";
  }
}

extern "C" const char* egtr_last_hip_error(void) { return hipGetErrorString(g_last); }
