// Host-side plumbing of libm3vit_hip.so: error text, version, device query.
#include <stdarg.h>

#include "common.h"

namespace m3 {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return M3_ERR_LAUNCH;
  }
  return M3_OK;
}

}  // namespace m3

extern "C" int m3_version(void) { return 100; }   // 0.1.0

extern "C" const char *m3_last_error(void) { return m3::g_err; }

// 1 when the library was built with `make EXPERIMENTAL=1`: the opt-in kernels the engine never takes (fused FFN, weight-stationary
// GEMM, wide weight-gradient tiles) are then compiled in; the default build carries only what the training step runs
extern "C" int m3_experimental(void) {
#ifdef M3_EXPERIMENTAL
  return 1;
#else
  return 0;
#endif
}

#ifndef M3_EXPERIMENTAL
extern "C" int m3_ffn_fwd(const m3_ffn_args *, void *) {
  m3::set_error("m3_ffn_fwd: the fused FFN kernel is only in EXPERIMENTAL builds (make EXPERIMENTAL=1); the default path is "
                "two m3_gemm_nt launches");
  return M3_ERR_ARG;
}
#endif

extern "C" int m3_device_query(char *name, int len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { m3::set_error("m3_device_query: no HIP device"); return M3_ERR_LAUNCH; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { m3::set_error("m3_device_query: hipGetDeviceProperties failed"); return M3_ERR_LAUNCH; }
  if (name && len > 0) {
    strncpy(name, prop.gcnArchName, (size_t)len - 1);
    name[len - 1] = 0;
  }
  return prop.multiProcessorCount;
}
