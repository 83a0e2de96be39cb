// core.hip — error reporting, version and device probe of libblvm_hip.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace blvm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace blvm

extern "C" int blvm_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* blvm_last_error(void) { return blvm::g_err; }

extern "C" int blvm_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return 0;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
