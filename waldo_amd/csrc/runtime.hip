// Error reporting and version of the C ABI (include/waldo_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "waldo_common.hip.h"

namespace waldo {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return WALDO_ELAUNCH;
  }
  return WALDO_OK;
}

}  // namespace waldo

extern "C" int waldo_version(void) { return 1000; }

extern "C" const char* waldo_last_error_string(void) { return waldo::g_err; }
