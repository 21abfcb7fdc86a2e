// Error reporting, version and the test-only debug options of the C ABI (include/waldo_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include <atomic>

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

static std::atomic<int> g_debug[WALDO_DEBUG_COUNT];

bool debug_option(int option) {
  return option >= 0 && option < WALDO_DEBUG_COUNT && g_debug[option].load(std::memory_order_relaxed) != 0;
}

}  // namespace waldo

// A timing-only ablation build (waldo_common.hip.h) reports version 0: no binding written against a real ABI
// version accepts it by accident.
#ifdef WALDO_TIMING_ONLY_BUILD
extern "C" int waldo_version(void) { return 0; }
#else
extern "C" int waldo_version(void) { return 1011; }
#endif

extern "C" int waldo_set_debug_option(int option, int value) {
  if (option < 0 || option >= WALDO_DEBUG_COUNT) {
    waldo::set_error("waldo_set_debug_option: unknown option %d", option);
    return WALDO_EINVAL;
  }
  waldo::g_debug[option].store(value, std::memory_order_relaxed);
  return WALDO_OK;
}

extern "C" const char* waldo_last_error_string(void) { return waldo::g_err; }
