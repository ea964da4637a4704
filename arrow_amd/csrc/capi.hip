// Error plumbing and misc entry points of the C ABI (include/arrow_amd.h).
#include "arx_common.h"

#include <stdarg.h>
#include <stdio.h>

namespace arx {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", static_cast<int>(e), hipGetErrorString(e), what);
  (void)hipGetLastError();  // clear the sticky error
  return e == hipErrorOutOfMemory ? ARX_OUT_OF_MEMORY : ARX_DEVICE_ERROR;
}

}  // namespace arx

extern "C" {

const char* arx_last_error(void) { return arx::g_error; }

int arx_abi_version(void) { return ARX_ABI_VERSION; }

int arx_set_option(const char* name, int64_t value) {
  if (name == nullptr) {
    arx::set_error("option name is NULL");
    return ARX_INVALID;
  }
  if (arx::set_selection_option(name, value) || arx::set_sort_option(name, value) ||
      arx::set_groupby_option(name, value) || arx::set_parquet_option(name, value)) {
    return ARX_OK;
  }
  arx::set_error("unknown option '%s'", name);
  return ARX_INVALID;
}

int64_t arx_get_counter(const char* name) {
  int64_t v = 0;
  if (name != nullptr && (arx::get_groupby_counter(name, &v) || arx::get_sort_counter(name, &v))) return v;
  arx::set_error("unknown counter '%s'", name == nullptr ? "(null)" : name);
  return -1;
}

int arx_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return arx::hip_fail(e, "hipGetDeviceCount");
  return n;
}

}  // extern "C"
