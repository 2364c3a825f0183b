// Error reporting and ABI version for libanemoi_hip.so.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace anemoi {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(err));
    return ANEMOI_E_LAUNCH;
  }
  return ANEMOI_OK;
}

}  // namespace anemoi

extern "C" int anemoi_hip_abi_version(void) { return ANEMOI_HIP_ABI_VERSION; }
extern "C" const char* anemoi_hip_last_error(void) { return anemoi::g_err; }
