// Error plumbing and version for the C ABI (include/alg_hip.h).
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace alg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return ALG_ELAUNCH;
  }
  return ALG_OK;
}

}  // namespace alg

extern "C" int alg_version(void) { return ALG_VERSION; }
extern "C" const char* alg_last_error(void) { return alg::g_err; }
