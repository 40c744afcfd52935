// Error plumbing and version for the C ABI (include/alg_hip.h).
#include <stdarg.h>
#include <stdlib.h>
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

// ---- run-time options: one table, read from the environment at load (and on alg_reload_env) ----------------------------
struct OptDef {
  const char* name;
  int def;
  int n_ok;        // number of accepted values (0: any integer)
  int ok[4];
};
static const OptDef g_defs[OPT_COUNT] = {
    {"ALG_ATTN_SPLIT_TAIL", 1, 2, {0, 1}},
    {"ALG_ATTN_PP", 4, 3, {0, 4, 7}},
    {"ALG_ATTN_VARIANT", 33, 2, {1, 33}},
    {"ALG_ATTN128_PIPE", 1, 2, {0, 1}},
    {"ALG_ATTN128_Q64", 1, 4, {0, 1, 2, 3}},
    {"ALG_GEMM_PIPE", 10, 3, {6, 9, 10}},
    {"ALG_LOWPASS_PATH", 0, 0, {}},
};
static std::atomic<int> g_opt[OPT_COUNT];

static void load_options() {
  for (int i = 0; i < OPT_COUNT; ++i) {
    const OptDef& d = g_defs[i];
    int v = d.def;
    const char* e = getenv(d.name);
    if (e && *e) {
      // a whole decimal integer or nothing: "off", "1x", " 2" must not silently parse as a number and flip the option
      char* end = nullptr;
      const long x = strtol(e, &end, 10);
      bool ok = end != e && *end == '\0' && x >= -2147483647L && x <= 2147483647L && (d.n_ok == 0);
      if (end != e && *end == '\0')
        for (int j = 0; j < d.n_ok; ++j) ok |= d.ok[j] == x;
      if (ok) v = (int)x;     // a value this build does not know leaves the default in place
    }
    g_opt[i].store(v, std::memory_order_relaxed);
  }
}
namespace {
struct OptInit {
  OptInit() { load_options(); }
} g_opt_init;
}  // namespace

int opt(Opt o) { return g_opt[o].load(std::memory_order_relaxed); }

}  // namespace alg

extern "C" void alg_reload_env(void) { alg::load_options(); }
extern "C" int alg_version(void) { return ALG_VERSION; }
extern "C" const char* alg_last_error(void) { return alg::g_err; }
