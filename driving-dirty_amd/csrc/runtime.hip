// Error reporting of the C ABI (thread-local message; see include/dd_hotpath.h).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "dd_common.h"

static thread_local char g_err[512] = "";

int dd_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static int g_cu_budget = DD_NUM_CU;

int dd_cu_budget_internal() { return g_cu_budget; }

int dd_mfma_wave_priority() {      // DD_MFMA_PRIO=0..3 (A/B knob; results unchanged)
  static const int prio = getenv("DD_MFMA_PRIO") ? max(0, min(3, atoi(getenv("DD_MFMA_PRIO")))) : DD_MFMA_PRIO_DEFAULT;
  return prio;
}

extern "C" {
int dd_abi_version(void) { return DD_ABI_VERSION; }

int dd_set_cu_budget(int32_t compute_units) {
  DD_REQUIRE(compute_units >= 1 && compute_units <= DD_NUM_CU, DD_ERR_BAD_ARG, "set_cu_budget: %d not in 1..%d", compute_units, DD_NUM_CU);
  g_cu_budget = compute_units;
  return 0;
}
int dd_get_cu_budget(void) { return g_cu_budget; }
const char* dd_last_error(void) { return g_err; }
}
