// Error reporting of the C ABI (thread-local message; see include/dd_hotpath.h).
#include <stdarg.h>
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

extern "C" {
int dd_abi_version(void) { return DD_ABI_VERSION; }
const char* dd_last_error(void) { return g_err; }
}
