// Library-wide entry points: version and the last-error text (no exceptions cross the C ABI).
#include <stdarg.h>
#include <stdio.h>

#include "rn_hip.h"

namespace rn {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace rn

extern "C" int rn_version(void) { return RN_API_VERSION; }
extern "C" const char* rn_last_error(void) { return rn::g_err; }
