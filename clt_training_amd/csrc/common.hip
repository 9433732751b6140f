// Error text + version for libcltmi355.so.
#include "common.h"

#include <stdarg.h>

namespace clt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace clt

extern "C" int clt_version(void) { return 100; }
extern "C" const char* clt_last_error(void) { return clt::g_err; }
