// Library-wide C ABI plumbing: version, thread-local error message.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void gct_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int gct_version(void) { return GCT_ABI_VERSION; }
extern "C" const char* gct_last_error(void) { return g_err; }
