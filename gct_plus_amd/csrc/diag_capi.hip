// Error plumbing of libgctplus_diag.so (the diagnostics library has no dependency on the operator library).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/gctplus_diag.h"

static thread_local char g_diag_err[512] = "";

void gct_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_diag_err, sizeof(g_diag_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* gct_diag_last_error(void) { return g_diag_err; }
