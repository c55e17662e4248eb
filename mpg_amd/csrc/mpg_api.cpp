// ABI bookkeeping: version + thread-local last-error string.
#include <stdarg.h>
#include <stdio.h>

#include "mpg_common.h"

static thread_local char g_err[512] = "";

void mpg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int mpg_abi_version(void) { return MPG_ABI_VERSION; }
extern "C" const char* mpg_last_error(void) { return g_err; }
