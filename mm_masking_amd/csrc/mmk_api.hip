// Version + per-thread error string of the C ABI (include/mmk.h).
#include <stdarg.h>
#include <stdio.h>

#include "mmk_common.h"

namespace {
thread_local char g_err[512] = "";
}

namespace mmk {
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mmk

extern "C" int mmk_version(void) { return MMK_VERSION; }
extern "C" const char *mmk_last_error(void) { return g_err; }
