// CPU-only build of the loader's host half for sanitizer runs (g++ -fsanitize=address,undefined; tests/test_loader_cpu.py).
// Not part of libmmk_hip.so: the same source text (mmk_loader_host.inc) with the error plumbing of mmk_api.hip / mmk_common.h
// restated without the HIP runtime.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "mmk.h"

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
extern "C" const char *mmk_last_error(void) { return g_err; }

#define MMK_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            mmk::set_error(__VA_ARGS__);  \
            return MMK_ERR_ARG;           \
        }                                 \
    } while (0)

#include "mmk_loader_host.inc"
