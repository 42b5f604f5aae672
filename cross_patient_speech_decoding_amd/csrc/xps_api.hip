// Error reporting and ABI version of libxps.so.
#include <stdarg.h>
#include "xps_common.h"

static thread_local char g_err[512] = "";

void xps_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* xps_last_error(void) { return g_err; }
extern "C" int xps_abi_version(void) { return 1; }
