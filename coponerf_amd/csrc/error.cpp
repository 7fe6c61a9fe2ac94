// Error string + version of the C ABI (include/coponerf_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/coponerf_hip.h"

static thread_local char g_err[512] = "";

void cpn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int cpn_abi_version(void) { return CPN_ABI_VERSION; }
extern "C" const char* cpn_last_error(void) { return g_err; }
