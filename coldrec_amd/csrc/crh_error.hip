// Error reporting and version for the C ABI (include/coldrec_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "crh_common.h"

static thread_local char g_err[512] = "";

void crh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* crh_last_error(void) { return g_err; }
extern "C" int crh_version(void) { return 100; }
