// Error reporting + version for libstudiosr_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include "sr_host.h"

static thread_local char g_err[512] = "";

void sr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int sr_abi_version(void) { return SR_ABI_VERSION; }
extern "C" const char* sr_last_error(void) { return g_err; }
