// Error channel + ABI version of libmolly_hip.so (host-only translation unit).
#include <stdarg.h>
#include <stdio.h>

#include "molly_hip.h"

static thread_local char g_err[512] = "";

void molly_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* molly_last_error(void) { return g_err; }
extern "C" int molly_abi_version(void) { return 2; }
