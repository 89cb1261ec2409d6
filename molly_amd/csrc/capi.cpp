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

// ---- the host-sanitizer build's launch recorder (common.h MOLLY_HOST_DRY; compiled into that build only) -----------------------
#if defined(MOLLY_HOST_DRY)
static thread_local long g_dry_launches = 0;
static thread_local int g_dry_failed = 0;
static thread_local char g_dry_last[256] = "";
extern "C" int molly_dry_record(const char* kernel, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
                                unsigned long lds) {
    ++g_dry_launches;
    snprintf(g_dry_last, sizeof(g_dry_last), "%.160s <<<(%u,%u,%u),(%u,%u,%u),%lu>>>", kernel, gx, gy, gz, bx, by, bz, lds);
    const unsigned long threads = (unsigned long)bx * by * bz;
    const bool ok = gx >= 1 && gy >= 1 && gz >= 1 && gx <= 2147483647u && gy <= 65535u && gz <= 65535u && threads >= 1 &&
                    threads <= 1024 && lds <= 163840;
    if (!ok) {
        g_dry_failed = 1;
        molly_set_error("dry launch outside the limits of gfx950: %s", g_dry_last);
    }
    return ok ? 0 : 1;
}
// read-and-clear: MOLLY_LAUNCH_CHECK asks once per entry point
extern "C" int molly_dry_failed(void) { const int f = g_dry_failed; g_dry_failed = 0; return f; }
extern "C" long molly_dry_launch_count(void) { return g_dry_launches; }
extern "C" const char* molly_dry_last_launch(void) { return g_dry_last; }
#endif
