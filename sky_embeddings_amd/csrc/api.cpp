// Error reporting + version for the libskyemb C ABI.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/skyemb.h"

static thread_local char g_err[512] = "";

void skyemb_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *skyemb_last_error(void) { return g_err; }
extern "C" int skyemb_version(void) { return 100; }
