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

// Measurement aid (bench.py): kernels of the families in `mask` are not launched (their entry points return 0), so that
// a timed region with and without them gives that family's in-step time.  bit 0: MFMA GEMM launches (single, grouped,
// split-K reduce).  Never set by the product path; results are garbage while a bit is set.
static int g_skip_mask = 0;
extern "C" int skyemb_debug_skip(int mask) {
    const int old = g_skip_mask;
    g_skip_mask = mask;
    return old;
}
int skyemb_skip_mask(void) { return g_skip_mask; }
