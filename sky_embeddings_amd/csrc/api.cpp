// Error reporting + version for the libskyemb C ABI.
#include <stdarg.h>
#include <stdio.h>

#include <atomic>

#include "../../include/skyemb.h"

static thread_local char g_err[512] = "";

void skyemb_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *skyemb_last_error(void) { return g_err; }
// 100: rounds 1-4.  110 (round 6): skyemb_gemm_args lost colsum_parts and gained prefetch / prefetch_bytes / prefetch_wgs in round 5
// (sizeof 240 -> 248, group blob + 128 B, skyemb_gemm_group_ws_bytes removed) without a bump; round 6 adds SKYEMB_F16 and the
// `dscale` argument of the two loss entry points.  Callers built against an older header must be rebuilt (_lib.py asserts it).
extern "C" int skyemb_version(void) { return 110; }

// Measurement aid (bench.py): kernels of the families in `mask` are not launched (their entry points return 0), so that
// a timed region with and without them gives that family's in-step time.  bit 0: MFMA GEMM launches (single, grouped,
// split-K reduce).  Exists in libskyemb_measure.so only (-DSKYEMB_MEASURE: csrc/Makefile builds it beside the product library,
// bench.py runs its skip leg on it in a process of its own); in the product library the call fails and nothing is ever skipped.
#ifdef SKYEMB_MEASURE
static int g_skip_mask = 0;
extern "C" int skyemb_debug_skip(int mask) {
    const int old = g_skip_mask;
    g_skip_mask = mask;
    return old;
}
int skyemb_skip_mask(void) { return g_skip_mask; }
#else
extern "C" int skyemb_debug_skip(int mask) {
    (void)mask;
    skyemb_set_error("skyemb_debug_skip: this is the product build; the measurement switch exists in libskyemb_measure.so only");
    return -1;
}
#endif

// Diagnostic: how many launches each GEMM kernel family has issued in this process (relaxed counters; tests assert through
// them that a configuration really ran on the kernels it is meant to exercise).  Slots: SKYEMB_GEMM_COUNT_* of skyemb.h.
static std::atomic<long long> g_gemm_counts[SKYEMB_GEMM_COUNT_SLOTS];
void skyemb_count_gemm(int slot) {
    if (slot >= 0 && slot < SKYEMB_GEMM_COUNT_SLOTS) g_gemm_counts[slot].fetch_add(1, std::memory_order_relaxed);
}
extern "C" int skyemb_gemm_launch_counts(long long *out, int n, int reset) {
    if (!out || n < 0) {
        skyemb_set_error("skyemb_gemm_launch_counts: bad arguments");
        return 1;
    }
    for (int i = 0; i < n && i < SKYEMB_GEMM_COUNT_SLOTS; ++i)
        out[i] = reset ? g_gemm_counts[i].exchange(0, std::memory_order_relaxed) : g_gemm_counts[i].load(std::memory_order_relaxed);
    for (int i = SKYEMB_GEMM_COUNT_SLOTS; i < n; ++i) out[i] = 0;
    return 0;
}
