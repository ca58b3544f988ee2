// Error plumbing and build identification for libmacr_hip.so.
#include "common.hpp"

#include <string.h>
#define __MACR_STR2(x) #x
#define __MACR_STR(x) __MACR_STR2(x)

namespace macr {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- optional per-kernel timing -------------------------------------------------
struct Mark { char name[32]; hipEvent_t ev; };
static thread_local bool g_timing = false;
static thread_local Mark g_marks[256];
static thread_local int g_nmarks = 0;

void timing_mark(const char *name, hipStream_t st) {
    if (!g_timing || g_nmarks >= 256) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;       // events cannot be timed inside a graph capture
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
    Mark &m = g_marks[g_nmarks];
    if (hipEventCreate(&m.ev) != hipSuccess) return;
    strncpy(m.name, name, sizeof(m.name) - 1);
    m.name[sizeof(m.name) - 1] = 0;
    (void)hipEventRecord(m.ev, st);
    ++g_nmarks;
}
}  // namespace macr

extern "C" int macr_timing_begin(void *stream) {
    using namespace macr;
    for (int k = 0; k < g_nmarks; ++k) (void)hipEventDestroy(g_marks[k].ev);
    g_nmarks = 0;
    g_timing = true;
    timing_mark("begin", as_stream(stream));
    return MACR_OK;
}

extern "C" int macr_timing_end(int max_n, char *names /* [max_n][32] */, float *ms /* [max_n] */) {
    using namespace macr;
    g_timing = false;
    if (g_nmarks == 0) return 0;
    (void)hipEventSynchronize(g_marks[g_nmarks - 1].ev);
    int n = 0;
    for (int k = 1; k < g_nmarks && n < max_n; ++k, ++n) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, g_marks[k - 1].ev, g_marks[k].ev);
        ms[n] = t;
        memcpy(names + (size_t)n * 32, g_marks[k].name, 32);
    }
    for (int k = 0; k < g_nmarks; ++k) (void)hipEventDestroy(g_marks[k].ev);
    g_nmarks = 0;
    return n;
}

extern "C" int macr_abi_version(void) { return MACR_ABI_VERSION; }
extern "C" const char *macr_last_error(void) { return macr::g_err; }
extern "C" const char *macr_build_info(void) {
    return "libmacr_hip gfx950 (MI355X) hip " __MACR_STR(HIP_VERSION_MAJOR) "." __MACR_STR(HIP_VERSION_MINOR)
           " abi " __MACR_STR(MACR_ABI_VERSION);
}
