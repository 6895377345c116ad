// Status strings / version of the C ABI (include/iiseg.h); launch profiling state.
#include <mutex>
#include <vector>
#include "common.h"

namespace {
std::mutex g_prof_mu;
std::vector<hipEvent_t> g_prof_ev;      // pairs (start, stop), created once and reused
int g_prof_on = 0, g_prof_n = 0, g_prof_cap = 0;
}  // namespace

extern "C" int iiseg_prof_next(hipEvent_t* start, hipEvent_t* stop) {
    if (!g_prof_on) return 0;
    std::lock_guard<std::mutex> g(g_prof_mu);
    if (!g_prof_on || g_prof_n >= g_prof_cap) return 0;
    *start = g_prof_ev[2 * g_prof_n];
    *stop = g_prof_ev[2 * g_prof_n + 1];
    ++g_prof_n;
    return 1;
}

extern "C" int iiseg_profile_begin(int capacity) {
    if (capacity <= 0) return IISEG_ERR_SHAPE;
    std::lock_guard<std::mutex> g(g_prof_mu);
    while ((int)g_prof_ev.size() < 2 * capacity) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return IISEG_ERR_LAUNCH;
        g_prof_ev.push_back(e);
    }
    g_prof_cap = capacity;
    g_prof_n = 0;
    g_prof_on = 1;
    return IISEG_OK;
}

extern "C" int iiseg_profile_count(void) { return g_prof_on ? g_prof_n : -1; }

extern "C" int iiseg_profile_end(float* ms, int capacity) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    if (!g_prof_on) return -1;
    g_prof_on = 0;
    const int n = g_prof_n < capacity ? g_prof_n : capacity;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(g_prof_ev[2 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms[i], g_prof_ev[2 * i], g_prof_ev[2 * i + 1]) != hipSuccess)
            return IISEG_ERR_LAUNCH;
    }
    return n;
}

extern "C" const char* iiseg_strerror(int status) {
    switch (status) {
        case IISEG_OK: return "ok";
        case IISEG_ERR_NULL: return "required pointer is NULL";
        case IISEG_ERR_SHAPE: return "inconsistent or unsupported shape";
        case IISEG_ERR_ALIGN: return "pointer is not 16-byte aligned";
        case IISEG_ERR_LAUNCH: return "kernel launch failed";
        case IISEG_ERR_UNSUPPORTED: return "no kernel variant implements this request";
        default: return "unknown iiseg status";
    }
}

extern "C" const char* iiseg_last_hip_error(void) {
    return hipGetErrorString((hipError_t)iiseg_hip_error_slot());
}

extern "C" int iiseg_abi_version(void) { return 31; }
extern "C" const char* iiseg_target_arch(void) { return "gfx950"; }
