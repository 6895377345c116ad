// Shared helpers of libiiseg_hip.so (gfx950 only; no CUDA/other-backend paths).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "iiseg.h"

// The HIP error of the last failed launch of this library (one instance across its translation
// units); iiseg_last_hip_error() names it.
inline int& iiseg_hip_error_slot() {
    static int code = 0;
    return code;
}
static inline int iiseg_check_launch() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return IISEG_OK;
    iiseg_hip_error_slot() = (int)e;
    return IISEG_ERR_LAUNCH;
}

// Launch profiling (include/iiseg.h, iiseg_profile_begin / _end): while it is on, every kernel launch
// of the library carries its own start / stop HIP events on the dispatch packet (hipExtLaunchKernelGGL), so
// the elapsed time of a pair is the kernel's execution time on its stream -- what rocprofv3 reports --
// without the barrier packets that separately recorded events put between two kernels.  Off (always,
// outside bench.py's roofline pass): a plain launch.
extern "C" int iiseg_prof_next(hipEvent_t* start, hipEvent_t* stop);   // 1 + the next pair, or 0 when off
#define IISEG_LAUNCH(kernel, grid, block, shmem, stream, ...)                                              \
    do {                                                                                                   \
        hipEvent_t iiseg_e0_, iiseg_e1_;                                                                   \
        if (iiseg_prof_next(&iiseg_e0_, &iiseg_e1_))                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, iiseg_e0_, iiseg_e1_, 0, __VA_ARGS__); \
        else                                                                                               \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                           \
    } while (0)

// 64-wide wavefront reductions (CDNA: wave = 64 lanes)
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
