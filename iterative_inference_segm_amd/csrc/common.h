// Shared helpers of libiiseg_hip.so (gfx950 only; no CUDA/other-backend paths).
#pragma once
#include <hip/hip_runtime.h>
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
