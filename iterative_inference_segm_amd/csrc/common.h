// Shared helpers of libiiseg_hip.so (gfx950 only; no CUDA/other-backend paths).
#pragma once
#include <hip/hip_runtime.h>
#include "iiseg.h"

static inline int iiseg_check_launch() {
    return hipGetLastError() == hipSuccess ? IISEG_OK : IISEG_ERR_LAUNCH;
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
