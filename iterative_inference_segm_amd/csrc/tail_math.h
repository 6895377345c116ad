// Arithmetic of the refinement tail for ONE pixel (channel softmax + the update of iterative_inference.py:
// 203-204, 270-277), shared by every kernel that runs it (tail.hip: refine_update_kernel; conv_small.hip: the
// context module's fused last layers) so that a pixel gets the same bits whichever kernel computes it.
#pragma once
#include <hip/hip_runtime.h>

namespace {

__device__ inline float exp_t(float x) { return expf(x); }
__device__ inline double exp_t(double x) { return exp(x); }
__device__ inline float sqrt_t(float x) { return sqrtf(x); }
__device__ inline double sqrt_t(double x) { return sqrt(x); }

// r <- softmax over the first C entries (same order of operations as ever: max, exp(v - max), sum in
// channel order, one reciprocal); entries >= C become 0
template <int CMAX, typename T>
__device__ inline void softmax_column(int C, T (&r)[CMAX]) {
    T m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (c < C) m = r[c] > m ? r[c] : m;
    T s = 0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (c < C) {
            r[c] = exp_t(r[c] - m);
            s += r[c];
        } else {
            r[c] = 0;
        }
    const T inv = (T)1 / s;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (c < C) r[c] *= inv;
}

// r: the pixel's scores (becomes softmax(score)); yv: its y column, updated in place when `act`;
// returns sum_c de_c^2 with de = y - r (before the update)
template <int CMAX, typename T>
__device__ __forceinline__ T refine_pixel(int C, T (&r)[CMAX], T (&yv)[CMAX], bool act, T step) {
    softmax_column<CMAX, T>(C, r);
    T ss = 0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (c < C) {
            const T de = yv[c] - r[c];  // iterative_inference.py:203-204
            ss = fma(de, de, ss);
            if (act) {
                T yn = yv[c] - step * de;  // :270
                yv[c] = yn < (T)0 ? (T)0 : (yn > (T)1 ? (T)1 : yn);  // :273
            }
        }
    return ss;
}

}  // namespace
