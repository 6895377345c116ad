// 2x2 max-pool (ignore_border) and the materialised equality-mask unpool, float32 and float64.
// HBM-bound element-wise kernels: one output element per thread-iteration, lanes along x.
// Replaces Pool2DLayer (reference models/fcn8.py:38-72, models/fcn_down.py:122) and
// DePool2D.get_output_for (layers/mylayers.py:88-115); see include/iiseg.h.
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const T* __restrict__ x,
                                                         T* __restrict__ out, int BC, int H,
                                                         int W, int h, int w) {
    const size_t n = (size_t)BC * h * w;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % w);
        const size_t t = i / w;
        const int oy = (int)(t % h);
        const size_t bc = t / h;
        const T* r0 = x + (bc * H + 2 * oy) * (size_t)W + 2 * ox;
        const T* r1 = r0 + W;
        const T a = r0[0] > r0[1] ? r0[0] : r0[1];
        const T b = r1[0] > r1[1] ? r1[0] : r1[1];
        out[i] = a > b ? a : b;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void unpool_eqmask_kernel(const T* __restrict__ up,
                                                            const T* __restrict__ pre,
                                                            const T* __restrict__ pooled,
                                                            T* __restrict__ out, int BC, int H,
                                                            int W, int h, int w) {
    const size_t n = (size_t)BC * H * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const size_t t = i / W;
        const int y = (int)(t % H);
        const size_t bc = t / H;
        T v = 0;
        if (y < 2 * h && x < 2 * w) {
            const size_t j = (bc * h + (y >> 1)) * (size_t)w + (x >> 1);
            v = (pre[i] == pooled[j]) ? up[j] : (T)0;
        }
        out[i] = v;
    }
}

// window form: only pooled outputs [y0,y0+wh) x [x0,x0+ww) of every (h,w) plane, in place
template <typename T>
__global__ __launch_bounds__(256) void maxpool2x2_window_kernel(const T* __restrict__ x,
                                                                T* __restrict__ out, int BC, int H,
                                                                int W, int h, int w, int y0, int x0,
                                                                int wh, int ww) {
    const size_t n = (size_t)BC * wh * ww;
    if (n < ((size_t)1 << 31)) {
        // (32-bit index arithmetic: the 64-bit divisions below are a few dozen instructions each, and the windows
        // of a refinement step are small -- this kernel is launched 40+ times per batch of the float64 path)
        const unsigned n32 = (unsigned)n, uww = (unsigned)ww, uwh = (unsigned)wh;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += gridDim.x * blockDim.x) {
            const unsigned t = i / uww;
            const int ox = x0 + (int)(i - t * uww);
            const unsigned bc = t / uwh;
            const int oy = y0 + (int)(t - bc * uwh);
            const T* r0 = x + ((size_t)bc * H + 2 * oy) * (size_t)W + 2 * ox;
            const T* r1 = r0 + W;
            const T a = r0[0] > r0[1] ? r0[0] : r0[1];
            const T b = r1[0] > r1[1] ? r1[0] : r1[1];
            out[((size_t)bc * h + oy) * (size_t)w + ox] = a > b ? a : b;
        }
        return;
    }
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ox = x0 + (int)(i % ww);
        const size_t t = i / ww;
        const int oy = y0 + (int)(t % wh);
        const size_t bc = t / wh;
        const T* r0 = x + (bc * H + 2 * oy) * (size_t)W + 2 * ox;
        const T* r1 = r0 + W;
        const T a = r0[0] > r0[1] ? r0[0] : r0[1];
        const T b = r1[0] > r1[1] ? r1[0] : r1[1];
        out[(bc * h + oy) * (size_t)w + ox] = a > b ? a : b;
    }
}

// window form: only [y0,y0+wh) x [x0,x0+ww) of every (H,W) plane is produced, in place
template <typename T>
__global__ __launch_bounds__(256) void unpool_eqmask_window_kernel(
    const T* __restrict__ up, const T* __restrict__ pre, const T* __restrict__ pooled,
    T* __restrict__ out, int BC, int H, int W, int h, int w, int y0, int x0, int wh, int ww) {
    const size_t n = (size_t)BC * wh * ww;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const int x = x0 + (int)(i % ww);
        const size_t t = i / ww;
        const int y = y0 + (int)(t % wh);
        const size_t bc = t / wh;
        const size_t o = (bc * H + y) * (size_t)W + x;
        T v = 0;
        if (y < 2 * h && x < 2 * w) {
            const size_t j = (bc * h + (y >> 1)) * (size_t)w + (x >> 1);
            v = (pre[o] == pooled[j]) ? up[j] : (T)0;
        }
        out[o] = v;
    }
}

// ---- backward kernels of the true-gradient mode ----------------------------
// DePool2D backward: the unpooled tensor is mask * repeat(up), so
//   g_up[i,j] = sum over the 2x2 window of (pre == pooled[i,j]) ? g_out : 0   (no gradient through
// the mask: it is piecewise constant).
template <typename T>
__global__ __launch_bounds__(256) void depool_bwd_kernel(const T* __restrict__ gout,
                                                         const T* __restrict__ pre,
                                                         const T* __restrict__ pooled,
                                                         T* __restrict__ gup, int BC, int H, int W,
                                                         int h, int w) {
    const size_t n = (size_t)BC * h * w;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % w);
        const size_t t = i / w;
        const int oy = (int)(t % h);
        const size_t bc = t / h;
        const size_t o = (bc * H + 2 * oy) * (size_t)W + 2 * ox;
        const T m = pooled[i];
        T s = 0;
        s += pre[o] == m ? gout[o] : (T)0;
        s += pre[o + 1] == m ? gout[o + 1] : (T)0;
        s += pre[o + W] == m ? gout[o + W] : (T)0;
        s += pre[o + W + 1] == m ? gout[o + W + 1] : (T)0;
        gup[i] = s;
    }
}

// max-pool backward (every position equal to the window maximum receives the gradient, the
// equality mask of SURVEY F4) fused with the backward of the ReLU in front of the pool
// (relu'(0) = 0):  g_z[y,x] = (pre == pooled[y/2,x/2] && pre > 0) ? g_pool[y/2,x/2] : 0.
template <typename T>
__global__ __launch_bounds__(256) void pool_relu_bwd_kernel(const T* __restrict__ gpool,
                                                            const T* __restrict__ pre,
                                                            const T* __restrict__ pooled,
                                                            T* __restrict__ gz, int BC, int H, int W,
                                                            int h, int w) {
    const size_t n = (size_t)BC * H * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const size_t t = i / W;
        const int y = (int)(t % H);
        const size_t bc = t / H;
        T v = 0;
        if (y < 2 * h && x < 2 * w) {
            const size_t j = (bc * h + (y >> 1)) * (size_t)w + (x >> 1);
            const T pv = pre[i];
            v = (pv == pooled[j] && pv > (T)0) ? gpool[j] : (T)0;
        }
        gz[i] = v;
    }
}

inline int grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

template <typename T>
int maxpool(void* stream, const T* x, T* out, int32_t BC, int32_t H, int32_t W) {
    if (!x || !out) return IISEG_ERR_NULL;
    if (BC <= 0 || H < 2 || W < 2) return IISEG_ERR_SHAPE;
    const int h = H / 2, w = W / 2;
    IISEG_LAUNCH(maxpool2x2_kernel<T>, dim3(grid_for((size_t)BC * h * w)), dim3(256), 0,
                       (hipStream_t)stream, x, out, BC, H, W, h, w);
    return iiseg_check_launch();
}

template <typename T>
int unpool(void* stream, const T* up, const T* pre, const T* pooled, T* out, int32_t BC, int32_t H,
           int32_t W) {
    if (!up || !pre || !pooled || !out) return IISEG_ERR_NULL;
    if (BC <= 0 || H < 2 || W < 2) return IISEG_ERR_SHAPE;
    IISEG_LAUNCH(unpool_eqmask_kernel<T>, dim3(grid_for((size_t)BC * H * W)), dim3(256), 0,
                       (hipStream_t)stream, up, pre, pooled, out, BC, H, W, H / 2, W / 2);
    return iiseg_check_launch();
}

template <typename T>
int maxpool_window(void* stream, const T* x, T* out, int32_t BC, int32_t H, int32_t W, int32_t y0,
                   int32_t x0, int32_t wh, int32_t ww) {
    if (!x || !out) return IISEG_ERR_NULL;
    if (BC <= 0 || H < 2 || W < 2 || y0 < 0 || x0 < 0 || wh <= 0 || ww <= 0 || y0 + wh > H / 2 ||
        x0 + ww > W / 2)
        return IISEG_ERR_SHAPE;
    IISEG_LAUNCH(maxpool2x2_window_kernel<T>, dim3(grid_for((size_t)BC * wh * ww)), dim3(256),
                       0, (hipStream_t)stream, x, out, BC, H, W, H / 2, W / 2, y0, x0, wh, ww);
    return iiseg_check_launch();
}

template <typename T>
int unpool_window(void* stream, const T* up, const T* pre, const T* pooled, T* out, int32_t BC,
                  int32_t H, int32_t W, int32_t y0, int32_t x0, int32_t wh, int32_t ww) {
    if (!up || !pre || !pooled || !out) return IISEG_ERR_NULL;
    if (BC <= 0 || H < 2 || W < 2 || y0 < 0 || x0 < 0 || wh <= 0 || ww <= 0 || y0 + wh > H ||
        x0 + ww > W)
        return IISEG_ERR_SHAPE;
    IISEG_LAUNCH(unpool_eqmask_window_kernel<T>, dim3(grid_for((size_t)BC * wh * ww)),
                       dim3(256), 0, (hipStream_t)stream, up, pre, pooled, out, BC, H, W, H / 2,
                       W / 2, y0, x0, wh, ww);
    return iiseg_check_launch();
}

template <typename T>
int depool_bwd(void* stream, const T* gout, const T* pre, const T* pooled, T* gup, int32_t BC,
               int32_t H, int32_t W) {
    if (!gout || !pre || !pooled || !gup) return IISEG_ERR_NULL;
    if (BC <= 0 || H < 2 || W < 2) return IISEG_ERR_SHAPE;
    IISEG_LAUNCH(depool_bwd_kernel<T>, dim3(grid_for((size_t)BC * (H / 2) * (W / 2))), dim3(256),
                       0, (hipStream_t)stream, gout, pre, pooled, gup, BC, H, W, H / 2, W / 2);
    return iiseg_check_launch();
}

template <typename T>
int pool_relu_bwd(void* stream, const T* gpool, const T* pre, const T* pooled, T* gz, int32_t BC,
                  int32_t H, int32_t W) {
    if (!gpool || !pre || !pooled || !gz) return IISEG_ERR_NULL;
    if (BC <= 0 || H < 2 || W < 2) return IISEG_ERR_SHAPE;
    IISEG_LAUNCH(pool_relu_bwd_kernel<T>, dim3(grid_for((size_t)BC * H * W)), dim3(256), 0,
                       (hipStream_t)stream, gpool, pre, pooled, gz, BC, H, W, H / 2, W / 2);
    return iiseg_check_launch();
}

}  // namespace

extern "C" int iiseg_depool_bwd_f32(void* stream, const float* gout, const float* pre,
                                    const float* pooled, float* gup, int32_t BC, int32_t H, int32_t W) {
    return depool_bwd<float>(stream, gout, pre, pooled, gup, BC, H, W);
}
extern "C" int iiseg_depool_bwd_f64(void* stream, const double* gout, const double* pre,
                                    const double* pooled, double* gup, int32_t BC, int32_t H,
                                    int32_t W) {
    return depool_bwd<double>(stream, gout, pre, pooled, gup, BC, H, W);
}
extern "C" int iiseg_pool_relu_bwd_f32(void* stream, const float* gpool, const float* pre,
                                       const float* pooled, float* gz, int32_t BC, int32_t H,
                                       int32_t W) {
    return pool_relu_bwd<float>(stream, gpool, pre, pooled, gz, BC, H, W);
}
extern "C" int iiseg_pool_relu_bwd_f64(void* stream, const double* gpool, const double* pre,
                                       const double* pooled, double* gz, int32_t BC, int32_t H,
                                       int32_t W) {
    return pool_relu_bwd<double>(stream, gpool, pre, pooled, gz, BC, H, W);
}

extern "C" int iiseg_maxpool2x2_window_f32(void* stream, const float* x, float* out, int32_t BC,
                                           int32_t H, int32_t W, int32_t y0, int32_t x0,
                                           int32_t wh, int32_t ww) {
    return maxpool_window<float>(stream, x, out, BC, H, W, y0, x0, wh, ww);
}
extern "C" int iiseg_maxpool2x2_window_f64(void* stream, const double* x, double* out, int32_t BC,
                                           int32_t H, int32_t W, int32_t y0, int32_t x0,
                                           int32_t wh, int32_t ww) {
    return maxpool_window<double>(stream, x, out, BC, H, W, y0, x0, wh, ww);
}
extern "C" int iiseg_unpool_eqmask_window_f32(void* stream, const float* up, const float* pre,
                                              const float* pooled, float* out, int32_t BC,
                                              int32_t H, int32_t W, int32_t y0, int32_t x0,
                                              int32_t wh, int32_t ww) {
    return unpool_window<float>(stream, up, pre, pooled, out, BC, H, W, y0, x0, wh, ww);
}
extern "C" int iiseg_unpool_eqmask_window_f64(void* stream, const double* up, const double* pre,
                                              const double* pooled, double* out, int32_t BC,
                                              int32_t H, int32_t W, int32_t y0, int32_t x0,
                                              int32_t wh, int32_t ww) {
    return unpool_window<double>(stream, up, pre, pooled, out, BC, H, W, y0, x0, wh, ww);
}

extern "C" int iiseg_maxpool2x2_f32(void* stream, const float* x, float* out, int32_t BC,
                                    int32_t H, int32_t W) {
    return maxpool<float>(stream, x, out, BC, H, W);
}
extern "C" int iiseg_maxpool2x2_f64(void* stream, const double* x, double* out, int32_t BC,
                                    int32_t H, int32_t W) {
    return maxpool<double>(stream, x, out, BC, H, W);
}
extern "C" int iiseg_unpool_eqmask_f32(void* stream, const float* up, const float* pre,
                                       const float* pooled, float* out, int32_t BC, int32_t H,
                                       int32_t W) {
    return unpool<float>(stream, up, pre, pooled, out, BC, H, W);
}
extern "C" int iiseg_unpool_eqmask_f64(void* stream, const double* up, const double* pre,
                                       const double* pooled, double* out, int32_t BC, int32_t H,
                                       int32_t W) {
    return unpool<double>(stream, up, pre, pooled, out, BC, H, W);
}
