// Batch-statistics BatchNorm + ReLU for the FC-DenseNet host network (float32 / float64).
// Replaces lasagne BatchNormLayer with batch_norm_use_averages=False (reference
// iterative_inference.py:187; SURVEY P10: batch mean, biased variance over (B,H,W), eps 1e-4) and
// the rectify that follows it in BN_ReLU_Conv (FC_DenseNet.layers, models/FCDenseNet.py:12).
// stats: 64 workgroups per channel + a finalize pass, fp64 accumulation, fixed reduction order
// (deterministic);
// apply: HBM-bound element-wise kernel.  A channel's statistics never change once the channel is
// in the stack, so the host computes them once per produced tensor, not once per consumer.
#include "common.h"

namespace {

// stats, stage 1: block (chunk, channel) reduces its slice of the B*HW elements of one channel to
// (sum, sum of squares) in fp64; stage 2: one thread per channel adds the NCHUNK partials in a
// fixed order (deterministic, no atomics) and writes mean / inv_std.  (A single workgroup per
// channel left 240 of the 256 CUs idle for the 16-channel tensors of FC-DenseNet.)
constexpr int BN_NCHUNK = 64;

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, int64_t bstride,
                                                       int B, int HW, double* __restrict__ partial) {
    __shared__ double red[2][4];
    const int c = blockIdx.y, chunk = blockIdx.x;
    const size_t n = (size_t)B * HW;
    const size_t per = (n + BN_NCHUNK - 1) / BN_NCHUNK;
    const size_t lo = (size_t)chunk * per, hi = lo + per < n ? lo + per : n;
    double s = 0.0, ss = 0.0;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        const size_t b = i / HW, r = i - b * HW;
        const double v = (double)x[b * bstride + (size_t)c * HW + r];
        s += v;
        ss += v * v;
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s;
        red[1][threadIdx.x >> 6] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((size_t)c * BN_NCHUNK + chunk) * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[((size_t)c * BN_NCHUNK + chunk) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

template <typename T>
__global__ void bn_stats_finalize_kernel(const double* __restrict__ partial, int C, double n,
                                         double eps, T* __restrict__ mean, T* __restrict__ inv_std) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < BN_NCHUNK; ++k) {
        s += partial[((size_t)c * BN_NCHUNK + k) * 2 + 0];
        ss += partial[((size_t)c * BN_NCHUNK + k) * 2 + 1];
    }
    const double m = s / n;
    double var = ss / n - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (T)m;
    inv_std[c] = (T)(1.0 / sqrt(var + eps));
}

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_kernel(const T* __restrict__ x, int64_t bstride,
                                                      int C, int HW, const T* __restrict__ beta,
                                                      const T* __restrict__ gamma,
                                                      const T* __restrict__ mean,
                                                      const T* __restrict__ inv_std,
                                                      T* __restrict__ out) {
    const int c = blockIdx.y, b = blockIdx.z;
    const T m = mean[c], g = gamma[c] * inv_std[c], be = beta[c];
    const T* xp = x + (size_t)b * bstride + (size_t)c * HW;
    T* op = out + ((size_t)b * C + c) * HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const T v = (xp[i] - m) * g + be;   // (input - mean) * (gamma * inv_std) + beta
        op[i] = v > (T)0 ? v : (T)0;
    }
}

// stored-average BatchNorm (deterministic=True with running averages: the DAE's bn=1 layers,
// reference iterative_inference.py:189, models/fcn_down.py:112-114), in place on a window of the
// (H, W) planes: x = (x - mean) * (gamma * inv_std) + beta
template <typename T>
__global__ __launch_bounds__(256) void bn_affine_window_kernel(T* __restrict__ x, int C, int H, int W,
                                                               int y0, int x0, int wh, int ww,
                                                               const T* __restrict__ beta,
                                                               const T* __restrict__ gamma,
                                                               const T* __restrict__ mean,
                                                               const T* __restrict__ inv_std) {
    const int c = blockIdx.y, b = blockIdx.z;
    const T m = mean[c], g = gamma[c] * inv_std[c], be = beta[c];
    T* xp = x + ((size_t)b * C + c) * H * W;
    const int n = wh * ww;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int yy = i / ww, xx = i - yy * ww;
        T* e = xp + (size_t)(y0 + yy) * W + x0 + xx;
        *e = (*e - m) * g + be;
    }
}

template <typename T>
int bn_affine_window(void* stream, T* x, int32_t B, int32_t C, int32_t H, int32_t W, int32_t y0,
                     int32_t x0, int32_t wh, int32_t ww, const T* beta, const T* gamma, const T* mean,
                     const T* inv_std) {
    if (!x || !beta || !gamma || !mean || !inv_std) return IISEG_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || y0 < 0 || x0 < 0 || wh <= 0 || ww <= 0 ||
        y0 + wh > H || x0 + ww > W)
        return IISEG_ERR_SHAPE;
    if (C > 65535 || B > 65535) return IISEG_ERR_UNSUPPORTED;
    int gx = (wh * ww + 255) / 256;
    if (gx > 64) gx = 64;
    IISEG_LAUNCH(bn_affine_window_kernel<T>, dim3(gx, C, B), dim3(256), 0, (hipStream_t)stream,
                       x, C, H, W, y0, x0, wh, ww, beta, gamma, mean, inv_std);
    return iiseg_check_launch();
}

template <typename T>
int bn_stats(void* stream, const T* x, int64_t bstride, int32_t B, int32_t C, int32_t HW, double eps,
             T* mean, T* inv_std, double* workspace) {
    if (!x || !mean || !inv_std || !workspace) return IISEG_ERR_NULL;
    if (B <= 0 || C <= 0 || HW <= 0 || bstride < (int64_t)C * HW) return IISEG_ERR_SHAPE;
    if (C > 65535) return IISEG_ERR_UNSUPPORTED;
    IISEG_LAUNCH(bn_stats_kernel<T>, dim3(BN_NCHUNK, C), dim3(256), 0, (hipStream_t)stream, x,
                       bstride, B, HW, workspace);
    IISEG_LAUNCH(bn_stats_finalize_kernel<T>, dim3((C + 63) / 64), dim3(64), 0,
                       (hipStream_t)stream, workspace, C, (double)B * HW, eps, mean, inv_std);
    return iiseg_check_launch();
}

template <typename T>
int bn_relu(void* stream, const T* x, int64_t bstride, int32_t B, int32_t C, int32_t HW,
            const T* beta, const T* gamma, const T* mean, const T* inv_std, T* out) {
    if (!x || !beta || !gamma || !mean || !inv_std || !out) return IISEG_ERR_NULL;
    if (B <= 0 || C <= 0 || HW <= 0 || bstride < (int64_t)C * HW) return IISEG_ERR_SHAPE;
    if (C > 65535 || B > 65535) return IISEG_ERR_UNSUPPORTED;
    int gx = (HW + 255) / 256;
    if (gx > 64) gx = 64;
    IISEG_LAUNCH(bn_relu_kernel<T>, dim3(gx, C, B), dim3(256), 0, (hipStream_t)stream, x,
                       bstride, C, HW, beta, gamma, mean, inv_std, out);
    return iiseg_check_launch();
}

}  // namespace

extern "C" int64_t iiseg_bn_stats_workspace_elems(int32_t C) { return (int64_t)C * BN_NCHUNK * 2; }
extern "C" int iiseg_bn_stats_f32(void* stream, const float* x, int64_t bstride, int32_t B,
                                  int32_t C, int32_t HW, float eps, float* mean, float* inv_std,
                                  double* workspace) {
    return bn_stats<float>(stream, x, bstride, B, C, HW, (double)eps, mean, inv_std, workspace);
}
extern "C" int iiseg_bn_stats_f64(void* stream, const double* x, int64_t bstride, int32_t B,
                                  int32_t C, int32_t HW, double eps, double* mean, double* inv_std,
                                  double* workspace) {
    return bn_stats<double>(stream, x, bstride, B, C, HW, eps, mean, inv_std, workspace);
}
extern "C" int iiseg_bn_relu_f32(void* stream, const float* x, int64_t bstride, int32_t B, int32_t C,
                                 int32_t HW, const float* beta, const float* gamma,
                                 const float* mean, const float* inv_std, float* out) {
    return bn_relu<float>(stream, x, bstride, B, C, HW, beta, gamma, mean, inv_std, out);
}
extern "C" int iiseg_bn_relu_f64(void* stream, const double* x, int64_t bstride, int32_t B, int32_t C,
                                 int32_t HW, const double* beta, const double* gamma,
                                 const double* mean, const double* inv_std, double* out) {
    return bn_relu<double>(stream, x, bstride, B, C, HW, beta, gamma, mean, inv_std, out);
}
extern "C" int iiseg_bn_affine_window_f32(void* stream, float* x, int32_t B, int32_t C, int32_t H,
                                          int32_t W, int32_t y0, int32_t x0, int32_t wh, int32_t ww,
                                          const float* beta, const float* gamma, const float* mean,
                                          const float* inv_std) {
    return bn_affine_window<float>(stream, x, B, C, H, W, y0, x0, wh, ww, beta, gamma, mean, inv_std);
}
extern "C" int iiseg_bn_affine_window_f64(void* stream, double* x, int32_t B, int32_t C, int32_t H,
                                          int32_t W, int32_t y0, int32_t x0, int32_t wh, int32_t ww,
                                          const double* beta, const double* gamma,
                                          const double* mean, const double* inv_std) {
    return bn_affine_window<double>(stream, x, B, C, H, W, y0, x0, wh, ww, beta, gamma, mean, inv_std);
}
