// Small-channel transposed convolution in gather form (each thread owns one output pixel and
// all output channels: <=16, or <=32 for the reference's 21-class default of models/fcn8.py:17).  The three FCN-8 upsamplers are 11->11 channels (k4 s2, k4 s2,
// k16 s8): a few MFLOP per image, latency/HBM-bound, so no matrix cores here.
// Replaces Deconv2DLayer = Theano CorrMM_gradInputs (reference models/fcn8.py:90,100,109;
// models/fcn_up.py:41-45), with Lasagne's filter_flip=True spatial flip:
//   out[c, Y, X] = bias[c] + sum_o sum_{a = Y mod s (s) < K} sum_{b = X mod s (s) < K}
//                  x[o, (Y-a)/s, (X-b)/s] * W[o, c, K-1-a, K-1-b]
#include "common.h"

namespace {

constexpr int MAXC_LIMIT = 32;

template <typename T>
struct DeconvParams {
    const T* x;
    const T* w;
    const T* bias;
    const T* add;
    T* out;
    int B, Cin, H, W, Cout, K, s;
    int oy0, ox0, OH, OW;
    int AH, AW, ay0, ax0;
};

template <typename T, int MAXC>
__global__ __launch_bounds__(256) void deconv_gather_kernel(const DeconvParams<T> p) {
    const int OHW = p.OH * p.OW;
    const size_t n = (size_t)p.B * OHW;
    const int KK = p.K * p.K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / OHW);
        const int rem = (int)(i - (size_t)b * OHW);
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        const int Y = p.oy0 + oy, X = p.ox0 + ox;
        T acc[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) acc[c] = (p.bias && c < p.Cout) ? p.bias[c] : (T)0;
        const T* xb = p.x + (size_t)b * p.Cin * p.H * p.W;
        for (int a = Y % p.s; a < p.K; a += p.s) {
            const int iy = (Y - a) / p.s;
            if (Y - a < 0 || iy >= p.H) continue;
            for (int bb = X % p.s; bb < p.K; bb += p.s) {
                const int ix = (X - bb) / p.s;
                if (X - bb < 0 || ix >= p.W) continue;
                const int tap = (p.K - 1 - a) * p.K + (p.K - 1 - bb);
                for (int o = 0; o < p.Cin; ++o) {
                    const T xv = xb[((size_t)o * p.H + iy) * p.W + ix];
                    const T* wr = p.w + (size_t)o * p.Cout * KK + tap;
#pragma unroll
                    for (int c = 0; c < MAXC; ++c)
                        if (c < p.Cout) acc[c] = fma(xv, wr[(size_t)c * KK], acc[c]);
                }
            }
        }
        T* op = p.out + (size_t)b * p.Cout * OHW + rem;
        const T* ap = nullptr;
        size_t AHW = 0;
        if (p.add) {
            AHW = (size_t)p.AH * p.AW;
            ap = p.add + (size_t)b * p.Cout * AHW + (size_t)(p.ay0 + oy) * p.AW + p.ax0 + ox;
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < p.Cout) {
                T v = acc[c];
                if (ap) v += ap[(size_t)c * AHW];
                op[(size_t)c * OHW] = v;
            }
    }
}

template <typename T>
int deconv(void* stream, const iiseg_deconv_desc* d, const T* x, const T* w, const T* bias,
           const T* add, T* out) {
    if (!d || !x || !w || !out) return IISEG_ERR_NULL;
    if (d->B <= 0 || d->Cin <= 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 || d->K <= 0 ||
        d->stride <= 0 || d->OH <= 0 || d->OW <= 0 || d->oy0 < 0 || d->ox0 < 0)
        return IISEG_ERR_SHAPE;
    if (d->Cout > MAXC_LIMIT) return IISEG_ERR_UNSUPPORTED;
    const int fullH = (d->H - 1) * d->stride + d->K, fullW = (d->W - 1) * d->stride + d->K;
    if (d->oy0 + d->OH > fullH || d->ox0 + d->OW > fullW) return IISEG_ERR_SHAPE;
    if (add && (d->ay0 < 0 || d->ax0 < 0 || d->AH < d->ay0 + d->OH || d->AW < d->ax0 + d->OW))
        return IISEG_ERR_SHAPE;
    DeconvParams<T> p;
    p.x = x; p.w = w; p.bias = bias; p.add = add; p.out = out;
    p.B = d->B; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.K = d->K;
    p.s = d->stride; p.oy0 = d->oy0; p.ox0 = d->ox0; p.OH = d->OH; p.OW = d->OW;
    p.AH = d->AH; p.AW = d->AW; p.ay0 = d->ay0; p.ax0 = d->ax0;
    const size_t n = (size_t)d->B * d->OH * d->OW;
    size_t g = (n + 255) / 256;
    if (g > 16384) g = 16384;
    if (d->Cout <= 16)
        IISEG_LAUNCH((deconv_gather_kernel<T, 16>), dim3((int)g), dim3(256), 0,
                           (hipStream_t)stream, p);
    else
        IISEG_LAUNCH((deconv_gather_kernel<T, 32>), dim3((int)g), dim3(256), 0,
                           (hipStream_t)stream, p);
    return iiseg_check_launch();
}

}  // namespace

extern "C" int iiseg_deconv_f32(void* stream, const iiseg_deconv_desc* d, const float* x,
                                const float* w, const float* bias, const float* add, float* out) {
    return deconv<float>(stream, d, x, w, bias, add, out);
}
extern "C" int iiseg_deconv_f64(void* stream, const iiseg_deconv_desc* d, const double* x,
                                const double* w, const double* bias, const double* add,
                                double* out) {
    return deconv<double>(stream, d, x, w, bias, add, out);
}
