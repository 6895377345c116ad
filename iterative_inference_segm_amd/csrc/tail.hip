// Channel-softmax tail kernels: crop + softmax (network outputs) and the fused refinement
// step  r = softmax(score); de = y - r; y <- clip(y - step*de, 0, 1); sum ||de||_2.
// HBM-bound: one pixel per thread, lanes along x, the C (<= 32) channel values of a pixel
// live in registers; per-image norms are reduced wave -> block -> fixed-order finalize, so the
// early-stop decision is deterministic (no float atomics).
// Replaces: models/fcn8.py:115-130,187-191 and models/fcn_up.py:104-113,154-169 (crop +
// softmax); iterative_inference.py:203-204,270-277 (update, clip, norm, early stop).
#include "common.h"
#include "column_io.h"
#include "tail_math.h"

namespace {

// the thread's pixel: byte offsets inside image b's score window / inside a (C, H, W) map, or T_OOB
struct TailPix {
    unsigned so, po;
    bool inb;
};
template <typename T>
__device__ __forceinline__ TailPix tail_pixel(int SW, int sy0, int sx0, int H, int W) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    TailPix t;
    t.inb = pix < H * W;
    const int y = pix / W, x = pix - y * W;
    t.so = t.inb ? (unsigned)((sy0 + y) * SW + sx0 + x) * (unsigned)sizeof(T) : T_OOB;
    t.po = t.inb ? (unsigned)pix * (unsigned)sizeof(T) : T_OOB;
    return t;
}

template <int CMAX, typename T>
__global__ __launch_bounds__(256) void crop_softmax_kernel(const T* __restrict__ score,
                                                           const T* __restrict__ minuend,
                                                           T* __restrict__ out, int C, int SH,
                                                           int SW, int sy0, int sx0, int H, int W) {
    const int HW = H * W, b = blockIdx.y;
    const unsigned SB = (unsigned)(SH * SW) * (unsigned)sizeof(T), PB = (unsigned)HW * (unsigned)sizeof(T);
    const TailPix t = tail_pixel<T>(SW, sy0, sx0, H, W);
    const __amdgpu_buffer_rsrc_t rs = t_rsrc(score + (size_t)b * C * SH * SW, (unsigned)C * SB);
    const __amdgpu_buffer_rsrc_t rm = t_rsrc(minuend ? minuend + (size_t)b * C * HW : nullptr, minuend ? (unsigned)C * PB : 0u);
    const __amdgpu_buffer_rsrc_t ro = t_rsrc(out + (size_t)b * C * HW, (unsigned)C * PB);
    T r[CMAX], mv[CMAX];
    load_column<CMAX, T>(rs, t.so, SB, C, r);
    load_column<CMAX, T>(rm, t.po, PB, C, mv);     // (no minuend: an empty descriptor, zeros)
    __builtin_amdgcn_sched_barrier(0);     // (every load of the thread in flight before the first use)
    softmax_column<CMAX, T>(C, r);
    if (minuend) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c) r[c] = mv[c] - r[c];
    }
    store_column<CMAX, T>(ro, t.po, PB, C, r);
}

__device__ __forceinline__ unsigned pack_bf16_t(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

template <int CMAX, typename T>
__global__ __launch_bounds__(256) void refine_update_kernel(const T* __restrict__ score,
                                                            T* __restrict__ yio,
                                                            const int* __restrict__ active,
                                                            double* __restrict__ partial, int C,
                                                            int SH, int SW, int sy0, int sx0, int H,
                                                            int W, T step, uint4* __restrict__ y8,
                                                            int C8n) {
    __shared__ double red[4];
    const int HW = H * W, b = blockIdx.y;
    const unsigned SB = (unsigned)(SH * SW) * (unsigned)sizeof(T), PB = (unsigned)HW * (unsigned)sizeof(T);
    const bool act = active[b] != 0;
    const TailPix t = tail_pixel<T>(SW, sy0, sx0, H, W);
    const __amdgpu_buffer_rsrc_t rs = t_rsrc(score + (size_t)b * C * SH * SW, (unsigned)C * SB);
    const __amdgpu_buffer_rsrc_t ry = t_rsrc(yio + (size_t)b * C * HW, (unsigned)C * PB);
    T r[CMAX], yv[CMAX];
    load_column<CMAX, T>(rs, t.so, SB, C, r);
    load_column<CMAX, T>(ry, t.po, PB, C, yv);
    __builtin_amdgcn_sched_barrier(0);     // (every load of the thread in flight before the first use)
    const T ss = refine_pixel<CMAX, T>(C, r, yv, act, step);     // (tail_math.h)
    store_column<CMAX, T>(ry, act ? t.po : T_OOB, PB, C, yv);
    const T nrm = t.inb ? sqrt_t(ss) : (T)0;  // np.linalg.norm(grad, axis=1), :275
    // bf16 C8 copy of the updated map for the next DAE forward (mma='bf16c8': saves the
    // nchw_to_c8 pass per step); chunk j = channels 8 j .. 8 j + 7, zeros beyond C
    if (y8 && t.inb) {
        const int pix = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
        for (int j = 0; j < CMAX / 8; ++j)
            if (j < C8n) {
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = 8 * j + e < C ? (float)yv[8 * j + e] : 0.f;
                y8[((size_t)b * C8n + j) * HW + pix] =
                    make_uint4(pack_bf16_t(f[0], f[1]), pack_bf16_t(f[2], f[3]), pack_bf16_t(f[4], f[5]),
                               pack_bf16_t(f[6], f[7]));
            }
    }
    double d = wave_sum((double)nrm);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[(size_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- true-gradient mode (not in the reference, F1) ----------------------------
// E(y) = sum_{c,px} (r(y|h) - y)^2.  sqerr_softmax_bwd: g_score = dE/dscore through the softmax,
// g_r = 2 (r - y):  g_score_c = r_c (g_r_c - sum_k r_k g_r_k).
template <int CMAX, typename T>
__global__ __launch_bounds__(256) void sqerr_softmax_bwd_kernel(const T* __restrict__ score,
                                                                const T* __restrict__ yin,
                                                                T* __restrict__ gscore, int C, int SH,
                                                                int SW, int sy0, int sx0, int H, int W) {
    const int HW = H * W, b = blockIdx.y;
    const unsigned SB = (unsigned)(SH * SW) * (unsigned)sizeof(T), PB = (unsigned)HW * (unsigned)sizeof(T);
    const TailPix t = tail_pixel<T>(SW, sy0, sx0, H, W);
    const __amdgpu_buffer_rsrc_t rs = t_rsrc(score + (size_t)b * C * SH * SW, (unsigned)C * SB);
    const __amdgpu_buffer_rsrc_t ry = t_rsrc(yin + (size_t)b * C * HW, (unsigned)C * PB);
    const __amdgpu_buffer_rsrc_t rg = t_rsrc(gscore + (size_t)b * C * HW, (unsigned)C * PB);
    T r[CMAX], g[CMAX];
    load_column<CMAX, T>(rs, t.so, SB, C, r);
    load_column<CMAX, T>(ry, t.po, PB, C, g);
    __builtin_amdgcn_sched_barrier(0);     // (every load of the thread in flight before the first use)
    softmax_column<CMAX, T>(C, r);
    T dot = 0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (c < C) {
            g[c] = (T)2 * (r[c] - g[c]);
            dot = fma(r[c], g[c], dot);
        }
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (c < C) g[c] = r[c] * (g[c] - dot);
    store_column<CMAX, T>(rg, t.po, PB, C, g);
}

// grad = gthrough - 2 (r - y)  (gthrough = the part of dE/dy that flows back through the DAE);
// y <- clip(y - step * grad, 0, 1); per-pixel ||grad||_2 reduced like refine_update_kernel.
template <int CMAX, typename T>
__global__ __launch_bounds__(256) void grad_update_kernel(const T* __restrict__ score,
                                                          const T* __restrict__ gthrough,
                                                          T* __restrict__ yio,
                                                          const int* __restrict__ active,
                                                          double* __restrict__ partial, int C, int SH,
                                                          int SW, int sy0, int sx0, int H, int W,
                                                          T step) {
    __shared__ double red[4];
    const int HW = H * W, b = blockIdx.y;
    const unsigned SB = (unsigned)(SH * SW) * (unsigned)sizeof(T), PB = (unsigned)HW * (unsigned)sizeof(T);
    const bool act = active[b] != 0;
    const TailPix t = tail_pixel<T>(SW, sy0, sx0, H, W);
    const __amdgpu_buffer_rsrc_t rs = t_rsrc(score + (size_t)b * C * SH * SW, (unsigned)C * SB);
    const __amdgpu_buffer_rsrc_t rg = t_rsrc(gthrough + (size_t)b * C * HW, (unsigned)C * PB);
    const __amdgpu_buffer_rsrc_t ry = t_rsrc(yio + (size_t)b * C * HW, (unsigned)C * PB);
    T r[CMAX], yv[CMAX], gt[CMAX];
    load_column<CMAX, T>(rs, t.so, SB, C, r);
    load_column<CMAX, T>(ry, t.po, PB, C, yv);
    load_column<CMAX, T>(rg, t.po, PB, C, gt);
    __builtin_amdgcn_sched_barrier(0);     // (every load of the thread in flight before the first use)
    softmax_column<CMAX, T>(C, r);
    T ss = 0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (c < C) {
            const T g = gt[c] - (T)2 * (r[c] - yv[c]);
            ss = fma(g, g, ss);
            if (act) {
                T yn = yv[c] - step * g;
                yv[c] = yn < (T)0 ? (T)0 : (yn > (T)1 ? (T)1 : yn);
            }
        }
    store_column<CMAX, T>(ry, act ? t.po : T_OOB, PB, C, yv);
    const T nrm = t.inb ? sqrt_t(ss) : (T)0;
    double d = wave_sum((double)nrm);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[(size_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// One wave per image: lane l adds the partials l, l + 64, ... in order, then the 64 lane sums are added in lane
// order by a fixed butterfly -- a fixed association (no float atomics: the early-stop decisions are reproducible),
// without one thread walking all nblk partials of an image with a dependent load each (0.029 -> 0.006 ms).
__global__ __launch_bounds__(64) void refine_finalize_kernel(const double* __restrict__ partial, int* active,
                                                             int* iters, double* last_norm, int B, int nblk,
                                                             int HW, double eps) {
    const int b = blockIdx.x;
    if (b >= B || !active[b]) return;                        // (uniform per wave)
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) s += partial[(size_t)b * nblk + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) {
        const double norm = s / (double)HW;  // .mean() over pixels, :275
        iters[b] += 1;
        last_norm[b] = norm;
        if (norm < eps) active[b] = 0;  // :276-277, after the update was applied
    }
}

template <typename T>
int crop_softmax(void* stream, const T* score, const T* minuend, T* out, int32_t B, int32_t C,
                 int32_t SH, int32_t SW, int32_t sy0, int32_t sx0, int32_t H, int32_t W) {
    if (!score || !out) return IISEG_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || sy0 < 0 || sx0 < 0 || sy0 + H > SH || sx0 + W > SW)
        return IISEG_ERR_SHAPE;
    // (one image of every tensor is addressed with 32-bit byte offsets)
    if (C > 32 || B > 65535 || (int64_t)C * SH * SW * (int64_t)sizeof(T) >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    const dim3 grid((H * W + 255) / 256, B);
    if (C <= 16)
        IISEG_LAUNCH((crop_softmax_kernel<16, T>), grid, dim3(256), 0, (hipStream_t)stream,
                           score, minuend, out, C, SH, SW, sy0, sx0, H, W);
    else
        IISEG_LAUNCH((crop_softmax_kernel<32, T>), grid, dim3(256), 0, (hipStream_t)stream,
                           score, minuend, out, C, SH, SW, sy0, sx0, H, W);
    return iiseg_check_launch();
}

template <typename T>
int refine_update(void* stream, const T* score, T* y, const int32_t* active, double* partial,
                  int32_t B, int32_t C, int32_t SH, int32_t SW, int32_t sy0, int32_t sx0, int32_t H,
                  int32_t W, T step, void* y8 = nullptr, int32_t C8n = 0) {
    if (!score || !y || !active || !partial) return IISEG_ERR_NULL;
    if (y8 && (C8n * 8 < C || C8n > 4 || ((uintptr_t)y8 & 15))) return IISEG_ERR_SHAPE;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || sy0 < 0 || sx0 < 0 || sy0 + H > SH || sx0 + W > SW)
        return IISEG_ERR_SHAPE;
    // (one image of every tensor is addressed with 32-bit byte offsets)
    if (C > 32 || B > 65535 || (int64_t)C * SH * SW * (int64_t)sizeof(T) >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    const dim3 grid((H * W + 255) / 256, B);
    if (C <= 16)
        IISEG_LAUNCH((refine_update_kernel<16, T>), grid, dim3(256), 0, (hipStream_t)stream,
                           score, y, active, partial, C, SH, SW, sy0, sx0, H, W, step, (uint4*)y8,
                           C8n);
    else
        IISEG_LAUNCH((refine_update_kernel<32, T>), grid, dim3(256), 0, (hipStream_t)stream,
                           score, y, active, partial, C, SH, SW, sy0, sx0, H, W, step, (uint4*)y8,
                           C8n);
    return iiseg_check_launch();
}

template <typename T>
int sqerr_softmax_bwd(void* stream, const T* score, const T* y, T* gscore, int32_t B, int32_t C,
                      int32_t SH, int32_t SW, int32_t sy0, int32_t sx0, int32_t H, int32_t W) {
    if (!score || !y || !gscore) return IISEG_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || sy0 < 0 || sx0 < 0 || sy0 + H > SH || sx0 + W > SW)
        return IISEG_ERR_SHAPE;
    // (one image of every tensor is addressed with 32-bit byte offsets)
    if (C > 32 || B > 65535 || (int64_t)C * SH * SW * (int64_t)sizeof(T) >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    const dim3 grid((H * W + 255) / 256, B);
    if (C <= 16)
        IISEG_LAUNCH((sqerr_softmax_bwd_kernel<16, T>), grid, dim3(256), 0, (hipStream_t)stream,
                           score, y, gscore, C, SH, SW, sy0, sx0, H, W);
    else
        IISEG_LAUNCH((sqerr_softmax_bwd_kernel<32, T>), grid, dim3(256), 0, (hipStream_t)stream,
                           score, y, gscore, C, SH, SW, sy0, sx0, H, W);
    return iiseg_check_launch();
}

template <typename T>
int grad_update(void* stream, const T* score, const T* gthrough, T* y, const int32_t* active,
                double* partial, int32_t B, int32_t C, int32_t SH, int32_t SW, int32_t sy0,
                int32_t sx0, int32_t H, int32_t W, T step) {
    if (!score || !gthrough || !y || !active || !partial) return IISEG_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || sy0 < 0 || sx0 < 0 || sy0 + H > SH || sx0 + W > SW)
        return IISEG_ERR_SHAPE;
    // (one image of every tensor is addressed with 32-bit byte offsets)
    if (C > 32 || B > 65535 || (int64_t)C * SH * SW * (int64_t)sizeof(T) >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    const dim3 grid((H * W + 255) / 256, B);
    if (C <= 16)
        IISEG_LAUNCH((grad_update_kernel<16, T>), grid, dim3(256), 0, (hipStream_t)stream, score,
                           gthrough, y, active, partial, C, SH, SW, sy0, sx0, H, W, step);
    else
        IISEG_LAUNCH((grad_update_kernel<32, T>), grid, dim3(256), 0, (hipStream_t)stream, score,
                           gthrough, y, active, partial, C, SH, SW, sy0, sx0, H, W, step);
    return iiseg_check_launch();
}

}  // namespace

extern "C" int iiseg_sqerr_softmax_bwd_f32(void* stream, const float* score, const float* y,
                                           float* gscore, int32_t B, int32_t C, int32_t SH,
                                           int32_t SW, int32_t sy0, int32_t sx0, int32_t H, int32_t W) {
    return sqerr_softmax_bwd<float>(stream, score, y, gscore, B, C, SH, SW, sy0, sx0, H, W);
}
extern "C" int iiseg_sqerr_softmax_bwd_f64(void* stream, const double* score, const double* y,
                                           double* gscore, int32_t B, int32_t C, int32_t SH,
                                           int32_t SW, int32_t sy0, int32_t sx0, int32_t H, int32_t W) {
    return sqerr_softmax_bwd<double>(stream, score, y, gscore, B, C, SH, SW, sy0, sx0, H, W);
}
extern "C" int iiseg_grad_update_f32(void* stream, const float* score, const float* gthrough, float* y,
                                     const int32_t* active, double* partial, int32_t B, int32_t C,
                                     int32_t SH, int32_t SW, int32_t sy0, int32_t sx0, int32_t H,
                                     int32_t W, float step) {
    return grad_update<float>(stream, score, gthrough, y, active, partial, B, C, SH, SW, sy0, sx0, H,
                              W, step);
}
extern "C" int iiseg_grad_update_f64(void* stream, const double* score, const double* gthrough,
                                     double* y, const int32_t* active, double* partial, int32_t B,
                                     int32_t C, int32_t SH, int32_t SW, int32_t sy0, int32_t sx0,
                                     int32_t H, int32_t W, double step) {
    return grad_update<double>(stream, score, gthrough, y, active, partial, B, C, SH, SW, sy0, sx0, H,
                               W, step);
}

extern "C" int iiseg_crop_softmax_f32(void* stream, const float* score, const float* minuend,
                                      float* out, int32_t B, int32_t C, int32_t SH, int32_t SW,
                                      int32_t sy0, int32_t sx0, int32_t H, int32_t W) {
    return crop_softmax<float>(stream, score, minuend, out, B, C, SH, SW, sy0, sx0, H, W);
}
extern "C" int iiseg_crop_softmax_f64(void* stream, const double* score, const double* minuend,
                                      double* out, int32_t B, int32_t C, int32_t SH, int32_t SW,
                                      int32_t sy0, int32_t sx0, int32_t H, int32_t W) {
    return crop_softmax<double>(stream, score, minuend, out, B, C, SH, SW, sy0, sx0, H, W);
}

extern "C" int iiseg_refine_partials(int32_t H, int32_t W) { return (H * W + 255) / 256; }

extern "C" int iiseg_refine_update_f32(void* stream, const float* score, float* y,
                                       const int32_t* active, double* partial, int32_t B, int32_t C,
                                       int32_t SH, int32_t SW, int32_t sy0, int32_t sx0, int32_t H,
                                       int32_t W, float step) {
    return refine_update<float>(stream, score, y, active, partial, B, C, SH, SW, sy0, sx0, H, W, step);
}
extern "C" int iiseg_refine_update_c8_f32(void* stream, const float* score, float* y,
                                          const int32_t* active, double* partial, void* y8,
                                          int32_t C8n, int32_t B, int32_t C, int32_t SH, int32_t SW,
                                          int32_t sy0, int32_t sx0, int32_t H, int32_t W, float step) {
    if (!y8) return IISEG_ERR_NULL;
    return refine_update<float>(stream, score, y, active, partial, B, C, SH, SW, sy0, sx0, H, W, step,
                                y8, C8n);
}
extern "C" int iiseg_refine_update_f64(void* stream, const double* score, double* y,
                                       const int32_t* active, double* partial, int32_t B, int32_t C,
                                       int32_t SH, int32_t SW, int32_t sy0, int32_t sx0, int32_t H,
                                       int32_t W, double step) {
    return refine_update<double>(stream, score, y, active, partial, B, C, SH, SW, sy0, sx0, H, W, step);
}

extern "C" int iiseg_refine_finalize(void* stream, const double* partial, int32_t* active,
                                     int32_t* iters, double* last_norm, int32_t B, int32_t nblk,
                                     int32_t HW, double eps) {
    if (!partial || !active || !iters || !last_norm) return IISEG_ERR_NULL;
    if (B <= 0 || nblk <= 0 || HW <= 0) return IISEG_ERR_SHAPE;
    IISEG_LAUNCH(refine_finalize_kernel, dim3(B), dim3(64), 0,
                       (hipStream_t)stream, partial, active, iters, last_norm, B, nblk, HW, eps);
    return iiseg_check_launch();
}

// ---- element-wise helpers of the noise>0 mask emulation ----------------------
// The reference's DePool2D re-evaluates the down path WITHOUT deterministic=True
// (layers/mylayers.py:91-93), so with dae_dict['noise'] > 0 its masks come from a forward with
// GaussianNoiseLayer (models/fcn_down.py:60-63: x + sigma * eps) and DropoutLayer (p, rescaled:
// x * keep / (1 - p)) active.  The random tensors are the caller's (any RNG); these kernels only
// apply them.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void axpy_kernel(const T* __restrict__ x, const T* __restrict__ e,
                                                   T a, T* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = x[i] + a * e[i];
}
template <typename T>
__global__ __launch_bounds__(256) void scale_mask_kernel(T* __restrict__ x, const T* __restrict__ keep,
                                                         T scale, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        x[i] = x[i] * keep[i] * scale;
}
inline int ew_grid(size_t n) {
    const size_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}
template <typename T>
int add_noise(void* stream, const T* x, const T* eps, T sigma, T* out, int64_t n) {
    if (!x || !eps || !out) return IISEG_ERR_NULL;
    if (n <= 0) return IISEG_ERR_SHAPE;
    IISEG_LAUNCH(axpy_kernel<T>, dim3(ew_grid((size_t)n)), dim3(256), 0, (hipStream_t)stream, x,
                       eps, sigma, out, (size_t)n);
    return iiseg_check_launch();
}
template <typename T>
int dropout_apply(void* stream, T* x, const T* keep, T p, int64_t n) {
    if (!x || !keep) return IISEG_ERR_NULL;
    if (n <= 0 || !(p >= 0) || !(p < 1)) return IISEG_ERR_SHAPE;
    IISEG_LAUNCH(scale_mask_kernel<T>, dim3(ew_grid((size_t)n)), dim3(256), 0,
                       (hipStream_t)stream, x, keep, (T)1 / ((T)1 - p), (size_t)n);
    return iiseg_check_launch();
}
}  // namespace

extern "C" int iiseg_add_noise_f32(void* stream, const float* x, const float* eps, float sigma,
                                   float* out, int64_t n) {
    return add_noise<float>(stream, x, eps, sigma, out, n);
}
extern "C" int iiseg_add_noise_f64(void* stream, const double* x, const double* eps, double sigma,
                                   double* out, int64_t n) {
    return add_noise<double>(stream, x, eps, sigma, out, n);
}
extern "C" int iiseg_dropout_apply_f32(void* stream, float* x, const float* keep, float p, int64_t n) {
    return dropout_apply<float>(stream, x, keep, p, n);
}
extern "C" int iiseg_dropout_apply_f64(void* stream, double* x, const double* keep, double p,
                                       int64_t n) {
    return dropout_apply<double>(stream, x, keep, p, n);
}
