// Metrics accumulator: argmax confusion matrix (prediction x truth incl. void column) and the
// void-masked squared error, one pixel per thread, LDS-binned counts, one global atomic per
// bin per block.  Integer counts are exact and order-independent.
// Replaces val_fn (reference iterative_inference.py:206-210): metrics.py:11-37 (jaccard's 121
// masked reductions), :40-65 (accuracy), :144-156 (squared_error).
#include "common.h"
#include "column_io.h"
#include "conv_common.h"

namespace {

constexpr int MAXC = 32;
constexpr int PPT = 8;   // pixels per thread

// CMAX: planes of a pixel held in registers (C + 1 <= CMAX).  The C prediction and C + 1 target values of a pixel
// are loaded as two channel columns through buffer descriptors (column_io.h) -- all in flight before the
// first compare; with a run-time channel loop the loads went out two at a time, each pair waited for
// (0.115 ms for 295 MB).
template <typename T, int CMAX>
__global__ __launch_bounds__(256) void confusion_kernel(const T* __restrict__ y,
                                                        const T* __restrict__ t,
                                                        const int* __restrict__ active,
                                                        unsigned long long* __restrict__ cm,
                                                        double* __restrict__ sums, int C, int HW) {
    __shared__ unsigned int bins[MAXC * (MAXC + 1)];
    __shared__ double red[2][4];
    const int nb = C * (C + 1);
    for (int i = threadIdx.x; i < nb; i += 256) bins[i] = 0;
    __syncthreads();
    const int b = blockIdx.y;
    double se = 0.0, mk = 0.0;
    // images whose flag is 0 are skipped (per-iteration metrics of the validation driver only
    // count images still being refined, iterative_inference_valid.py:280-288)
    const bool on = !active || active[b] != 0;
    const unsigned PB = (unsigned)HW * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t ry = t_rsrc(y + (size_t)b * C * HW, (unsigned)C * PB);
    const __amdgpu_buffer_rsrc_t rt = t_rsrc(t + (size_t)b * (C + 1) * HW, (unsigned)(C + 1) * PB);
    // PPT pixels per thread: 8x fewer blocks, i.e. 8x fewer global atomics on the two `sums` words
    // and the bins every block ends with (12 544 blocks hitting two addresses made this kernel 5x
    // slower than its 0.3 GB of traffic)
    // software pipeline: the 2 C + 1 loads of pixel q + 1 are in flight while pixel q is compared and binned
    T yv[2][CMAX], tv[2][CMAX];
    auto fetch = [&](int q, auto S) __attribute__((always_inline)) {
        constexpr int sl = decltype(S)::value;
        const int pix = (blockIdx.x * PPT + q) * 256 + threadIdx.x;
        const unsigned off = (q < PPT && pix < HW && on) ? (unsigned)pix * (unsigned)sizeof(T) : T_OOB;
        load_column<CMAX, T>(ry, off, PB, C, yv[sl]);
        load_column<CMAX, T>(rt, off, PB, C + 1, tv[sl]);
    };
    auto bin = [&](int q, auto S) __attribute__((always_inline)) {
        constexpr int sl = decltype(S)::value;
        const int pix = (blockIdx.x * PPT + q) * 256 + threadIdx.x;
        const bool live = pix < HW && on;
        // argmax returns the FIRST maximal index (T.argmax / np.argmax)
        int ip = 0, it = 0;
        T bp = yv[sl][0], bt = tv[sl][0];
        T msum = 0, esum = 0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) {
                if (yv[sl][c] > bp) { bp = yv[sl][c]; ip = c; }
                if (tv[sl][c] > bt) { bt = tv[sl][c]; it = c; }
                msum += tv[sl][c];                    // mask = y_true[:, :void].sum(1)   metrics.py:148
                esum = fma(yv[sl][c] - tv[sl][c], yv[sl][c] - tv[sl][c], esum);
            }
        T tvoid = 0;
#pragma unroll
        for (int c = 1; c < CMAX; ++c)
            if (c == C) tvoid = tv[sl][c];
        if (tvoid > bt) it = C;
        if (live) {
            atomicAdd(&bins[ip * (C + 1) + it], 1u);
            se += (double)(esum / (T)C) * (double)msum;  // .mean(axis=1) * mask   :147,153
            mk += (double)msum;
        }
    };
    static_assert(PPT % 2 == 0, "two pixels per trip");
    fetch(0, iiseg::ic<0>{});
#pragma unroll 1
    for (int q = 0; q < PPT; q += 2) {
        fetch(q + 1, iiseg::ic<1>{});
        __builtin_amdgcn_sched_barrier(0);
        bin(q, iiseg::ic<0>{});
        fetch(q + 2, iiseg::ic<0>{});
        __builtin_amdgcn_sched_barrier(0);
        bin(q + 1, iiseg::ic<1>{});
    }
    se = wave_sum(se);
    mk = wave_sum(mk);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = se;
        red[1][threadIdx.x >> 6] = mk;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 256)
        if (bins[i]) atomicAdd(&cm[i], (unsigned long long)bins[i]);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        atomicAdd(&sums[1], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

template <typename T>
int confusion(void* stream, const T* y, const T* t, const int32_t* active, int64_t* cm, double* sums,
              int32_t B, int32_t C, int32_t HW) {
    if (!y || !t || !cm || !sums) return IISEG_ERR_NULL;
    if (B <= 0 || C <= 0 || HW <= 0) return IISEG_ERR_SHAPE;
    // (one image of either tensor is addressed with 32-bit byte offsets)
    if (C >= MAXC || B > 65535 || (int64_t)(C + 1) * HW * (int64_t)sizeof(T) >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    const dim3 grid((HW + 256 * PPT - 1) / (256 * PPT), B);
    if (C + 1 <= 16)
        IISEG_LAUNCH((confusion_kernel<T, 16>), grid, dim3(256), 0, (hipStream_t)stream, y, t, active,
                     reinterpret_cast<unsigned long long*>(cm), sums, C, HW);
    else
        IISEG_LAUNCH((confusion_kernel<T, 32>), grid, dim3(256), 0, (hipStream_t)stream, y, t, active,
                     reinterpret_cast<unsigned long long*>(cm), sums, C, HW);
    return iiseg_check_launch();
}

}  // namespace

extern "C" int iiseg_confusion_f32(void* stream, const float* y, const float* t, int64_t* cm,
                                   double* sums, int32_t B, int32_t C, int32_t HW) {
    return confusion<float>(stream, y, t, nullptr, cm, sums, B, C, HW);
}
extern "C" int iiseg_confusion_f64(void* stream, const double* y, const double* t, int64_t* cm,
                                   double* sums, int32_t B, int32_t C, int32_t HW) {
    return confusion<double>(stream, y, t, nullptr, cm, sums, B, C, HW);
}

extern "C" int iiseg_confusion_masked_f32(void* stream, const float* y, const float* t,
                                          const int32_t* active, int64_t* cm, double* sums,
                                          int32_t B, int32_t C, int32_t HW) {
    if (!active) return IISEG_ERR_NULL;
    return confusion<float>(stream, y, t, active, cm, sums, B, C, HW);
}
extern "C" int iiseg_confusion_masked_f64(void* stream, const double* y, const double* t,
                                          const int32_t* active, int64_t* cm, double* sums,
                                          int32_t B, int32_t C, int32_t HW) {
    if (!active) return IISEG_ERR_NULL;
    return confusion<double>(stream, y, t, active, cm, sums, B, C, HW);
}

// ---- non-finite inputs -----------------------------------------------------------------------------------------
// The fp32 / bf16 convolution sources are built with relaxed NaN handling (build.py EXTRA_FLAGS: the ReLU and
// max-pool epilogues are bare v_max_f32 on MFMA results), so a NaN that ENTERS the network is not guaranteed to
// come out as a NaN -- the first ReLU may swallow it (the reference's Theano `maximum` propagates it).  What the
// product promises instead: non-finite values are detected where they enter.  Weights are checked on the host
// when a layer is built; every image batch is counted here (one HBM-bound pass, no synchronisation: the counter
// stays on the device and is read with the results, api.Metrics.result / IterativeInference.check_finite).
namespace {
template <typename T>
__global__ __launch_bounds__(256) void count_nonfinite_kernel(const T* __restrict__ x, int64_t n, int* counter) {
    int bad = 0;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const T v = x[i];
        // (integer test of the exponent field: immune to the compiler's no-NaN assumptions)
        if constexpr (sizeof(T) == 4)
            bad += (__builtin_bit_cast(unsigned, v) & 0x7f800000u) == 0x7f800000u;
        else
            bad += (__builtin_bit_cast(unsigned long long, v) & 0x7ff0000000000000ull) == 0x7ff0000000000000ull;
    }
    if (__any(bad != 0) && bad) atomicAdd(counter, bad);
}
template <typename T>
int count_nonfinite(void* stream, const T* x, int64_t n, int32_t* counter) {
    if (!x || !counter) return IISEG_ERR_NULL;
    if (n <= 0) return IISEG_ERR_SHAPE;
    const int64_t blocks = (n + 255) / 256;
    const int grid = (int)(blocks < 4096 ? blocks : 4096);
    IISEG_LAUNCH(count_nonfinite_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, counter);
    return iiseg_check_launch();
}
}  // namespace

extern "C" int iiseg_count_nonfinite_f32(void* stream, const float* x, int64_t n, int32_t* counter) {
    return count_nonfinite<float>(stream, x, n, counter);
}
extern "C" int iiseg_count_nonfinite_f64(void* stream, const double* x, int64_t n, int32_t* counter) {
    return count_nonfinite<double>(stream, x, n, counter);
}
