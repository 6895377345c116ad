// The FCN-8 upsamplers (Deconv2DLayer 11 -> 11, k4 s2 twice and k16 s8: models/fcn8.py:90,100,109) as
// K = 2 s transposed convolutions by OUTPUT PHASE, on the vector ALU with scalar weight operands.
//
// With K = 2 s an output pixel (Y, X) = (s i + py, s j + px) has 2 x 2 taps, a = py + s dy, b = px + s dx,
// reading x[o, i - dy, j - dx]: per phase (py, px) the layer is a 2x2 convolution of the INPUT grid,
//   out[c, s i + py, s j + px] = bias[c] + sum_{dy, dx, o} x[o, i - dy, j - dx] W[o, c, K-1-py-s dy, K-1-px-s dx].
// The gather kernel (deconv.hip) gives a thread one output pixel: its weights depend on the lane's phase, so
// they are vector loads -- 484 of them next to 484 FMAs per pixel of the k16 s8 layer, 0.235 ms per launch for
// 141 MB of output (18 us of HBM time).  Here a thread owns one INPUT position (i, j) and one row phase py (the
// block's): the s pixels of its output row segment x all output channels live in registers, every weight
// W[., ., py-row, .] is the same for all lanes of the wave -- a scalar load, an SGPR operand of the FMA (packed
// pairs of output channels in fp32) -- the x values are four coalesced loads per input channel and the stores
// are 4 s bytes contiguous per lane.  Sum order per output value: bias, then (dy, dx) in the gather kernel's tap
// order, input channels ascending -- the same chain of FMAs, bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "iiseg.h"
#include "common.h"
#include "conv_common.h"

using namespace iiseg;

namespace {

constexpr unsigned OOB = 0x80000000u;
constexpr int RSRC_W3 = 0x00027000;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct PhaseParams {
    const T* x;
    const T* wp;      // packed weights [py][dy][dx][o][px][CP] (iiseg_deconv_phase_pack_*)
    const T* bias;
    const T* add;
    T* out;
    int B, Cin, H, W, Cout;
    int oy0, ox0, OH, OW;
    int AH, AW, ay0, ax0;
    int i_lo, j_lo, ni, nj;   // input-grid positions that reach the window
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, RSRC_W3);
}
__device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned so, float) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, (int)so, 0));
}
__device__ __forceinline__ double ld(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned so, double) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, (int)so, 0));
}
__device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0);
}
__device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, unsigned off, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)off, 0, 0);
}

// S: stride (K = 2 S).  CP: output channels padded to a multiple of 4.  PXB: pixels of the row segment per pass
// (fp32: all S; float64 at S = 8: two passes of four -- 48 accumulator pairs a pass).  ADD: skip tensor fused.
// Grid: (position tiles of 64, S row phases, B images); one wave per block.
template <typename T, int S, int CP, int PXB, bool ADD>
__global__ __launch_bounds__(64) void deconv_phase_kernel(const PhaseParams<T> p) {
    constexpr int TS = (int)sizeof(T);
    constexpr int EPV = 16 / TS;                       // weights per 16-byte scalar load
    typedef typename std::conditional<TS == 4, f32x4, f64x2>::type WV;
    const int py = blockIdx.y, b = blockIdx.z;
    const int pos = blockIdx.x * 64 + threadIdx.x;
    const bool ok = pos < p.ni * p.nj;
    const int ii = pos / p.nj, jj = pos - ii * p.nj;
    const int i = p.i_lo + ii, j = p.j_lo + jj;
    const int wy = i * S + py - p.oy0;                 // window row of the segment
    const bool row_ok = ok && wy >= 0 && wy < p.OH;
    const int HW = p.H * p.W;
    unsigned xoff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int iy = i - (t >> 1), ix = j - (t & 1);
        xoff[t] = (row_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                      ? (unsigned)(iy * p.W + ix) * TS : OOB;
    }
    const __amdgpu_buffer_rsrc_t rx = mk(p.x + (size_t)b * p.Cin * HW, (unsigned)(p.Cin * HW) * TS);
    const __amdgpu_buffer_rsrc_t ro =
        mk(p.out + (size_t)b * p.Cout * p.OH * p.OW, (unsigned)(p.Cout * p.OH * p.OW) * TS);
    const __amdgpu_buffer_rsrc_t ra =
        mk(ADD ? p.add + (size_t)b * p.Cout * p.AH * p.AW : nullptr, ADD ? (unsigned)(p.Cout * p.AH * p.AW) * TS : 0u);
    const T* wpy = p.wp + (size_t)py * (4 * p.Cin * S * CP);
    const int nm = 4 * p.Cin;

#pragma unroll 1
    for (int pxb = 0; pxb < S; pxb += PXB) {
        T acc[PXB][CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            const T bv = (p.bias && c < p.Cout) ? p.bias[c] : (T)0;
#pragma unroll
            for (int px = 0; px < PXB; ++px) acc[px][c] = bv;
        }
        // weights of (tap t, channel o): PXB x CP values at wpy + ((t Cin + o) S + pxb) CP, fetched by scalar
        // loads in chunks of PC pixels (24 - 32 dwords), chunk q + 1 while chunk q multiplies (two SGPR sets; the
        // order is pinned: left to itself hipcc hoists whole rows and spills scalar registers); x[o] of the
        // tap one channel ahead
        constexpr int PC = (TS == 4 && S == 8) ? 2 : 1;
        constexpr int NCH = PXB / PC, NV = PC * CP / EPV;
        static_assert(NCH % 2 == 0, "the ring slot of a chunk is a compile-time index");
        WV wq[2][NV];
        auto wload = [&](int m, int q, auto SL) __attribute__((always_inline)) {
            constexpr int sl = decltype(SL)::value;
            const T* wr = wpy + ((size_t)m * S + pxb + q * PC) * CP;
#pragma unroll
            for (int v = 0; v < NV; ++v) wq[sl][v] = *reinterpret_cast<const WV*>(wr + v * EPV);
        };
        T xv = ld(rx, xoff[0], 0u, T());
        wload(0, 0, ic<0>{});
        int m = 0;
        static_for<0, 4>([&](auto TT) __attribute__((always_inline)) {
            constexpr int t = decltype(TT)::value;
#pragma unroll 1
            for (int o = 0; o < p.Cin; ++o, ++m) {
                const T xc = xv;
                const bool last_o = o + 1 >= p.Cin;
                xv = ld(rx, last_o ? xoff[t < 3 ? t + 1 : 3] : xoff[t], last_o ? 0u : (unsigned)((o + 1) * HW) * TS, T());
                const int mn = m + 1 < nm ? m + 1 : m;
                static_for<0, NCH>([&](auto QQ) __attribute__((always_inline)) {
                    constexpr int q = decltype(QQ)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (q + 1 < NCH) wload(m, q + 1, ic<(q + 1) & 1>{});
                    else wload(mn, 0, ic<0>{});
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int px = 0; px < PC; ++px) {
                        if constexpr (TS == 4) {
#pragma unroll
                            for (int c = 0; c < CP; c += 2) {
                                const int e = px * CP + c;
                                const f32x2 w2 = {(float)wq[q & 1][e / EPV][e % EPV], (float)wq[q & 1][e / EPV][e % EPV + 1]};
                                const f32x2 x2 = {(float)xc, (float)xc};
                                f32x2 a2 = {(float)acc[q * PC + px][c], (float)acc[q * PC + px][c + 1]};
                                a2 = __builtin_elementwise_fma(x2, w2, a2);
                                acc[q * PC + px][c] = a2[0]; acc[q * PC + px][c + 1] = a2[1];
                            }
                        } else {
#pragma unroll
                            for (int c = 0; c < CP; ++c) {
                                const int e = px * CP + c;
                                acc[q * PC + px][c] = __builtin_fma(xc, wq[q & 1][e / EPV][e % EPV], acc[q * PC + px][c]);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        });
        // ---- the row segment: PXB pixels x Cout channels, 16-byte stores where the window allows ----
        constexpr int G = EPV;                            // pixels per 16-byte store
        const int wx0 = j * S + pxb - p.ox0;              // window column of the first pixel
        bool pok[PXB];
#pragma unroll
        for (int px = 0; px < PXB; ++px) pok[px] = row_ok && wx0 + px >= 0 && wx0 + px < p.OW;
        T ad[ADD ? PXB : 1][ADD ? CP : 1];
        if constexpr (ADD) {
#pragma unroll
            for (int c = 0; c < CP; ++c)
#pragma unroll
                for (int px = 0; px < PXB; ++px)
                    ad[px][c] = ld(ra, (pok[px] && c < p.Cout)
                                           ? (unsigned)((c * p.AH + p.ay0 + wy) * p.AW + p.ax0 + wx0 + px) * TS : OOB,
                                   0u, T());
        }
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            if (c >= p.Cout) continue;                    // (uniform)
            const unsigned o0 = (unsigned)((c * p.OH + wy) * p.OW + wx0) * TS;
#pragma unroll
            for (int g0 = 0; g0 < PXB; g0 += G) {
                T v[G];
#pragma unroll
                for (int e = 0; e < G; ++e) {
                    v[e] = g0 + e < PXB ? acc[g0 + e < PXB ? g0 + e : 0][c] : (T)0;
                    if constexpr (ADD) v[e] += ad[g0 + e < PXB ? g0 + e : 0][c];
                }
                bool all = g0 + G <= PXB;
#pragma unroll
                for (int e = 0; e < G; ++e) all = all && (g0 + e < PXB ? pok[g0 + e] : false);
                if (all) {
                    u32x4 w4;
                    if constexpr (TS == 4) {
                        const f32x4 f = {(float)v[0], (float)v[1], (float)v[G > 2 ? 2 : 0], (float)v[G > 3 ? 3 : 0]};
                        w4 = __builtin_bit_cast(u32x4, f);
                    } else {
                        const f64x2 f = {(double)v[0], (double)v[1]};
                        w4 = __builtin_bit_cast(u32x4, f);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(w4, ro, (int)(o0 + (unsigned)g0 * TS), 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < G; ++e)
                        if (g0 + e < PXB)
                            st(ro, pok[g0 + e] ? o0 + (unsigned)(g0 + e) * TS : OOB, v[e]);
                }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void deconv_phase_pack_kernel(const T* __restrict__ w, T* __restrict__ wp, int Cin,
                                                                int Cout, int S, int CP) {
    // wp[py][t = 2 dy + dx][o][px][c] = W[o][c][K - 1 - py - S dy][K - 1 - px - S dx]  (0 for c >= Cout)
    const int K = 2 * S;
    const int n = S * 4 * Cin * S * CP;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    int r = e;
    const int c = r % CP; r /= CP;
    const int px = r % S; r /= S;
    const int o = r % Cin; r /= Cin;
    const int t = r % 4; r /= 4;
    const int py = r;
    const int ky = K - 1 - py - S * (t >> 1), kx = K - 1 - px - S * (t & 1);
    wp[e] = c < Cout ? w[(((size_t)o * Cout + c) * K + ky) * K + kx] : (T)0;
}

bool phase_ok(const iiseg_deconv_desc* d, int tsize) {
    static const int on = getenv("IISEG_DECONV_PHASE") ? atoi(getenv("IISEG_DECONV_PHASE")) : 1;
    if (!on || !d) return false;
    if (d->K != 2 * d->stride || (d->stride != 2 && d->stride != 8)) return false;
    if (d->Cin <= 0 || d->Cin > 16 || d->Cout <= 0 || d->Cout > 16) return false;
    if (d->B <= 0 || d->B > 65535 || d->H <= 0 || d->W <= 0 || d->OH <= 0 || d->OW <= 0) return false;
    // one image of every tensor within 32-bit byte offsets
    if ((int64_t)d->Cin * d->H * d->W * tsize >= (1ll << 31)) return false;
    if ((int64_t)d->Cout * d->OH * d->OW * tsize >= (1ll << 31)) return false;
    if ((int64_t)d->Cout * d->AH * d->AW * tsize >= (1ll << 31)) return false;
    return true;
}

int cp_of(int Cout) { return Cout <= 12 ? 12 : 16; }

template <typename T>
int launch_phase(void* stream, const iiseg_deconv_desc* d, const T* x, const T* wp, const T* bias, const T* add,
                 T* out) {
    if (!d || !x || !wp || !out) return IISEG_ERR_NULL;
    if (!phase_ok(d, (int)sizeof(T))) return IISEG_ERR_UNSUPPORTED;
    const int S = d->stride;
    const int fullH = (d->H - 1) * S + d->K, fullW = (d->W - 1) * S + d->K;
    if (d->oy0 < 0 || d->ox0 < 0 || d->oy0 + d->OH > fullH || d->ox0 + d->OW > fullW) return IISEG_ERR_SHAPE;
    if (add && (d->ay0 < 0 || d->ax0 < 0 || d->AH < d->ay0 + d->OH || d->AW < d->ax0 + d->OW)) return IISEG_ERR_SHAPE;
    PhaseParams<T> p;
    p.x = x; p.wp = wp; p.bias = bias; p.add = add; p.out = out;
    p.B = d->B; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout;
    p.oy0 = d->oy0; p.ox0 = d->ox0; p.OH = d->OH; p.OW = d->OW;
    p.AH = d->AH; p.AW = d->AW; p.ay0 = d->ay0; p.ax0 = d->ax0;
    p.i_lo = d->oy0 / S; p.j_lo = d->ox0 / S;
    p.ni = (d->oy0 + d->OH - 1) / S - p.i_lo + 1;
    p.nj = (d->ox0 + d->OW - 1) / S - p.j_lo + 1;
    const dim3 grid((p.ni * p.nj + 63) / 64, S, d->B), block(64);
    hipStream_t s = (hipStream_t)stream;
    const int cp = cp_of(d->Cout);
    constexpr bool F64 = sizeof(T) == 8;
#define PHASE_LAUNCH(SS, CPV, PXBV, ADDV) \
    IISEG_LAUNCH((deconv_phase_kernel<T, SS, CPV, PXBV, ADDV>), grid, block, 0, s, p)
    if (S == 8) {
        if (add) return IISEG_ERR_UNSUPPORTED;            // (no such layer: the gather kernel runs it)
        if (cp == 12) PHASE_LAUNCH(8, 12, (F64 ? 4 : 8), false); else PHASE_LAUNCH(8, 16, (F64 ? 4 : 8), false);
    } else {
        if (add) { if (cp == 12) PHASE_LAUNCH(2, 12, 2, true); else PHASE_LAUNCH(2, 16, 2, true); }
        else { if (cp == 12) PHASE_LAUNCH(2, 12, 2, false); else PHASE_LAUNCH(2, 16, 2, false); }
    }
#undef PHASE_LAUNCH
    return iiseg_check_launch();
}

template <typename T>
int pack_phase(void* stream, const iiseg_deconv_desc* d, const T* w, T* wp) {
    if (!d || !w || !wp) return IISEG_ERR_NULL;
    if (!phase_ok(d, (int)sizeof(T))) return IISEG_ERR_UNSUPPORTED;
    const int S = d->stride, cp = cp_of(d->Cout);
    const int n = S * 4 * d->Cin * S * cp;
    IISEG_LAUNCH((deconv_phase_pack_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, wp,
                 d->Cin, d->Cout, S, cp);
    return iiseg_check_launch();
}

}  // namespace

extern "C" int iiseg_deconv_phase_supported(const iiseg_deconv_desc* d, int has_add, int is_f64) {
    if (!phase_ok(d, is_f64 ? 8 : 4)) return 0;
    return (d->stride == 8 && has_add) ? 0 : 1;
}
extern "C" int64_t iiseg_deconv_phase_weight_elems(const iiseg_deconv_desc* d) {
    if (!phase_ok(d, 8)) return 0;
    return (int64_t)d->stride * 4 * d->Cin * d->stride * cp_of(d->Cout);
}
extern "C" int iiseg_deconv_phase_pack_f32(void* stream, const iiseg_deconv_desc* d, const float* w, float* wp) {
    return pack_phase<float>(stream, d, w, wp);
}
extern "C" int iiseg_deconv_phase_pack_f64(void* stream, const iiseg_deconv_desc* d, const double* w, double* wp) {
    return pack_phase<double>(stream, d, w, wp);
}
extern "C" int iiseg_deconv_phase_f32(void* stream, const iiseg_deconv_desc* d, const float* x, const float* wp,
                                      const float* bias, const float* add, float* out) {
    return launch_phase<float>(stream, d, x, wp, bias, add, out);
}
extern "C" int iiseg_deconv_phase_f64(void* stream, const iiseg_deconv_desc* d, const double* x, const double* wp,
                                      const double* bias, const double* add, double* out) {
    return launch_phase<double>(stream, d, x, wp, bias, add, out);
}
