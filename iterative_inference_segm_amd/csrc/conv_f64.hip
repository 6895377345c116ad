// float64 static-tap implicit-GEMM convolution (1x1 / 3x3, any pad / dilation, stride 1) on
// v_mfma_f64_16x16x4_f64 -- the strict-parity variant of conv_taps.hip.
//
// Why it exists: the reference's CPU path computes in float64 (Theano floatX default)
// and DePool2D compares activations for exact equality (layers/mylayers.py:111-114).  In fp32 a
// handful of near-tied pooling windows per image pick a different maximum than float64 does and
// the difference spreads through the decoder.  With float64 arithmetic the
// HIP path reproduces the oracle's mask decisions, so the refined map matches end to end.
//
// Same structure as conv_taps.hip: k-tile = CPT whole channels x T taps, per-lane tap offsets
// computed once, gather = buffer_load_dwordx2 with the hardware range check supplying the zero
// padding.  Tile 64 channels x 128 pixels, 4 waves of 64 x 32 (4 x 2 MFMA tiles of 16x16), k-step 4.
// fp64 MFMA peak on MI355X is half the fp32 one (78.6 TFLOP/s); this kernel is written for
// exactness first (single LDS buffer, register-staged prefetch).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "iiseg.h"
#include "common.h"
#include "conv_common.h"
#include "conv_f64_common.h"

using namespace iiseg;

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr int RSRC_W3 = 0x00027000;
constexpr unsigned OOB = 0x80000000u;
constexpr int BM = 64, BN = 128;


__device__ __forceinline__ double buf_ld64(const double* base, int bytes, unsigned voff, unsigned soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, RSRC_W3);
    i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
    return __builtin_bit_cast(double, v);
}

template <int KH, int KW, int CPT, bool UNPOOL>
__global__ __launch_bounds__(256) void conv_taps_f64_kernel(const ConvParams64 p) {
    constexpr int T = KH * KW;
    constexpr int BK = CPT * T;
    constexpr int NKS = BK / 4;          // MFMA k-steps per tile
    constexpr int TM = 4, TN = 2;        // 16x16 tiles per wave: 64 channels x 32 pixels
    constexpr int CPG = CPT / 2;         // channels staged per thread (2 row groups of 128 px)
    constexpr int XE = CPG * T;
    constexpr int WVEC = BK * BM / 2;    // double2 per weight tile
    constexpr int WPT = (WVEC + 255) / 256;
    static_assert(BK % 4 == 0 && CPT % 2 == 0, "tile config");

    __shared__ __attribute__((aligned(16))) double Ws[BK][BM];
    __shared__ __attribute__((aligned(16))) double Xs[BK][BN];

    int pt, mt;
    tile_of_block(blockIdx.x, gridDim.x, p.n_ptiles, p.n_mtiles, pt, mt);
    const int p0 = pt * BN, m0 = mt * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int OHW = p.OH * p.OW, HW = p.H * p.W, hw2 = p.h2 * p.w2;

    const int lp = tid % BN;
    const int rg = __builtin_amdgcn_readfirstlane(tid / BN);
    const int pg = p0 + lp;
    const bool pvalid = pg < p.P;
    int gb = 0, goy = 0, gox = 0;
    if (pvalid) {
        gb = pg / OHW;
        const int rem = pg - gb * OHW;
        goy = rem / p.OW;
        gox = rem - goy * p.OW;
    }
    goy += p.oy0;
    gox += p.ox0;
    const int b0 = __builtin_amdgcn_readfirstlane(p0 / OHW);
    const int nb = min(p.B - b0, BN / OHW + 2);
    const int db = gb - b0;
    const int C1 = p.C1, C2 = p.C2, Ctot = C1 + C2;

    const double *base1, *base2, *basep = nullptr;
    int n1, n2, np = 0;
    if constexpr (UNPOOL) {
        base1 = p.x1 + (size_t)b0 * C1 * hw2;
        base2 = p.pooled + (size_t)b0 * C1 * hw2;
        basep = p.pre + (size_t)b0 * C1 * HW;
        n1 = n2 = nb * C1 * hw2 * 8;
        np = nb * C1 * HW * 8;
    } else {
        base1 = p.x1 + (size_t)b0 * C1 * HW;
        base2 = C2 > 0 ? p.x2 + (size_t)b0 * C2 * HW : base1;
        n1 = nb * C1 * HW * 8;
        n2 = C2 > 0 ? nb * C2 * HW * 8 : n1;
    }

    unsigned voff[T], voff2[T];
    static_for<0, T>([&](auto TT) __attribute__((always_inline)) {
        constexpr int t = decltype(TT)::value;
        int iy = goy + (t / KW) * p.dil - p.pad;
        int ix = gox + (t % KW) * p.dil - p.pad;
        bool par = true;
        if (p.transposed) {
            const int ty = goy - t / KW, tx = gox - t % KW;
            par = ty >= 0 && tx >= 0 && !((ty | tx) & 1);
            iy = ty >> 1;
            ix = tx >> 1;
        }
        bool ok = pvalid && par && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        voff[t] = ok ? 8u * (unsigned)(db * C1 * HW + iy * p.W + ix) : OOB;
        if constexpr (UNPOOL) {
            ok = ok && iy < 2 * p.h2 && ix < 2 * p.w2;
            voff2[t] = ok ? 8u * (unsigned)(db * C1 * hw2 + (iy >> 1) * p.w2 + (ix >> 1)) : OOB;
        } else {
            voff2[t] = ok ? 8u * (unsigned)(db * C2 * HW + iy * p.W + ix) : OOB;
        }
    });

    f64x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

    double xv[XE];
    double xq[UNPOOL ? XE : 1];
    double xu[UNPOOL ? XE : 1];
    double2 wv[WPT];
    const int wrow0 = tid / (BM / 2);
    constexpr int RPJ = 256 / (BM / 2);
    const int wc2 = tid % (BM / 2);

    auto gather = [&](int kt) __attribute__((always_inline)) {
        static_for<0, XE>([&](auto JJ) __attribute__((always_inline)) {
            constexpr int j = decltype(JJ)::value;
            constexpr int t = j % T;
            const int c = min(kt * CPT + rg * CPG + j / T, Ctot - 1);  // k padding: zero weights
            if constexpr (UNPOOL) {
                xv[j] = buf_ld64(basep, np, voff[t], (unsigned)(c * HW) * 8u);
                xq[j] = buf_ld64(base2, n2, voff2[t], (unsigned)(c * hw2) * 8u);
                xu[j] = buf_ld64(base1, n1, voff2[t], (unsigned)(c * hw2) * 8u);
            } else {
                const bool s1 = c < C1;
                xv[j] = buf_ld64(s1 ? base1 : base2, s1 ? n1 : n2, s1 ? voff[t] : voff2[t],
                                 (unsigned)((s1 ? c : c - C1) * HW) * 8u);
            }
        });
        static_for<0, WPT>([&](auto JJ) __attribute__((always_inline)) {
            constexpr int j = decltype(JJ)::value;
            if ((j + 1) * 256 <= WVEC || tid + 256 * j < WVEC)
                wv[j] = *reinterpret_cast<const double2*>(
                    p.wp + (size_t)(kt * BK + wrow0 + j * RPJ) * p.Mpad + m0 + wc2 * 2);
        });
    };
    auto store = [&]() __attribute__((always_inline)) {
        static_for<0, XE>([&](auto JJ) __attribute__((always_inline)) {
            constexpr int j = decltype(JJ)::value;
            double v = xv[j];
            if constexpr (UNPOOL) v = (xv[j] == xq[j]) ? xu[j] : 0.0;
            Xs[rg * XE + j][lp] = v;
        });
        static_for<0, WPT>([&](auto JJ) __attribute__((always_inline)) {
            constexpr int j = decltype(JJ)::value;
            if ((j + 1) * 256 <= WVEC || tid + 256 * j < WVEC)
                *reinterpret_cast<double2*>(&Ws[wrow0 + j * RPJ][wc2 * 2]) = wv[j];
        });
    };

    const int nkt = p.Kpad / BK;
    const int l15 = lane & 15, lq = lane >> 4;
    gather(0);
    for (int kt = 0; kt < nkt; ++kt) {
        store();
        __syncthreads();
        if (kt + 1 < nkt) gather(kt + 1);  // in flight while the MFMAs below run
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int kk = ks * 4 + lq;
            double a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Ws[kk][i * 16 + l15];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Xs[kk][wave * 32 + j * 16 + l15];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue.  f64 16x16x4 C/D layout: column = lane & 15 (pixel), row = (lane >> 4) + 4*r
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pe = p0 + wave * 32 + j * 16 + l15;
        if (pe >= p.P) continue;
        const int eb = pe / OHW;
        const int rem = pe - eb * OHW;
        const int eoy = rem / p.OW, eox = rem - eoy * p.OW;
        const size_t OPL = (size_t)p.out_H * p.out_W;
        double* outp = p.out + ((size_t)eb * p.out_ctot + p.out_c0) * OPL +
                       (size_t)(p.out_y0 + eoy) * p.out_W + p.out_x0 + eox;
        const double* addp = nullptr;
        size_t AHW = 0;
        if (p.add) {
            AHW = (size_t)p.AH * p.AW;
            addp = p.add + (size_t)eb * p.Cout * AHW + (size_t)(p.ay0 + eoy) * p.AW + p.ax0 + eox;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m0 + i * 16 + lq + 4 * r;
                if (co < p.Cout) {
                    double v = acc[i][j][r];
                    if (p.bias) v += p.bias[co];
                    if (addp) v += addp[(size_t)co * AHW];
                    if (p.relu) v = fmax(v, 0.0);
                    outp[(size_t)co * OPL] = v;
                }
            }
    }
}

__global__ void conv_pack_f64_kernel(const double* __restrict__ w, int64_t so, int64_t sc,
                                     double* wp, int KK, int Cout, int K, int Kpad, int Mpad,
                                     int flip) {
    const int64_t n = (int64_t)Kpad * Mpad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / Mpad), m = (int)(i % Mpad);
        double v = 0.0;
        if (k < K && m < Cout) v = w[m * so + (k / KK) * sc + (flip ? KK - 1 - k % KK : k % KK)];
        wp[i] = v;
    }
}

// (B, C, H, W) -> (B, C*KH*KW, OH, OW) valid patches: turns the 7x7 fc6 into a 1x1 convolution
// whose weight matrix is the reference W[out][in*7*7] as it lies in memory.
__global__ void im2col_f64_kernel(const double* __restrict__ x, double* __restrict__ out, int B,
                                  int C, int H, int W, int KH, int KW, int OH, int OW) {
    const size_t n = (size_t)B * C * KH * KW * OH * OW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % OW);
        size_t t = i / OW;
        const int oy = (int)(t % OH);
        t /= OH;
        const int kx = (int)(t % KW);
        t /= KW;
        const int ky = (int)(t % KH);
        t /= KH;
        const int c = (int)(t % C);
        const size_t b = t / C;
        out[i] = x[((b * C + c) * H + oy + ky) * (size_t)W + ox + kx];
    }
}

template <int KH, int KW, int CPT>
int launch64(hipStream_t s, ConvParams64 p, bool unpool) {
    p.n_ptiles = (p.P + BN - 1) / BN;
    p.n_mtiles = p.Mpad / BM;
    const int grid = p.n_ptiles * p.n_mtiles;
    if (unpool)
        IISEG_LAUNCH((conv_taps_f64_kernel<KH, KW, CPT, true>), dim3(grid), dim3(256), 0, s, p);
    else
        IISEG_LAUNCH((conv_taps_f64_kernel<KH, KW, CPT, false>), dim3(grid), dim3(256), 0, s, p);
    return iiseg_check_launch();
}

int cpt64(int KH, int KW) { return (KH == 3 && KW == 3) ? 4 : ((KH == 1 && KW == 1) ? 16 : 0); }

int check64(const iiseg_conv_desc* d) {
    if (!d) return IISEG_ERR_NULL;
    if (d->B <= 0 || d->C1 <= 0 || d->C2 < 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 ||
        d->pad < 0 || d->dil <= 0 || d->OH <= 0 || d->OW <= 0 || d->oy0 < 0 || d->ox0 < 0)
        return IISEG_ERR_SHAPE;
    const int cpt = cpt64(d->KH, d->KW);
    if (cpt == 0) return IISEG_ERR_UNSUPPORTED;
    int fullH = d->H + 2 * d->pad - d->dil * (d->KH - 1);
    int fullW = d->W + 2 * d->pad - d->dil * (d->KW - 1);
    if (d->flags & IISEG_CONV_TRANSPOSED2) {
        if (d->KH != 3 || d->KW != 3 || (d->flags & IISEG_CONV_UNPOOL) || d->C2 != 0)
            return IISEG_ERR_UNSUPPORTED;
        fullH = 2 * d->H + 1;
        fullW = 2 * d->W + 1;
    }
    if (fullH <= 0 || fullW <= 0 || d->oy0 + d->OH > fullH || d->ox0 + d->OW > fullW)
        return IISEG_ERR_SHAPE;
    if (d->out_ctot != 0 && (d->out_c0 < 0 || d->out_c0 + d->Cout > d->out_ctot))
        return IISEG_ERR_SHAPE;
    if (d->out_H != 0 && (d->out_y0 < 0 || d->out_x0 < 0 || d->out_y0 + d->OH > d->out_H ||
                          d->out_x0 + d->OW > d->out_W))
        return IISEG_ERR_SHAPE;
    const int C = d->C1 + d->C2, T = d->KH * d->KW;
    if (d->Kpad != (C + cpt - 1) / cpt * cpt * T || d->Mpad != (d->Cout + BM - 1) / BM * BM)
        return IISEG_ERR_SHAPE;
    if ((int64_t)d->B * d->OH * d->OW >= (1ll << 31) - 512) return IISEG_ERR_SHAPE;
    const int64_t span = BN / ((int64_t)d->OH * d->OW) + 2;
    const int64_t cmax = d->C1 > d->C2 ? d->C1 : d->C2;
    if (span * cmax * d->H * d->W * 8 >= (1ll << 31) - (1 << 20)) return IISEG_ERR_SHAPE;
    return IISEG_OK;
}

// launch parameters of a validated request (pointers left to the caller)
ConvParams64 params64(const iiseg_conv_desc* d) {
    ConvParams64 p = {};
    p.B = d->B; p.C1 = d->C1; p.C2 = d->C2; p.H = d->H; p.W = d->W;
    p.h2 = d->H / 2; p.w2 = d->W / 2;
    p.Cout = d->Cout; p.OH = d->OH; p.OW = d->OW; p.oy0 = d->oy0; p.ox0 = d->ox0;
    p.AH = d->AH; p.AW = d->AW; p.ay0 = d->ay0; p.ax0 = d->ax0;
    p.Kpad = d->Kpad; p.Mpad = d->Mpad; p.pad = d->pad; p.dil = d->dil;
    p.P = d->B * d->OH * d->OW;
    p.n_ptiles = p.n_mtiles = 0;
    p.relu = (d->flags & IISEG_CONV_RELU) ? 1 : 0;
    p.out_ctot = d->out_ctot ? d->out_ctot : d->Cout;
    p.out_c0 = d->out_ctot ? d->out_c0 : 0;
    p.transposed = (d->flags & IISEG_CONV_TRANSPOSED2) ? 1 : 0;
    p.out_H = d->out_H ? d->out_H : d->OH;
    p.out_W = d->out_H ? d->out_W : d->OW;
    p.out_y0 = d->out_H ? d->out_y0 : 0;
    p.out_x0 = d->out_H ? d->out_x0 : 0;
    return p;
}

}  // namespace

extern "C" int iiseg_conv_halo_f64_supported(const iiseg_conv_desc* d) {
    if (check64(d) != IISEG_OK) return 0;
    return iiseg_conv_halo_f64_ok(params64(d), d->KH, d->KW) ? 1 : 0;
}

extern "C" int iiseg_conv_plan_f64(iiseg_conv_desc* d) {
    if (!d) return IISEG_ERR_NULL;
    if (d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->C1 <= 0 || d->C2 < 0) return IISEG_ERR_SHAPE;
    const int cpt = cpt64(d->KH, d->KW);
    if (cpt == 0) return IISEG_ERR_UNSUPPORTED;
    d->Kpad = (d->C1 + d->C2 + cpt - 1) / cpt * cpt * d->KH * d->KW;
    d->Mpad = (d->Cout + BM - 1) / BM * BM;
    return IISEG_OK;
}

extern "C" int iiseg_conv_pack_f64(void* stream, const iiseg_conv_desc* d, const double* w,
                                   int64_t stride_o, int64_t stride_c, double* wp) {
    int st = check64(d);
    if (st) return st;
    if (!w || !wp) return IISEG_ERR_NULL;
    if ((uintptr_t)wp & 15) return IISEG_ERR_ALIGN;
    const int K = (d->C1 + d->C2) * d->KH * d->KW;
    const int64_t n = (int64_t)d->Kpad * d->Mpad;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    IISEG_LAUNCH(conv_pack_f64_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w,
                       stride_o, stride_c, wp, d->KH * d->KW, d->Cout, K, d->Kpad, d->Mpad,
                       (d->flags & IISEG_CONV_TRANSPOSED2) ? 1 : 0);
    return iiseg_check_launch();
}

// 1 when iiseg_conv_pool_f64 can fuse the 2x2 max-pool into this (planned) request: a halo-kernel layer with
// more than 16 output channels, whole pooling windows (even origin; an odd extent only where the window ends
// at the map's last, unpaired row / column), no skip-add
extern "C" int iiseg_conv_pool_f64_supported(const iiseg_conv_desc* d) {
    if (check64(d) != IISEG_OK) return 0;
    if (d->Cout <= 16 || !iiseg_conv_halo_f64_ok(params64(d), d->KH, d->KW)) return 0;
    const int fullH = d->H + 2 * d->pad - 2, fullW = d->W + 2 * d->pad - 2;
    if ((d->oy0 | d->ox0) & 1) return 0;
    if ((d->OH & 1) && d->oy0 + d->OH != fullH) return 0;
    if ((d->OW & 1) && d->ox0 + d->OW != fullW) return 0;
    return 1;
}

extern "C" int iiseg_conv_f64(void* stream, const iiseg_conv_desc* d, const double* x1,
                              const double* x2, const double* pre, const double* pooled,
                              const double* wp, const double* bias, const double* add, double* out) {
    return iiseg_conv_pool_f64(stream, d, x1, x2, pre, pooled, wp, bias, add, out, nullptr);
}

extern "C" int iiseg_conv_pool_f64(void* stream, const iiseg_conv_desc* d, const double* x1,
                                   const double* x2, const double* pre, const double* pooled,
                                   const double* wp, const double* bias, const double* add, double* out,
                                   double* pool_out) {
    return iiseg_conv_mask_f64(stream, d, x1, x2, pre, pooled, nullptr, wp, bias, add, out, pool_out, nullptr);
}

// DePool2D masks as bytes on the halo-tile kernel: any plain 3x3 request it runs
extern "C" int iiseg_conv_mask_f64_supported(const iiseg_conv_desc* d) {
    if (check64(d) != IISEG_OK) return 0;
    return iiseg_conv_halo_f64_ok(params64(d), d->KH, d->KW) ? 1 : 0;
}

extern "C" int iiseg_conv_mask_f64(void* stream, const iiseg_conv_desc* d, const double* x1,
                                   const double* x2, const double* pre, const double* pooled,
                                   const uint8_t* mask_in, const double* wp, const double* bias,
                                   const double* add, double* out, double* pool_out, uint8_t* mask_out) {
    int st = check64(d);
    if (st) return st;
    if (!x1 || !wp) return IISEG_ERR_NULL;
    if (!out && !(pool_out && mask_out)) return IISEG_ERR_NULL;     // the pre-pool map may be skipped
    if (d->C2 > 0 && !x2) return IISEG_ERR_NULL;
    if ((uintptr_t)wp & 15) return IISEG_ERR_ALIGN;
    const bool unpool = (d->flags & IISEG_CONV_UNPOOL) != 0;
    if ((mask_in || mask_out) && !iiseg_conv_mask_f64_supported(d)) return IISEG_ERR_UNSUPPORTED;
    if (mask_out && !pool_out) return IISEG_ERR_UNSUPPORTED;
    if (mask_in && !unpool) return IISEG_ERR_UNSUPPORTED;
    if (unpool && !mask_in && (!pre || !pooled)) return IISEG_ERR_NULL;
    if (unpool && d->C2 != 0) return IISEG_ERR_UNSUPPORTED;
    if (add && (d->AH < d->ay0 + d->OH || d->AW < d->ax0 + d->OW || d->ay0 < 0 || d->ax0 < 0))
        return IISEG_ERR_SHAPE;
    if (pool_out && (add || !iiseg_conv_pool_f64_supported(d))) return IISEG_ERR_UNSUPPORTED;
    ConvParams64 p = params64(d);
    p.x1 = x1; p.x2 = x2; p.pre = pre; p.pooled = pooled; p.wp = wp; p.bias = bias; p.add = add;
    p.out = out;
    p.pool = pool_out;
    p.mask_in = mask_in; p.mask_out = mask_out;
    p.pool_H = (d->H + 2 * d->pad - d->dil * (d->KH - 1)) / 2;
    p.pool_W = (d->W + 2 * d->pad - d->dil * (d->KW - 1)) / 2;
    hipStream_t s = (hipStream_t)stream;
    // plain 3x3 layers: the halo-tile kernel (conv_halo_f64.hip), bit-identical to the static-tap one
    if (iiseg_conv_halo_f64_ok(p, d->KH, d->KW)) return iiseg_launch_conv_halo_f64(s, p, unpool);
    // (the static-tap kernel knows neither mask bytes nor the fused pool: a request that carries them and is
    // refused by the halo kernel only on the limits that depend on the pointers set above -- the size of the
    // skip-add tensor -- must not fall through to it)
    if (mask_in || mask_out || pool_out) return IISEG_ERR_UNSUPPORTED;
    if (d->KH == 3) return launch64<3, 3, 4>(s, p, unpool);
    return launch64<1, 1, 16>(s, p, unpool);
}

extern "C" int iiseg_im2col_f64(void* stream, const double* x, double* out, int32_t B, int32_t C,
                                int32_t H, int32_t W, int32_t KH, int32_t KW) {
    if (!x || !out) return IISEG_ERR_NULL;
    if (B <= 0 || C <= 0 || KH <= 0 || KW <= 0 || H < KH || W < KW) return IISEG_ERR_SHAPE;
    const int OH = H - KH + 1, OW = W - KW + 1;
    const size_t n = (size_t)B * C * KH * KW * OH * OW;
    size_t g = (n + 255) / 256;
    if (g > 16384) g = 16384;
    IISEG_LAUNCH(im2col_f64_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, out, B,
                       C, H, W, KH, KW, OH, OW);
    return iiseg_check_launch();
}
