// Implicit-GEMM convolution for gfx950 (CDNA4), fp32 in / fp32 accumulate on the matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, a k-ordered fmaf chain per output element).
//
// GEMM view:  D[co][p] = sum_k Wp[k][co] * X[k][p],  k = (c, ky, kx),  p = (b, oy, ox).
// Output channels are the MFMA row index and pixels the column index, so that one accumulator
// register of a wave covers 32 consecutive pixels of one channel: 128-byte NCHW stores.
//
// X is never materialised: each workgroup gathers its [BK][BN] slice straight from the NCHW
// activation(s) through a per-layer table (one int4 per k: element offset, dy, dx, channel /
// source id), which makes kernel size, padding, dilation, the two-source h-concat and the
// equality-mask unpool (DePool2D) properties of the gather, not of the kernel.
//
// Replaces Theano CorrMM behind Lasagne Conv2DLayer (reference models/fcn8.py:34-85,
// models/fcn_down.py:102-104, models/fcn_up.py:83-86); see include/iiseg.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "iiseg.h"
#include "common.h"
#include "conv_common.h"

using namespace iiseg;

// conv_taps.hip
int iiseg_taps_cpt(int KH, int KW);
int iiseg_launch_conv_taps(hipStream_t s, const ConvParams& p, int KH, int KW, int bm, bool unpool);
// conv_halo.hip: halo-tile direct 3x3 kernel
bool iiseg_conv_halo_ok(const ConvParams& p, int KH, int KW);
int iiseg_launch_conv_halo(hipStream_t s, const ConvParams& p, int bm, bool unpool);
// conv_small.hip: layers between at most 16 channels on either side, on the vector ALU
bool iiseg_conv_small_ok(const ConvParams& p, int KH, int KW);
int iiseg_launch_conv_small(hipStream_t s, const ConvParams& p, int KH);

namespace {

constexpr int BK = 16;  // k-tile depth: 8 MFMA k-steps

// N x 64 bytes of the gather table (4 entries each) into SGPRs with wide scalar loads.  Inline
// asm so that the loads stay wide, unconditional and ahead of the address arithmetic; the
// wait is inside the statement (hipcc does not track asm loads).
template <int N>
__device__ __forceinline__ void load_ktab(const int4* tab, i32x16 (&t)[N]) {
    static_assert(N == 2 || N == 4, "8 or 16 rows per wave");
    if constexpr (N == 2)
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=s"(t[0]), "=s"(t[1]) : "s"(tab) : "memory");
    else
        asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\t"
                     "s_load_dwordx16 %2, %4, 0x80\n\ts_load_dwordx16 %3, %4, 0xc0\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=s"(t[0]), "=s"(t[1]), "=s"(t[2]), "=s"(t[3]) : "s"(tab) : "memory");
}

// scalar (wave-uniform) base + 32-bit unsigned per-lane byte offset: global_load saddr form
__device__ __forceinline__ float ld_off(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

template <int BM, int BN, int WM, int WN, bool UNPOOL>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvParams p) {
    constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile (channels x pixels)
    constexpr int TM = WTM / 32, TN = WTN / 32;  // 32x32 MFMA tiles per wave
    constexpr int RG = 256 / BN;                 // gather row groups
    constexpr int XROWS = BK / RG;               // gathered elements per thread per k-tile
    constexpr int WVEC = BK * BM / 4;            // float4 per weight tile
    constexpr int WPT = (WVEC + 255) / 256;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(BN >= 64 && 256 % BN == 0, "row group must be wave-uniform");

    __shared__ __attribute__((aligned(16))) float Ws[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Xs[2][BK][BN];

    int pt, mt;
    tile_of_block(blockIdx.x, gridDim.x, p.n_ptiles, p.n_mtiles, pt, mt);
    const int p0 = pt * BN, m0 = mt * BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int OHW = p.OH * p.OW, HW = p.H * p.W;

    // ---- gather setup: this thread's pixel and row group ------------------------------
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
    const int pixoff = goy * p.W + gox;  // may address outside the image; guarded per tap
    // Addressing: one wave-uniform (SGPR) base per source = the tensor at the first image this
    // tile touches, plus a 32-bit per-lane element offset (the tile spans few images, so the
    // offset stays far below 2^31): global_load with scalar base + vector offset, one VALU add
    // per gathered element.
    const int b0 = __builtin_amdgcn_readfirstlane(p0 / OHW);  // first image of the tile
    const int db = gb - b0;                        // 0..few
    const float* sx1;
    const float* sx2 = nullptr;
    const float* spre = nullptr;
    const float* spool = nullptr;
    unsigned lo1, lo2 = 0;                         // per-lane BYTE offsets of this pixel's image
    if constexpr (UNPOOL) {
        const size_t hw2 = (size_t)p.h2 * p.w2;
        sx1 = p.x1 + (size_t)b0 * p.C1 * hw2;      // up
        spool = p.pooled + (size_t)b0 * p.C1 * hw2;
        spre = p.pre + (size_t)b0 * p.C1 * HW;
        lo1 = 4u * (unsigned)(db * p.C1 * HW + pixoff); // into pre
        lo2 = 4u * (unsigned)(db * p.C1 * (p.h2 * p.w2));  // into up / pooled
    } else {
        sx1 = p.x1 + (size_t)b0 * p.C1 * HW;
        sx2 = p.x2 ? p.x2 + (size_t)b0 * p.C2 * HW : p.x1;
        lo1 = 4u * (unsigned)(db * p.C1 * HW + pixoff);
        lo2 = 4u * (unsigned)(db * p.C2 * HW + pixoff);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Register staging of the next k-tile (global -> regs while the MFMAs of the current tile
    // run, regs -> LDS afterwards).  Row group `rg` owns the XROWS consecutive k-rows
    // [rg*XROWS, (rg+1)*XROWS) of a tile, so its gather-table entries are one contiguous,
    // wave-uniform block (wide scalar loads).
    float xv[XROWS];
    float xq[UNPOOL ? XROWS : 1];  // pooled value (unpool mode)
    float xu[UNPOOL ? XROWS : 1];  // up value
    float4 wv0 = make_float4(0.f, 0.f, 0.f, 0.f), wv1 = wv0;
    unsigned okmask = 0;  // bit j: gathered element j is inside the image
    constexpr bool W_ALL = (WVEC % 256 == 0);  // every thread stages weights
    const bool w_on = W_ALL || tid < WVEC;
    const int wrow0 = tid / (BM / 4), wc4 = tid % (BM / 4);
    const int wrow1 = (tid + 256) / (BM / 4);
    const unsigned uH = (unsigned)p.H, uW = (unsigned)p.W;
    const int pv = pvalid ? 1 : 0;

    // Branch-free gather: out-of-image taps read element 0 of their tensor (always mapped) and
    // are zeroed by a select when the tile is written to LDS.
    //
    // Schedule of one k-tile per wave (PMC showed the waves of co-resident workgroups fall into
    // lockstep, so a separate "gather phase" leaves the matrix pipe idle a third of the time):
    // the tile is cut into BK/2 chunks, chunk c = { gather GPC elements of the NEXT tile (address
    // VALU + global_load), LDS-read the MFMA operands of k-step c+1, TM*TN MFMAs of k-step c },
    // pinned in that order, so every piece of vector work issues in the shadow of an MFMA of the
    // same wave.  Gathers are issued in the first half of the chunks, their LDS stores (into the
    // idle buffer) in the second half, so only the barrier itself is left at the tile boundary.
    i32x16 tb[XROWS / 4];
#define IISEG_GATHER_ONE(J)                                                                     \
    {                                                                                           \
        constexpr int j = (J);                                                                  \
        const int4 e = make_int4(tb[j / 4][(j % 4) * 4], tb[j / 4][(j % 4) * 4 + 1],            \
                                 tb[j / 4][(j % 4) * 4 + 2], tb[j / 4][(j % 4) * 4 + 3]);       \
        const int iy = goy + e.y, ix = gox + e.z;                                               \
        int ok = pv & ((unsigned)iy < uH ? 1 : 0) & ((unsigned)ix < uW ? 1 : 0);                \
        if constexpr (UNPOOL) {                                                                 \
            /* DePool2D (layers/mylayers.py:95-114): up where pre == pooled, inside 2h x 2w */  \
            ok &= (iy < 2 * p.h2 ? 1 : 0) & (ix < 2 * p.w2 ? 1 : 0);                            \
            const unsigned offp = lo2 + 4u * (unsigned)((e.w & 0xFFFFFF) * (p.h2 * p.w2) +      \
                                                        (iy >> 1) * p.w2 + (ix >> 1));          \
            const unsigned msk = (unsigned)(-ok);                                               \
            const unsigned o1 = (lo1 + 4u * (unsigned)e.x) & msk;                               \
            const unsigned o2 = offp & msk;                                                     \
            xv[j] = ld_off(spre, o1);                                                           \
            xq[j] = ld_off(spool, o2);                                                          \
            xu[j] = ld_off(sx1, o2);                                                            \
        } else {                                                                                \
            const bool s2 = (e.w >> 30) != 0; /* wave-uniform */                                \
            const float* src = s2 ? sx2 : sx1;                                                  \
            const unsigned o = ((s2 ? lo2 : lo1) + 4u * (unsigned)e.x) & (unsigned)(-ok);       \
            xv[j] = ld_off(src, o);                                                             \
        }                                                                                       \
        okmask |= (unsigned)ok << j;                                                            \
    }
#define IISEG_LOAD_W(KT)                                                                        \
    if (w_on) {                                                                                 \
        wv0 = *reinterpret_cast<const float4*>(p.wp + (size_t)((KT) * BK + wrow0) * p.Mpad +    \
                                               m0 + wc4 * 4);                                   \
        if constexpr (WPT > 1)                                                                  \
            wv1 = *reinterpret_cast<const float4*>(p.wp + (size_t)((KT) * BK + wrow1) * p.Mpad + \
                                                   m0 + wc4 * 4);                               \
    }
#define IISEG_STORE_X(BUF, J)                                                                   \
    {                                                                                           \
        constexpr int j = (J);                                                                  \
        const bool ok = (okmask >> j) & 1u;                                                     \
        float v;                                                                                \
        if constexpr (UNPOOL)                                                                   \
            v = (ok && xv[j] == xq[j]) ? xu[j] : 0.f;                                           \
        else                                                                                    \
            v = ok ? xv[j] : 0.f;                                                               \
        Xs[BUF][rg * XROWS + j][lp] = v;                                                        \
    }
#define IISEG_STORE_W(BUF)                                                                      \
    if (w_on) {                                                                                 \
        *reinterpret_cast<float4*>(&Ws[BUF][wrow0][wc4 * 4]) = wv0;                             \
        if constexpr (WPT > 1) *reinterpret_cast<float4*>(&Ws[BUF][wrow1][wc4 * 4]) = wv1;      \
    }

    constexpr int NCH = BK / 2;                 // chunks = MFMA k-steps per tile
    constexpr int GPC = XROWS / (NCH / 2);      // gathers per chunk, first half of the chunks
    static_assert(GPC * (NCH / 2) == XROWS, "gather split");
    const int nkt = p.Kpad / BK;
    const int l31 = lane & 31, lh = lane >> 5;

    // prologue: tile 0 -> LDS buffer 0
    load_ktab<XROWS / 4>(p.ktab + rg * XROWS, tb);
    static_for<0, XROWS>([&](auto J) __attribute__((always_inline)) { IISEG_GATHER_ONE(decltype(J)::value) });
    IISEG_LOAD_W(0)
    static_for<0, XROWS>([&](auto J) __attribute__((always_inline)) { IISEG_STORE_X(0, decltype(J)::value) });
    IISEG_STORE_W(0)
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        const bool more = (kt + 1 < nkt) && !p.debug_nogather;
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = Ws[buf][lh][wm * WTM + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = Xs[buf][lh][wn * WTN + j * 32 + l31];
        if (more) {
            load_ktab<XROWS / 4>(p.ktab + (kt + 1) * BK + rg * XROWS, tb);
            okmask = 0;
        }
        static_for<0, NCH>([&](auto CH) __attribute__((always_inline)) {
            constexpr int ch = decltype(CH)::value;
            if (more) {
                if constexpr (ch < NCH / 2)
                    static_for<0, GPC>([&](auto G) __attribute__((always_inline)) {
                        IISEG_GATHER_ONE(ch * GPC + decltype(G)::value)
                    });
                if constexpr (ch == 0) IISEG_LOAD_W(kt + 1)
                // LDS stores of the next tile ride in the second half of the chunks: buffer
                // buf^1 is idle during this tile (the barrier below fences its last readers)
                if constexpr (ch >= NCH / 2)
                    static_for<0, GPC>([&](auto G) __attribute__((always_inline)) {
                        IISEG_STORE_X(buf ^ 1, (ch - NCH / 2) * GPC + decltype(G)::value)
                    });
                if constexpr (ch == NCH - 1) IISEG_STORE_W(buf ^ 1)
            }
            if constexpr (ch + 1 < NCH) {
                const int kk = (ch + 1) * 2 + lh;
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[(ch + 1) & 1][i] = Ws[buf][kk][wm * WTM + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[(ch + 1) & 1][j] = Xs[buf][kk][wn * WTN + j * 32 + l31];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ch & 1][i], b[ch & 1][j],
                                                                     acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);  // pin the chunk order
        });
        __syncthreads();
    }
#undef IISEG_GATHER_ONE
#undef IISEG_LOAD_W
#undef IISEG_STORE_X
#undef IISEG_STORE_W

    conv_epilogue<BM, BN, WM, WN>(p, acc, p0, m0, wm, wn, lane);
}

__global__ void conv_pack_kernel(const float* __restrict__ w, int64_t so, int64_t sc, float* wp,
                                 int4* ktab, int C1, int C2, int KH, int KW, int pad, int dil,
                                 int H, int W, int Cout, int K, int Kpad, int Mpad, int flip) {
    const int64_t n = (int64_t)Kpad * Mpad;
    const int KK = KH * KW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / Mpad), m = (int)(i % Mpad);
        float v = 0.f;
        if (k < K && m < Cout) {
            const int c = k / KK, r = k % KK;
            v = w[m * so + c * sc + (flip ? KK - 1 - r : r)];
        }
        wp[i] = v;
        if (m == 0) {
            int4 e;
            if (k < K) {
                const int c = k / KK, r = k % KK;
                const int ky = r / KW, kx = r % KW;
                const int dy = ky * dil - pad, dx = kx * dil - pad;
                const int src = c >= C1 ? 1 : 0;
                const int cl = src ? c - C1 : c;
                e.x = cl * H * W + dy * W + dx;
                e.y = dy;
                e.z = dx;
                e.w = cl | (src << 30);
            } else {  // padded k: always out of bounds -> contributes 0
                e.x = 0;
                e.y = -(1 << 24);
                e.z = 0;
                e.w = 0;
            }
            ktab[k] = e;
        }
    }
}

int pick_bm(int Cout) { return Cout > 64 ? 128 : (Cout > 32 ? 64 : 32); }

// channel padding of the packed weights: wide 3x3 layers are padded to 256 so that the
// 256-channel tile of conv_taps applies (128-channel kernels read the same layout)
int mpad_for(const iiseg_conv_desc* d) {
    // deep 1x1 layers (score_fr: 4096 -> 11): padded to the 128-channel tile of the split-K GEMM
    // path (iiseg_conv_gemm_f32); a handful of workgroups walking K = 4096 alone is latency-bound
    if (d->KH == 1 && d->KW == 1 && d->C1 + d->C2 >= 1024 && d->Cout < 128) return 128;
    const int bm = (d->KH == 3 && d->KW == 3 && d->Cout >= 256) ? 256 : pick_bm(d->Cout);
    return (d->Cout + bm - 1) / bm * bm;
}

// k extent of the packed weights: whole channel groups for the static-tap kernel (1x1, 3x3),
// multiples of BK for the table-driven one.
int kpad_for(const iiseg_conv_desc* d) {
    const int C = d->C1 + d->C2, T = d->KH * d->KW;
    int cpt = iiseg_taps_cpt(d->KH, d->KW);
    if (d->KH == 3 && d->KW == 3) cpt = 4;  // conv_halo's k-tile (conv_taps' 2 divides it)
    if (cpt > 0) return (C + cpt - 1) / cpt * cpt * T;
    return (C * T + BK - 1) / BK * BK;
}

}  // namespace

extern "C" int iiseg_conv_plan(iiseg_conv_desc* d) {
    if (!d) return IISEG_ERR_NULL;
    if (d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->C1 <= 0 || d->C2 < 0) return IISEG_ERR_SHAPE;
    const int bm = pick_bm(d->Cout);
    d->Kpad = kpad_for(d);
    d->Mpad = mpad_for(d);
    return IISEG_OK;
}

extern "C" int iiseg_conv_ktab_entries(const iiseg_conv_desc* d) { return d ? d->Kpad : IISEG_ERR_NULL; }

static int check_desc(const iiseg_conv_desc* d) {
    if (!d) return IISEG_ERR_NULL;
    if (d->B <= 0 || d->C1 <= 0 || d->C2 < 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 ||
        d->KH <= 0 || d->KW <= 0 || d->pad < 0 || d->dil <= 0 || d->OH <= 0 || d->OW <= 0 ||
        d->oy0 < 0 || d->ox0 < 0)
        return IISEG_ERR_SHAPE;
    int fullH = d->H + 2 * d->pad - d->dil * (d->KH - 1);
    int fullW = d->W + 2 * d->pad - d->dil * (d->KW - 1);
    if (d->flags & IISEG_CONV_TRANSPOSED2) {
        if (!((d->KH == 3 && d->KW == 3) || (d->KH == 4 && d->KW == 4)) ||
            (d->flags & IISEG_CONV_UNPOOL) || d->C2 != 0)
            return IISEG_ERR_UNSUPPORTED;
        fullH = (d->H - 1) * 2 + d->KH;
        fullW = (d->W - 1) * 2 + d->KW;
    }
    if (fullH <= 0 || fullW <= 0 || d->oy0 + d->OH > fullH || d->ox0 + d->OW > fullW)
        return IISEG_ERR_SHAPE;
    if (d->out_ctot != 0 && (d->out_c0 < 0 || d->out_c0 + d->Cout > d->out_ctot))
        return IISEG_ERR_SHAPE;
    if (d->out_H != 0 && (d->out_y0 < 0 || d->out_x0 < 0 || d->out_y0 + d->OH > d->out_H ||
                          d->out_x0 + d->OW > d->out_W))
        return IISEG_ERR_SHAPE;
    const int bm = pick_bm(d->Cout);
    if (d->Kpad != kpad_for(d) || d->Mpad != mpad_for(d)) return IISEG_ERR_SHAPE;
    // int32 index ranges used by the kernel: pixel index, and the per-tile relative BYTE offsets
    // (a 256-pixel tile touches at most 256/(OH*OW) + 2 images)
    if ((int64_t)d->B * d->OH * d->OW >= (1ll << 31) - 512) return IISEG_ERR_SHAPE;
    const int64_t span = 256 / ((int64_t)d->OH * d->OW) + 2;
    const int64_t cmax = d->C1 > d->C2 ? d->C1 : d->C2;
    if (span * cmax * d->H * d->W * 4 >= (1ll << 32) - (1 << 20)) return IISEG_ERR_SHAPE;
    if (d->C1 >= (1 << 24) || d->C2 >= (1 << 24)) return IISEG_ERR_SHAPE;
    return IISEG_OK;
}

extern "C" int iiseg_conv_pack_f32(void* stream, const iiseg_conv_desc* d, const float* w,
                                   int64_t stride_o, int64_t stride_c, float* wp, int32_t* ktab) {
    int st = check_desc(d);
    if (st) return st;
    if (!w || !wp || !ktab) return IISEG_ERR_NULL;
    if (((uintptr_t)wp & 15) || ((uintptr_t)ktab & 15)) return IISEG_ERR_ALIGN;
    const int K = (d->C1 + d->C2) * d->KH * d->KW;
    const int64_t n = (int64_t)d->Kpad * d->Mpad;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    IISEG_LAUNCH(conv_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, stride_o,
                       stride_c, wp, reinterpret_cast<int4*>(ktab), d->C1, d->C2, d->KH, d->KW,
                       d->pad, d->dil, d->H, d->W, d->Cout, K, d->Kpad, d->Mpad,
                       (d->flags & IISEG_CONV_TRANSPOSED2) ? 1 : 0);
    return iiseg_check_launch();
}

template <int BM, int BN, int WM, int WN>
static int launch_conv(hipStream_t s, const ConvParams& cp, bool unpool) {
    ConvParams p = cp;
    p.n_ptiles = (p.P + BN - 1) / BN;
    p.n_mtiles = p.Mpad / BM;
    const int grid = p.n_ptiles * p.n_mtiles;
    static const int dyn = getenv("IISEG_DEBUG_DYNLDS") ? atoi(getenv("IISEG_DEBUG_DYNLDS")) : 0;
    static const int nog = getenv("IISEG_DEBUG_NOGATHER") ? 1 : 0;
    p.debug_nogather = nog;
    if (unpool)
        IISEG_LAUNCH((conv_igemm_f32_kernel<BM, BN, WM, WN, true>), dim3(grid), dim3(256), dyn, s, p);
    else
        IISEG_LAUNCH((conv_igemm_f32_kernel<BM, BN, WM, WN, false>), dim3(grid), dim3(256), dyn, s, p);
    return iiseg_check_launch();
}

extern "C" int iiseg_conv_f32(void* stream, const iiseg_conv_desc* d, const float* x1,
                              const float* x2, const float* pre, const float* pooled,
                              const float* wp, const int32_t* ktab, const float* bias,
                              const float* add, float* out) {
    return iiseg_conv_pool_f32(stream, d, x1, x2, pre, pooled, wp, ktab, bias, add, out, nullptr);
}

extern "C" int iiseg_conv_small_supported(const iiseg_conv_desc* d) {
    if (!d || check_desc(d) || (d->flags & (IISEG_CONV_UNPOOL | IISEG_CONV_TRANSPOSED2))) return 0;
    ConvParams p = {};
    p.B = d->B; p.C1 = d->C1; p.C2 = d->C2; p.H = d->H; p.W = d->W;
    p.Cout = d->Cout; p.OH = d->OH; p.OW = d->OW; p.Mpad = d->Mpad; p.pad = d->pad;
    p.out = reinterpret_cast<float*>(16);         // (only tested for presence)
    return iiseg_conv_small_ok(p, d->KH, d->KW) ? 1 : 0;
}

extern "C" int iiseg_conv_pool_supported(const iiseg_conv_desc* d) {
    if (!d || check_desc(d)) return 0;
    static const int halo = getenv("IISEG_CONV_HALO") ? atoi(getenv("IISEG_CONV_HALO")) : 1;
    if (!halo || d->KH != 3 || d->KW != 3 || d->dil != 1 || (d->flags & IISEG_CONV_TRANSPOSED2) ||
        d->Cout >= 256 || d->Cout <= 16 || d->Kpad % 36 || (d->C2 > 0 && d->C1 % 4))
        return 0;
    // pooling windows must be whole inside the computed window: even origin, and an even extent
    // unless the window ends at the map's last (unpaired) row / column
    const int fullH = d->H + 2 * d->pad - 2, fullW = d->W + 2 * d->pad - 2;
    if ((d->oy0 | d->ox0) & 1) return 0;
    if ((d->OH & 1) && d->oy0 + d->OH != fullH) return 0;
    if ((d->OW & 1) && d->ox0 + d->OW != fullW) return 0;
    return 1;
}

struct BnIn {   // fused input BatchNorm + ReLU (conv_halo16 only)
    const float *beta, *gamma, *mean, *inv_std;
    int64_t bstride;
};
static int conv_run(void* stream, const iiseg_conv_desc* d, const float* x1, const float* x2,
                    const float* pre, const float* pooled, const float* wp, const int32_t* ktab,
                    const float* bias, const float* add, float* out, float* pool_out,
                    const BnIn* bn, const unsigned char* mask_in = nullptr,
                    unsigned char* mask_out = nullptr);

extern "C" int iiseg_conv_mask_supported(const iiseg_conv_desc* d) {
    if (!d || check_desc(d)) return 0;
    static const int halo = getenv("IISEG_CONV_HALO") ? atoi(getenv("IISEG_CONV_HALO")) : 1;
    return halo && d->KH == 3 && d->KW == 3 && d->dil == 1 && !(d->flags & IISEG_CONV_TRANSPOSED2) &&
           d->Cout < 256 && d->Kpad % 36 == 0 && !(d->C2 > 0 && d->C1 % 4);
}

extern "C" int iiseg_conv_mask_f32(void* stream, const iiseg_conv_desc* d, const float* x1,
                                   const float* x2, const float* pre, const float* pooled,
                                   const uint8_t* mask_in, const float* wp, const int32_t* ktab,
                                   const float* bias, const float* add, float* out, float* pool_out,
                                   uint8_t* mask_out) {
    if (!iiseg_conv_mask_supported(d)) return IISEG_ERR_UNSUPPORTED;
    if (mask_out && !pool_out) return IISEG_ERR_UNSUPPORTED;
    if (mask_in && !(d->flags & IISEG_CONV_UNPOOL)) return IISEG_ERR_UNSUPPORTED;
    return conv_run(stream, d, x1, x2, pre, pooled, wp, ktab, bias, add, out, pool_out, nullptr,
                    mask_in, mask_out);
}

extern "C" int iiseg_conv_pool_f32(void* stream, const iiseg_conv_desc* d, const float* x1,
                                   const float* x2, const float* pre, const float* pooled,
                                   const float* wp, const int32_t* ktab, const float* bias,
                                   const float* add, float* out, float* pool_out) {
    return conv_run(stream, d, x1, x2, pre, pooled, wp, ktab, bias, add, out, pool_out, nullptr);
}

extern "C" int iiseg_conv_bnrelu_supported(const iiseg_conv_desc* d) {
    if (!d || check_desc(d)) return 0;
    static const int halo = getenv("IISEG_CONV_HALO") ? atoi(getenv("IISEG_CONV_HALO")) : 1;
    return halo && d->KH == 3 && d->KW == 3 && d->dil == 1 && d->Cout <= 16 && d->C2 == 0 &&
           !(d->flags & (IISEG_CONV_UNPOOL | IISEG_CONV_TRANSPOSED2)) && d->Kpad % 36 == 0;
}

extern "C" int iiseg_conv_bnrelu_f32(void* stream, const iiseg_conv_desc* d, const float* x,
                                     int64_t x_bstride, const float* beta, const float* gamma,
                                     const float* mean, const float* inv_std, const float* wp,
                                     const int32_t* ktab, const float* bias, float* out) {
    if (!beta || !gamma || !mean || !inv_std) return IISEG_ERR_NULL;
    if (!iiseg_conv_bnrelu_supported(d)) return IISEG_ERR_UNSUPPORTED;
    if (x_bstride < (int64_t)d->C1 * d->H * d->W) return IISEG_ERR_SHAPE;
    const BnIn bn = {beta, gamma, mean, inv_std, x_bstride};
    return conv_run(stream, d, x, nullptr, nullptr, nullptr, wp, ktab, bias, nullptr, out, nullptr, &bn);
}

static int conv_run(void* stream, const iiseg_conv_desc* d, const float* x1, const float* x2,
                    const float* pre, const float* pooled, const float* wp, const int32_t* ktab,
                    const float* bias, const float* add, float* out, float* pool_out,
                    const BnIn* bn, const unsigned char* mask_in, unsigned char* mask_out) {
    int st = check_desc(d);
    if (pool_out && !iiseg_conv_pool_supported(d)) return IISEG_ERR_UNSUPPORTED;
    if (st) return st;
    if (!x1 || !wp || !ktab) return IISEG_ERR_NULL;
    if (!out && !(pool_out && mask_out)) return IISEG_ERR_NULL;   // pre-pool map may be skipped
    if (d->C2 > 0 && !x2) return IISEG_ERR_NULL;
    if (((uintptr_t)wp & 15) || ((uintptr_t)ktab & 15)) return IISEG_ERR_ALIGN;
    const bool unpool = (d->flags & IISEG_CONV_UNPOOL) != 0;
    if (unpool && !mask_in && (!pre || !pooled)) return IISEG_ERR_NULL;
    if (unpool && d->C2 != 0) return IISEG_ERR_UNSUPPORTED;
    if (add && (d->AH < d->ay0 + d->OH || d->AW < d->ax0 + d->OW || d->ay0 < 0 || d->ax0 < 0))
        return IISEG_ERR_SHAPE;

    ConvParams p;
    p.x1 = x1; p.x2 = x2; p.pre = pre; p.pooled = pooled; p.wp = wp;
    p.ktab = reinterpret_cast<const int4*>(ktab);
    p.bias = bias; p.add = add; p.out = out;
    p.B = d->B; p.C1 = d->C1; p.C2 = d->C2; p.H = d->H; p.W = d->W;
    p.h2 = d->H / 2; p.w2 = d->W / 2;
    p.Cout = d->Cout; p.OH = d->OH; p.OW = d->OW; p.oy0 = d->oy0; p.ox0 = d->ox0;
    p.AH = d->AH; p.AW = d->AW; p.ay0 = d->ay0; p.ax0 = d->ax0;
    p.Kpad = d->Kpad; p.Mpad = d->Mpad;
    p.pad = d->pad; p.dil = d->dil;
    p.debug_nogather = 0;
    p.pool = pool_out;
    p.mask_in = mask_in; p.mask_out = mask_out;
    p.bn_beta = bn ? bn->beta : nullptr;
    p.bn_gamma = bn ? bn->gamma : nullptr;
    p.bn_mean = bn ? bn->mean : nullptr;
    p.bn_inv_std = bn ? bn->inv_std : nullptr;
    p.in_bstride = bn ? bn->bstride : 0;
    p.pool_H = (d->H + 2 * d->pad - d->dil * (d->KH - 1)) / 2;
    p.pool_W = (d->W + 2 * d->pad - d->dil * (d->KW - 1)) / 2;
    p.out_ctot = d->out_ctot ? d->out_ctot : d->Cout;
    p.out_c0 = d->out_ctot ? d->out_c0 : 0;
    p.transposed = (d->flags & IISEG_CONV_TRANSPOSED2) ? 1 : 0;
    p.out_H = d->out_H ? d->out_H : d->OH;
    p.out_W = d->out_H ? d->out_W : d->OW;
    p.out_y0 = d->out_H ? d->out_y0 : 0;
    p.out_x0 = d->out_H ? d->out_x0 : 0;
    p.P = d->B * d->OH * d->OW;
    p.n_ptiles = p.n_mtiles = 0;
    p.relu = (d->flags & IISEG_CONV_RELU) ? 1 : 0;

    hipStream_t s = (hipStream_t)stream;
    // IISEG_CONV_HALO: 0 = never, 1 = 3x3 layers with Cout < 256 (default), 2 = every 3x3 layer
    static const int halo = getenv("IISEG_CONV_HALO") ? atoi(getenv("IISEG_CONV_HALO")) : 1;
    const bool use_halo = halo && (halo > 1 || d->Cout < 256) && iiseg_conv_halo_ok(p, d->KH, d->KW);
    if (pool_out && (!use_halo || add || d->Cout <= 16)) return IISEG_ERR_UNSUPPORTED;
    if (bn && !use_halo) return IISEG_ERR_UNSUPPORTED;
    if ((mask_in || mask_out) && !use_halo) return IISEG_ERR_UNSUPPORTED;
    // tiny layers (context module: 11 -> 11, dilated): HBM-bound, half of an MFMA tile would be padding
    if (!unpool && !bn && !pool_out && iiseg_conv_small_ok(p, d->KH, d->KW))
        return iiseg_launch_conv_small(s, p, d->KH);
    if (use_halo) return iiseg_launch_conv_halo(s, p, pick_bm(d->Cout), unpool);
    if (iiseg_taps_cpt(d->KH, d->KW) > 0)
        return iiseg_launch_conv_taps(s, p, d->KH, d->KW, pick_bm(d->Cout), unpool);
    switch (pick_bm(d->Cout)) {
        case 128: return launch_conv<128, 128, 2, 2>(s, p, unpool);
        case 64: return launch_conv<64, 256, 1, 4>(s, p, unpool);
        default: return launch_conv<32, 256, 1, 4>(s, p, unpool);
    }
}
