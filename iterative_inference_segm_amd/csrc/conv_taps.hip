// Static-tap implicit-GEMM convolution for gfx950 (CDNA4): 1x1 and 3x3 (any pad, any dilation),
// stride 1, fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32.  This is the kernel that runs
// >97 % of the FLOPs of the hot path; conv_igemm.hip keeps the table-driven gather for every
// other filter shape (7x7 fc6).
//
// Same GEMM view as conv_igemm.hip (D[co][p] = sum_k Wp[k][co] * X[k][p], k = (c, tap)), but the
// im2col gather costs ZERO vector-ALU work per element:
//   * a k-tile is CPT whole channels x T taps, so the tap of every staged element is a
//     compile-time constant;
//   * each lane computes, once per workgroup, one byte offset per tap (T VGPRs); taps that fall
//     outside the image get an offset that is out of range of the buffer descriptor;
//   * the gather is `buffer_load_dword v, voff[tap], rsrc, soffset=channel offset`: the hardware
//     range check returns 0 for the padding, the channel stride rides in the scalar offset.
// PMC on the table-driven version showed the matrix pipe only 67-70 % busy because ~5 VALU ops
// per MFMA of address arithmetic share the issue port with the MFMAs; without the gather the
// same MFMA/LDS/barrier structure measured 138 TFLOP/s.
//
// Fusions: two-source channel concat (h first; model_helpers.py:93-94), DePool2D equality-mask
// unpool as the gather (layers/mylayers.py:88-115), bias / skip-add with center crop / ReLU /
// output window in the epilogue (fcn_up.py:96-113).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "iiseg.h"
#include "common.h"
#include "conv_common.h"

using namespace iiseg;

namespace {

// raw buffer descriptor word 3 for gfx9/CDNA: DST_SEL = XYZW, DATA_FORMAT = 32 (raw dword access)
constexpr int RSRC_W3 = 0x00027000;
constexpr unsigned OOB = 0x80000000u;  // voffset that fails every range check (buffers < 2 GiB)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const float* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, RSRC_W3);
}

__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

template <int BM, int BN, int WM, int WN, int KH, int KW, int CPT, bool UNPOOL, bool DMA>
__global__ __launch_bounds__(WM * WN * 64, (UNPOOL || BN > 128) ? 2 : 4) void conv_taps_f32_kernel(const ConvParams p) {
    constexpr int T = KH * KW;                   // taps
    constexpr int BK = CPT * T;                  // k-tile depth (whole channels)
    constexpr int NCH = BK / 2;                  // MFMA k-steps per tile
    constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile (channels x pixels)
    constexpr int TM = WTM / 32, TN = WTN / 32;  // 32x32 MFMA tiles per wave
    constexpr int NW = WM * WN;                  // waves: 4, or 8 (256-channel tiles: waves 0-3
    constexpr int NT = NW * 64;                  // stage X, waves 4-7 stage W, all 8 compute)
    constexpr int RG = 256 / BN;                 // staging row groups
    constexpr int CPG = CPT / RG;                // channels staged per thread per tile
    constexpr int XE = CPG * T;                  // staged elements per thread per tile
    constexpr int WVEC = BK * BM / 4;            // float4 per weight tile
    constexpr int WPT = (WVEC + 255) / 256;
    static_assert((NW == 4 || NW == 8) && BK % 2 == 0 && CPT % RG == 0, "tile config");
    static_assert(WPT <= 5, "weight staging registers");
    static_assert(!(DMA && UNPOOL), "LDS-DMA staging is for the plain gather");

    __shared__ __attribute__((aligned(16))) float Ws[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Xs[2][BK][BN];

    int pt, mt;
    tile_of_block(blockIdx.x, gridDim.x, p.n_ptiles, p.n_mtiles, pt, mt);
    const int p0 = pt * BN, m0 = mt * BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int OHW = p.OH * p.OW, HW = p.H * p.W;
    const int hw2 = p.h2 * p.w2;

    // ---- staging setup: this thread's pixel, its T tap offsets --------------------------
    const bool xstager = NW == 4 || wave < 4;           // wave-uniform staging roles
    const bool wstager = NW == 4 || wave >= NW - 4;
    const int wtid = tid - (NT - 256);                  // 0..255 on the W-staging waves
    const int lp = tid % BN;
    const int rg = __builtin_amdgcn_readfirstlane((tid & 255) / BN);
    const int lp0 = __builtin_amdgcn_readfirstlane(lp & ~63);  // first pixel column of this wave
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
    // descriptors start at the first image the tile touches (wave-uniform); a 256-pixel tile
    // spans at most 256/(OH*OW)+2 images, so per-lane byte offsets stay far below 2^31
    const int b0 = __builtin_amdgcn_readfirstlane(p0 / OHW);
    const int nb = min(p.B - b0, 256 / OHW + 2);
    const int db = gb - b0;
    const int C1 = p.C1, C2 = p.C2;

    // Buffer descriptors are built from wave-uniform (pointer, bytes) pairs right at the loads
    // (selecting between whole 128-bit descriptors makes hipcc spill them to a scratch table).
    // Channels beyond C1+C2 (k padding) are clamped to the last real channel: their packed
    // weights are zero, so whatever finite value they read contributes nothing.
    const int Ctot = C1 + C2;
    const float* base1;   // x1 (plain) / up (unpool)
    const float* base2;   // x2 (plain) / pooled (unpool)
    const float* basep = nullptr;  // pre (unpool)
    int n1, n2, np = 0;
    if constexpr (UNPOOL) {
        base1 = p.x1 + (size_t)b0 * C1 * hw2;
        base2 = p.pooled + (size_t)b0 * C1 * hw2;
        basep = p.pre + (size_t)b0 * C1 * HW;
        n1 = n2 = nb * C1 * hw2 * 4;
        np = nb * C1 * HW * 4;
    } else {
        base1 = p.x1 + (size_t)b0 * C1 * HW;
        base2 = C2 > 0 ? p.x2 + (size_t)b0 * C2 * HW : base1;
        n1 = nb * C1 * HW * 4;
        n2 = C2 > 0 ? nb * C2 * HW * 4 : n1;
    }

    // per-lane byte offsets of the T taps relative to channel 0 of the descriptor's first image;
    // voff2 = same for source 2 / pooled+up (their per-image channel counts / sizes differ)
    unsigned voff[T], voff2[T];
    static_for<0, T>([&](auto TT) __attribute__((always_inline)) {
        constexpr int t = decltype(TT)::value;
        int iy = goy + (t / KW) * p.dil - p.pad;
        int ix = gox + (t % KW) * p.dil - p.pad;
        bool par = true;
        if (p.transposed) {
            // stride-2 transposed conv: out(Y,X) += x((Y-a)/2, (X-b)/2) * Wf[a][b] where the
            // differences are even and inside the input -- just another validity pattern
            const int ty = goy - t / KW, tx = gox - t % KW;
            par = ty >= 0 && tx >= 0 && !((ty | tx) & 1);
            iy = ty >> 1;
            ix = tx >> 1;
        }
        bool ok = pvalid && par && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        voff[t] = ok ? 4u * (unsigned)(db * C1 * HW + iy * p.W + ix) : OOB;
        if constexpr (UNPOOL) {
            // DePool2D (layers/mylayers.py:95-114): only the 2h x 2w region has pooling windows
            ok = ok && iy < 2 * p.h2 && ix < 2 * p.w2;
            voff2[t] = ok ? 4u * (unsigned)(db * C1 * hw2 + (iy >> 1) * p.w2 + (ix >> 1)) : OOB;
        } else {
            voff2[t] = ok ? 4u * (unsigned)(db * C2 * HW + iy * p.W + ix) : OOB;
        }
    });

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float xv[XE];
    float xq[UNPOOL ? XE : 1];
    float xu[UNPOOL ? XE : 1];
    float4 wv0 = make_float4(0.f, 0.f, 0.f, 0.f), wv1 = wv0, wv2 = wv0, wv3 = wv0, wv4 = wv0;
    const int wrow0 = wtid / (BM / 4);  // row of this thread's j-th weight vector = wrow0 + j*RPJ
    constexpr int RPJ = 256 / (BM / 4);
    const int wc4 = wtid % (BM / 4);    // (256 % (BM/4) == 0: same column group for every j)
    const __amdgpu_buffer_rsrc_t wrsrc = mk_rsrc(p.wp, p.Kpad * p.Mpad * 4);

    // stage element j = (channel cc of this thread's group, tap t): one buffer_load, no VALU.
    // Channel -> (descriptor, scalar byte offset) is wave-uniform scalar work.
#define IISEG_GATHER(KT, J)                                                                     \
    {                                                                                           \
        constexpr int j = (J);                                                                  \
        constexpr int t = j % T;                                                                \
        const int c = min((KT) * CPT + rg * CPG + j / T, Ctot - 1);                             \
        if constexpr (UNPOOL) {                                                                 \
            xv[j] = buf_ld(mk_rsrc(basep, np), voff[t], (unsigned)(c * HW) * 4u);               \
            xq[j] = buf_ld(mk_rsrc(base2, n2), voff2[t], (unsigned)(c * hw2) * 4u);             \
            xu[j] = buf_ld(mk_rsrc(base1, n1), voff2[t], (unsigned)(c * hw2) * 4u);             \
        } else {                                                                                \
            const bool s1 = c < C1; /* wave-uniform */                                          \
            const unsigned so = (unsigned)((s1 ? c : c - C1) * HW) * 4u;                        \
            if constexpr (DMA) {                                                                \
                /* buffer_load_dword ... lds: global -> LDS without touching VGPRs; the LDS   */ \
                /* address is M0 (wave-uniform row base) + lane*4, i.e. this wave's 64 pixels */ \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                       \
                    mk_rsrc(s1 ? base1 : base2, s1 ? n1 : n2),                                  \
                    (__attribute__((address_space(3))) void*)&Xs[DMABUF][rg * XE + j][lp0],     \
                    4, (int)(s1 ? voff[t] : voff2[t]), (int)so, 0, 0);                          \
            } else {                                                                            \
                xv[j] = buf_ld(mk_rsrc(s1 ? base1 : base2, s1 ? n1 : n2),                       \
                               s1 ? voff[t] : voff2[t], so);                                    \
            }                                                                                   \
        }                                                                                       \
    }
#define IISEG_W_ON(j) \
    ((j) < WPT && wstager && (((j) + 1) * 256 <= WVEC || wtid + 256 * (j) < WVEC))
#define IISEG_W_SRC(KT, j) \
    (*reinterpret_cast<const float4*>(p.wp + (size_t)((KT) * BK + wrow0 + (j) * RPJ) * p.Mpad + m0 + wc4 * 4))
#define IISEG_W_DMA(KT, j)                                                                      \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                   \
        wrsrc, (__attribute__((address_space(3))) void*)(&Ws[DMABUF][0][0] + ((wave - (NW - 4)) * 64 + 256 * (j)) * 4), \
        16, (int)(4u * (unsigned)(((KT) * BK + wrow0 + (j) * RPJ) * p.Mpad + m0 + wc4 * 4)), 0, 0, 0)
#define IISEG_LOAD_W(KT)                                                                        \
    if constexpr (DMA) {                                                                        \
        /* weight tile global -> LDS directly: thread-linear == LDS-linear, 1 KiB per wave op */ \
        if (IISEG_W_ON(0)) IISEG_W_DMA(KT, 0);                                                  \
        if (IISEG_W_ON(1)) IISEG_W_DMA(KT, 1);                                                  \
        if (IISEG_W_ON(2)) IISEG_W_DMA(KT, 2);                                                  \
        if (IISEG_W_ON(3)) IISEG_W_DMA(KT, 3);                                                  \
        if (IISEG_W_ON(4)) IISEG_W_DMA(KT, 4);                                                  \
    } else {                                                                                    \
        if (IISEG_W_ON(0)) wv0 = IISEG_W_SRC(KT, 0);                                            \
        if (IISEG_W_ON(1)) wv1 = IISEG_W_SRC(KT, 1);                                            \
        if (IISEG_W_ON(2)) wv2 = IISEG_W_SRC(KT, 2);                                            \
        if (IISEG_W_ON(3)) wv3 = IISEG_W_SRC(KT, 3);                                            \
        if (IISEG_W_ON(4)) wv4 = IISEG_W_SRC(KT, 4);                                            \
    }
#define IISEG_STORE_TILE(BUF)                                                                   \
    {                                                                                           \
        if constexpr (!DMA) if (xstager)                                                        \
            static_for<0, XE>([&](auto JJ) __attribute__((always_inline)) {                     \
                constexpr int j = decltype(JJ)::value;                                          \
                float v = xv[j];                                                                \
                /* padding / odd trailing row+col read 0 == 0 -> up, which is also 0 there */   \
                if constexpr (UNPOOL) v = (xv[j] == xq[j]) ? xu[j] : 0.f;                       \
                Xs[BUF][rg * XE + j][lp] = v;                                                   \
            });                                                                                 \
        if constexpr (!DMA) {                                                                   \
        if (IISEG_W_ON(0)) *reinterpret_cast<float4*>(&Ws[BUF][wrow0][wc4 * 4]) = wv0;          \
        if (IISEG_W_ON(1)) *reinterpret_cast<float4*>(&Ws[BUF][wrow0 + RPJ][wc4 * 4]) = wv1;    \
        if (IISEG_W_ON(2)) *reinterpret_cast<float4*>(&Ws[BUF][wrow0 + 2 * RPJ][wc4 * 4]) = wv2; \
        if (IISEG_W_ON(3)) *reinterpret_cast<float4*>(&Ws[BUF][wrow0 + 3 * RPJ][wc4 * 4]) = wv3; \
        if (IISEG_W_ON(4)) *reinterpret_cast<float4*>(&Ws[BUF][wrow0 + 4 * RPJ][wc4 * 4]) = wv4; \
        }                                                                                       \
    }

    const int nkt = p.Kpad / BK;
    const int l31 = lane & 31, lh = lane >> 5;

    {
        constexpr int DMABUF = 0;
        if (xstager)
            static_for<0, XE>([&](auto J) __attribute__((always_inline)) { IISEG_GATHER(0, decltype(J)::value) });
        IISEG_LOAD_W(0)
    }
    IISEG_STORE_TILE(0)
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Per k-tile: NCH chunks = { stage part of the NEXT tile, LDS-read operands of k-step c+1,
    // TM*TN MFMAs of k-step c }, pinned in that order so the few remaining vector-memory /
    // LDS instructions issue in the shadow of this wave's own MFMAs.
    constexpr int GCH = NCH < 4 ? NCH : (NCH + 1) / 2;   // chunks that carry gathers
    constexpr int GPC = (XE + GCH - 1) / GCH;            // gathers per such chunk
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nkt;
        const int DMABUF = buf ^ 1;
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = Ws[buf][lh][wm * WTM + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = Xs[buf][lh][wn * WTN + j * 32 + l31];
        static_for<0, NCH>([&](auto CH) __attribute__((always_inline)) {
            constexpr int ch = decltype(CH)::value;
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
            // staging loads of the next tile go right BEHIND this chunk's MFMAs
            if (more) {
                if constexpr (ch == 0) IISEG_LOAD_W(kt + 1)
                if (xstager)
                    static_for<0, GPC>([&](auto G) __attribute__((always_inline)) {
                        constexpr int ge = ch * GPC + decltype(G)::value;
                        if constexpr (ch < GCH && ge < XE) IISEG_GATHER(kt + 1, ge)
                    });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (more) IISEG_STORE_TILE(buf ^ 1)
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef IISEG_GATHER
#undef IISEG_LOAD_W
#undef IISEG_W_DMA
#undef IISEG_W_ON
#undef IISEG_W_SRC
#undef IISEG_STORE_TILE

    conv_epilogue<BM, BN, WM, WN>(p, acc, p0, m0, wm, wn, lane);
}

template <int BM, int BN, int WM, int WN, int KH, int KW, int CPT>
int launch_taps(hipStream_t s, const ConvParams& cp, bool unpool) {
    ConvParams p = cp;
    p.n_ptiles = (p.P + BN - 1) / BN;
    p.n_mtiles = p.Mpad / BM;
    const int grid = p.n_ptiles * p.n_mtiles;
    static const int dma = getenv("IISEG_CONV_DMA") ? atoi(getenv("IISEG_CONV_DMA")) : 1;
    if (unpool)
        IISEG_LAUNCH((conv_taps_f32_kernel<BM, BN, WM, WN, KH, KW, CPT, true, false>),
                           dim3(grid), dim3(WM * WN * 64), 0, s, p);
    else if (dma)
        IISEG_LAUNCH((conv_taps_f32_kernel<BM, BN, WM, WN, KH, KW, CPT, false, true>),
                           dim3(grid), dim3(WM * WN * 64), 0, s, p);
    else
        IISEG_LAUNCH((conv_taps_f32_kernel<BM, BN, WM, WN, KH, KW, CPT, false, false>),
                           dim3(grid), dim3(WM * WN * 64), 0, s, p);
    return iiseg_check_launch();
}

}  // namespace

// Channels per k-tile of the static-tap kernel for a filter shape, 0 if it has no variant.
int iiseg_taps_cpt(int KH, int KW) {
    if (KH == 3 && KW == 3) return 2;    // BK = 18
    if (KH == 1 && KW == 1) return 16;   // BK = 16
    if (KH == 4 && KW == 4) return 2;    // BK = 32 (the 4x4/2 transposed conv of unpool 'standard')
    return 0;
}

int iiseg_launch_conv_taps(hipStream_t s, const ConvParams& p, int KH, int KW, int bm, bool unpool) {
    if (KH == 3 && KW == 3) {
        if (!unpool && p.Mpad % 256 == 0 && p.Cout >= 256)
            return launch_taps<256, 128, 4, 2, 3, 3, 2>(s, p, unpool);
        switch (bm) {
            case 128: return launch_taps<128, 128, 2, 2, 3, 3, 2>(s, p, unpool);
            case 64: return launch_taps<64, 256, 1, 4, 3, 3, 2>(s, p, unpool);
            default: return launch_taps<32, 256, 1, 4, 3, 3, 2>(s, p, unpool);
        }
    }
    if (KH == 1 && KW == 1) {
        switch (bm) {
            case 128: return launch_taps<128, 128, 2, 2, 1, 1, 16>(s, p, unpool);
            case 64: return launch_taps<64, 256, 1, 4, 1, 1, 16>(s, p, unpool);
            default: return launch_taps<32, 256, 1, 4, 1, 1, 16>(s, p, unpool);
        }
    }
    if (KH == 4 && KW == 4) {
        switch (bm) {
            case 128: return launch_taps<128, 128, 2, 2, 4, 4, 2>(s, p, unpool);
            case 64: return launch_taps<64, 128, 2, 2, 4, 4, 2>(s, p, unpool);
            default: return launch_taps<32, 128, 1, 4, 4, 4, 2>(s, p, unpool);
        }
    }
    return IISEG_ERR_UNSUPPORTED;
}
