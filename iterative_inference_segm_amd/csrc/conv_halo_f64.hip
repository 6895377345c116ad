// Halo-tile direct 3x3 convolution in float64 (v_mfma_f64_16x16x4_f64): the THIN 3x3 layers of the
// strict-parity path -- conv1_x / conv2_1 of both nets and the DAE's class-score layer -- which the
// float64 Winograd form (conv_wino_f64.hip) does not take (below 128 input channels its V / M round
// trips cost more HBM time than the direct MFMAs) and which the static-tap kernel of conv_f64.hip runs
// with one 8-byte gather per (channel, tap, pixel), a single LDS buffer and 64 MFMA rows for the 11
// output channels of the score layer (matrix pipe 46 % busy, profiles/r03_pmc.md).
//
// Same structure as conv_halo.hip: a workgroup owns a TH x 32 pixel tile of one image and BM output
// channels (BM = 64, or 16 for layers with at most 16 output channels: 11 of 16 MFMA rows used instead
// of 11 of 64).  Per k-tile (4 input channels = 36 k = 9 MFMA k-steps) the (TH+2) x 34 input patch of
// those channels is staged ONCE and the MFMA B operand is read straight out of it (element k = (c, ky,
// kx) of pixel (y, x) = patch[c][y + ky][x + kx]).  Staging is asynchronous and double buffered, one
// barrier per k-tile:
//   * plain input: dword LDS-DMA, a lane PAIR per double (inline asm, see conv_common.h dma16 for why),
//     zero padding from the buffer descriptor's range check;
//   * DePool2D input (layers/mylayers.py:88-115): a thread stages POOLED positions -- `up`, `pooled`
//     and the four `pre` values of the window, 6 loads per 4 patch elements instead of 12 -- through
//     registers, `pre == pooled ? up : 0` compared in float64, written after the k-tile's MFMAs;
//   * weights: 16-byte LDS-DMA pieces of the packed Wp[Kpad][Mpad] rows (conv_f64.hip's layout).
// The accumulation order over k (channel-major, tap-minor, four k per MFMA, sequential) is that of
// conv_taps_f64_kernel: results are bit-identical to it, on every tile, window and placement.
// Call sites: models/fcn8.py:34-45, models/fcn_down.py:102-104, models/fcn_up.py:83-86.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
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

__device__ __forceinline__ double ld64(const double* base, int bytes, unsigned voff, unsigned soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, RSRC_W3);
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}

// one dword per lane, the wave's 64 dwords contiguous at LDS byte address `lds` (wave-uniform)
__device__ __forceinline__ void dma4(i32x4s rsrc, unsigned lds, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

// MASKIN (with UNPOOL): the DePool2D mask comes as bytes (ConvParams64::mask_in, written by the pool-fused
// encoder layer of the level: bit (y & 1) * 2 + (x & 1) of the window's byte = pre == pooled, compared in
// float64 THERE) -- `up` + one byte per window instead of `up`, `pooled` and four `pre` values.
template <int BM, int TH, bool UNPOOL, bool MASKIN = false>
__global__ __launch_bounds__(256, BM == 16 ? 3 : 2) void conv_halo_f64_kernel(const ConvParams64 p,
                                                                             const int tiles_y,
                                                                             const int tiles_x) {
    constexpr int CPT = 4, BK = 9 * CPT, NS = BK / 4;
    constexpr int TW = 32, PH = TH + 2, PW = TW + 2, PP = PH * PW;
    constexpr int PE = CPT * PP;                 // doubles of a k-tile's patch
    constexpr int ND = (2 * PE + 255) / 256;     // dwords per thread (DMA form)
    constexpr int PSZ = ND * 128;                // doubles per patch buffer
    constexpr int TM = BM / 16;                  // 16-channel row tiles per wave
    constexpr int RW = TH / 4, TN = 2 * RW;      // rows per wave, 16-pixel column tiles per wave
    constexpr int WPC = BK * BM / 2;             // 16-byte weight pieces per k-tile
    constexpr int NWP = (WPC + 255) / 256;
    constexpr int WSZ = NWP * 512;               // doubles per weight buffer (whole DMA passes)
    constexpr int QH = PH / 2 + 1, QW = PW / 2 + 1, QP = QH * QW, QE = CPT * QP;
    constexpr int NQ = (QE + 255) / 256;         // pooled positions per thread (DePool2D form)
    static_assert(TH % 4 == 0 && (BM == 16 || BM == 64), "tile config");

    __shared__ __attribute__((aligned(16))) double Ws[2][WSZ];
    __shared__ __attribute__((aligned(16))) double Ps[2][PSZ];

    int pt, mt;
    tile_of_block(blockIdx.x, gridDim.x, p.n_ptiles, p.n_mtiles, pt, mt);
    const int m0 = mt * BM;
    const int tpi = tiles_y * tiles_x;
    const int b = pt / tpi;
    const int tr = pt - b * tpi;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int wy0 = ty * TH, wx0 = tx * TW;
    const int iy0 = p.oy0 + wy0 - p.pad, ix0 = p.ox0 + wx0 - p.pad;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int HW = p.H * p.W, hw2 = p.h2 * p.w2;
    const int C1 = p.C1, Ctot = p.C1 + p.C2;

    // ---- staging maps (loop-invariant) ----
    unsigned voff[UNPOOL ? 1 : ND];
    int cl[UNPOOL ? 1 : ND];
    unsigned qv[UNPOOL ? NQ : 1], pv[UNPOOL ? NQ : 1][4];
    int qs[UNPOOL ? NQ : 1][4], qc[UNPOOL ? NQ : 1];
    if constexpr (!UNPOOL) {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int w = i * 256 + tid;
            const int e = w >> 1;
            const int c = e / PP, rr = e - c * PP;
            const int py = rr / PW, px = rr - py * PW;
            const int iy = iy0 + py, ix = ix0 + px;
            const bool ok = e < PE && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            cl[i] = c;
            voff[i] = ok ? 8u * (unsigned)(c * HW + iy * p.W + ix) + 4u * (unsigned)(w & 1) : OOB;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int e = i * 256 + tid;
            const int c = e / QP, r = e - c * QP;
            const int qy = r / QW, qx = r - qy * QW;
            const int Y2 = (iy0 >> 1) + qy, X2 = (ix0 >> 1) + qx;       // arithmetic shifts: floor
            const bool in = e < QE;
            // outside the h2 x w2 pooled map there is no window (padding, the odd trailing row /
            // column): the patch elements there are written as zeros
            const bool win = in && (unsigned)Y2 < (unsigned)p.h2 && (unsigned)X2 < (unsigned)p.w2;
            qc[i] = c;
            qv[i] = win ? 8u * (unsigned)(c * hw2 + Y2 * p.w2 + X2) : OOB;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int py = 2 * Y2 + (sl >> 1) - iy0, px = 2 * X2 + (sl & 1) - ix0;
                const bool ip = in && (unsigned)py < (unsigned)PH && (unsigned)px < (unsigned)PW;
                qs[i][sl] = ip ? c * PP + py * PW + px : -1;
                pv[i][sl] = (ip && win)
                                ? 8u * (unsigned)(c * HW + (2 * Y2 + (sl >> 1)) * p.W + 2 * X2 + (sl & 1))
                                : OOB;
            }
        }
    }
    unsigned woff[NWP];
#pragma unroll
    for (int j = 0; j < NWP; ++j) {
        const int f = j * 256 + tid;
        woff[j] = f < WPC ? (unsigned)((f / (BM / 2)) * p.Mpad + m0) * 8u + (unsigned)(f % (BM / 2)) * 16u
                          : OOB;
    }

    // one image per tile: descriptors start at image b of each source
    const double* base1 = UNPOOL ? (MASKIN ? p.x1 : p.pre + (size_t)b * C1 * HW) : p.x1 + (size_t)b * C1 * HW;
    const unsigned char* basem = MASKIN ? p.mask_in + (size_t)b * C1 * hw2 : nullptr;
    const double* base2 = p.C2 > 0 ? p.x2 + (size_t)b * p.C2 * HW : base1;
    const unsigned n1 = (unsigned)(C1 * HW) * 8u, n2 = p.C2 > 0 ? (unsigned)(p.C2 * HW) * 8u : n1;
    const double* baseq = (UNPOOL && !MASKIN) ? p.pooled + (size_t)b * C1 * hw2 : nullptr;
    const double* baseu = UNPOOL ? p.x1 + (size_t)b * C1 * hw2 : nullptr;
    const int nq = C1 * hw2 * 8;
    const i32x4s s_x1 = mk_srsrc(base1, n1), s_x2 = mk_srsrc(base2, n2);
    const i32x4s s_w = mk_srsrc(p.wp, (unsigned)(p.Kpad * p.Mpad) * 8u);
    const unsigned lds_w = __builtin_amdgcn_readfirstlane(lds_addr(&Ws[0][0]) + (unsigned)wave * 1024u);
    const unsigned lds_p = __builtin_amdgcn_readfirstlane(lds_addr(&Ps[0][0]) + (unsigned)wave * 256u);

    double xq[(UNPOOL && !MASKIN) ? NQ : 1], xu[UNPOOL ? NQ : 1], xp[(UNPOOL && !MASKIN) ? NQ : 1][4];
    unsigned xm[MASKIN ? NQ : 1];

#define H64_STAGE_W(KT, BUF)                                                                       \
    {                                                                                              \
        const unsigned sow = __builtin_amdgcn_readfirstlane((unsigned)((KT) * BK * p.Mpad) * 8u);  \
        _Pragma("unroll") for (int j = 0; j < NWP; ++j)                                            \
            if ((j + 1) * 256 <= WPC || tid + 256 * j < WPC)                                       \
                dma16(s_w, lds_w + (unsigned)((BUF) * WSZ * 8 + j * 4096), woff[j], sow);          \
    }
#define H64_LOAD_X(KT, BUF)                                                                        \
    {                                                                                              \
        const int c0 = (KT) * CPT;                                                                 \
        if constexpr (UNPOOL) {                                                                    \
            const int crem = C1 - c0;                                                              \
            static_for<0, NQ>([&](auto I) __attribute__((always_inline)) {                         \
                constexpr int i = decltype(I)::value;                                              \
                const bool cok = qc[i] < crem;                                                     \
                const unsigned vo2 = cok ? qv[i] : OOB;                                            \
                xu[i] = ld64(baseu, nq, vo2, (unsigned)(c0 * hw2) * 8u);                           \
                if constexpr (MASKIN) {                                                            \
                    xm[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(                         \
                        __builtin_amdgcn_make_buffer_rsrc((void*)basem, 0, nq >> 3, RSRC_W3),      \
                        (int)(vo2 == OOB ? OOB : vo2 >> 3), (int)(c0 * hw2), 0);                   \
                } else {                                                                           \
                    xq[i] = ld64(baseq, nq, vo2, (unsigned)(c0 * hw2) * 8u);                       \
                    _Pragma("unroll") for (int sl = 0; sl < 4; ++sl)                               \
                        xp[i][sl] = ld64(base1, (int)n1, cok ? pv[i][sl] : OOB, (unsigned)(c0 * HW) * 8u); \
                }                                                                                  \
            });                                                                                    \
        } else {                                                                                   \
            const bool s1 = c0 < C1;                                                               \
            const int crem = (s1 ? C1 : Ctot) - c0;                                                \
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((s1 ? c0 : c0 - C1) * HW) * 8u); \
            static_for<0, ND>([&](auto I) __attribute__((always_inline)) {                         \
                constexpr int i = decltype(I)::value;                                              \
                const unsigned vo = cl[i] < crem ? voff[i] : OOB;                                  \
                if (s1) dma4(s_x1, lds_p + (unsigned)((BUF) * PSZ * 8 + i * 1024), vo, so);        \
                else dma4(s_x2, lds_p + (unsigned)((BUF) * PSZ * 8 + i * 1024), vo, so);           \
            });                                                                                    \
        }                                                                                          \
    }
#define H64_STORE_X(BUF)                                                                           \
    if constexpr (UNPOOL) {                                                                        \
        static_for<0, NQ>([&](auto I) __attribute__((always_inline)) {                             \
            constexpr int i = decltype(I)::value;                                                  \
            _Pragma("unroll") for (int sl = 0; sl < 4; ++sl)                                       \
                if (qs[i][sl] >= 0) {                                                              \
                    if constexpr (MASKIN) Ps[BUF][qs[i][sl]] = ((xm[i] >> sl) & 1u) ? xu[i] : 0.0; \
                    else Ps[BUF][qs[i][sl]] = (xp[i][sl] == xq[i]) ? xu[i] : 0.0;                  \
                }                                                                                  \
        });                                                                                        \
    }

    f64x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

    // patch offset (doubles) of this lane's k = 4 s + kq for every k-step s of a k-tile
    int koff[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = 4 * s + kq;
        const int c = k / 9, tap = k - 9 * c;
        koff[s] = c * PP + (tap / 3) * PW + tap % 3;
    }
    const int lbase = wave * RW * PW + n;

    const int nkt = p.Kpad / BK;
    H64_STAGE_W(0, 0)
    H64_LOAD_X(0, 0)
    H64_STORE_X(0)
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nkt;
        // k-tile kt has landed (own pieces retired / own LDS writes done, the barrier publishes
        // everyone's) and every wave is done reading the other ring slot, overwritten from here on
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (more) {
            H64_STAGE_W(kt + 1, buf ^ 1)
            H64_LOAD_X(kt + 1, buf ^ 1)
        }
        const double* wsb = &Ws[buf][0];
        const double* psb = &Ps[buf][0];
        double a[2][TM], bq[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = wsb[kq * BM + i * 16 + n];
#pragma unroll
        for (int j = 0; j < TN; ++j) bq[0][j] = psb[lbase + koff[0] + (j >> 1) * PW + (j & 1) * 16];
        static_for<0, NS>([&](auto S) __attribute__((always_inline)) {
            constexpr int st = decltype(S)::value;
            if constexpr (st + 1 < NS) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(st + 1) & 1][i] = wsb[(4 * (st + 1) + kq) * BM + i * 16 + n];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bq[(st + 1) & 1][j] = psb[lbase + koff[st + 1] + (j >> 1) * PW + (j & 1) * 16];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[st & 1][i], bq[st & 1][j], acc[i][j],
                                                                     0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        if (more) H64_STORE_X(buf ^ 1)
    }
#undef H64_STAGE_W
#undef H64_LOAD_X
#undef H64_STORE_X

    // epilogue.  f64 16x16x4 C/D layout: column = lane & 15 (pixel), row = (lane >> 4) + 4 r
    const size_t OPL = (size_t)p.out_H * p.out_W, APL = (size_t)p.AH * p.AW, PPL = (size_t)p.pool_H * p.pool_W;
    // Fused 2x2 max-pool (Pool2DLayer(x, 2), ignore_border): window origin, tile origin and a wave's RW = 2 rows
    // are even, so column tiles j, j + 2 of a lane are the two rows and lane ^ 1 the other column of one window;
    // the pooled value is the max of the four STORED values.  A trailing unpaired row / column has no window.
    const bool pooling = BM == 64 && RW == 2 && p.pool != nullptr;
    // One image's planes through buffer descriptors with 32-bit offsets (host: conv_halo_f64_ok); elements outside
    // the window / past Cout get the out-of-range offset instead of a branch, and the skip-add loads of a column
    // tile are all issued before its stores (a load behind a store through plain pointers cannot move above it).
    typedef int i32x2e __attribute__((ext_vector_type(2)));
    const int OPLi = p.out_H * p.out_W, APLi = p.AH * p.AW;
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.out ? p.out + (size_t)b * p.out_ctot * OPL : nullptr), 0, p.out ? p.out_ctot * OPLi * 8 : 0, RSRC_W3);
    const __amdgpu_buffer_rsrc_t r_add = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.add ? p.add + (size_t)b * p.Cout * APL : nullptr), 0, p.add ? p.Cout * APLi * 8 : 0, RSRC_W3);
    double bv[TM][4];
    unsigned coo[TM][4], coa[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = m0 + i * 16 + kq + 4 * r;
            const bool cok = co < p.Cout;
            bv[i][r] = (p.bias && cok) ? p.bias[co] : 0.0;
            coo[i][r] = cok ? 8u * (unsigned)((p.out_c0 + co) * OPLi) : OOB;
            coa[i][r] = cok ? 8u * (unsigned)(co * APLi) : OOB;
        }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int wy = wy0 + wave * RW + (j >> 1), wx = wx0 + (j & 1) * 16 + n;
        const bool ok = wy < p.OH && wx < p.OW;
        const unsigned o0 = ok ? 8u * (unsigned)((p.out_y0 + wy) * p.out_W + p.out_x0 + wx) : OOB;
        const unsigned a0 = (ok && p.add) ? 8u * (unsigned)((p.ay0 + wy) * p.AW + p.ax0 + wx) : OOB;
        double av[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                av[i][r] = __builtin_bit_cast(double, (i32x2e)__builtin_amdgcn_raw_buffer_load_b64(
                    r_add, (int)((a0 == OOB || coa[i][r] == OOB) ? OOB : a0 + coa[i][r]), 0, 0));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = acc[i][j][r];
                if (ok && coo[i][r] != OOB) {
                    if (p.bias) v += bv[i][r];
                    if (p.add) v += av[i][r];
                    if (p.relu) v = fmax(v, 0.0);
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2e, v), r_out,
                                                      (int)((o0 == OOB || coo[i][r] == OOB) ? OOB : o0 + coo[i][r]), 0, 0);
                acc[i][j][r] = v;
            }
    }
    if (pooling) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {                           // column half of the tile row pair
            const int wy = wy0 + wave * RW, wx = wx0 + jj * 16 + n;
            const int py = (p.oy0 + wy) >> 1, px = (p.ox0 + wx) >> 1;
            const bool ok = !(n & 1) && wy + 1 < p.OH && wx + 1 < p.OW && py < p.pool_H && px < p.pool_W;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = m0 + i * 16 + kq + 4 * r;
                    const double m0_ = fmax(acc[i][jj][r], acc[i][jj + 2][r]);
                    const double m = fmax(m0_, __shfl_xor(m0_, 1, 64));
                    // bit (row & 1) * 2 + (col & 1): pre == pooled -- this lane's column, then lane ^ 1's
                    const unsigned mine = (acc[i][jj][r] == m ? 1u : 0u) | (acc[i][jj + 2][r] == m ? 4u : 0u);
                    const unsigned bits = mine | ((unsigned)__shfl_xor((int)mine, 1, 64) << 1);
                    if (ok && co < p.Cout) {
                        const size_t po = ((size_t)b * p.Cout + co) * PPL + (size_t)py * p.pool_W + px;
                        p.pool[po] = m;
                        if (p.mask_out) p.mask_out[po] = (unsigned char)bits;
                    }
                }
        }
    }
}

template <int BM, int TH>
int launch_h64(hipStream_t s, const ConvParams64& cp, bool unpool) {
    ConvParams64 p = cp;
    const int tiles_y = (p.OH + TH - 1) / TH, tiles_x = (p.OW + 31) / 32;
    p.n_ptiles = p.B * tiles_y * tiles_x;
    p.n_mtiles = BM == 16 ? (p.Cout + 15) / 16 : p.Mpad / 64;
    const dim3 grid(p.n_ptiles * p.n_mtiles), block(256);
    if (unpool && p.mask_in)
        IISEG_LAUNCH((conv_halo_f64_kernel<BM, TH, true, true>), grid, block, 0, s, p, tiles_y, tiles_x);
    else if (unpool)
        IISEG_LAUNCH((conv_halo_f64_kernel<BM, TH, true>), grid, block, 0, s, p, tiles_y, tiles_x);
    else
        IISEG_LAUNCH((conv_halo_f64_kernel<BM, TH, false>), grid, block, 0, s, p, tiles_y, tiles_x);
    return iiseg_check_launch();
}

}  // namespace

bool iiseg::iiseg_conv_halo_f64_ok(const ConvParams64& p, int KH, int KW) {
    static const int on = getenv("IISEG_F64_HALO") ? atoi(getenv("IISEG_F64_HALO")) : 1;
    if (!on || KH != 3 || KW != 3 || p.transposed || p.dil != 1) return false;
    if (p.Kpad % 36 || p.Mpad % 64) return false;
    if (p.C2 > 0 && p.C1 % 4) return false;      // a k-tile (4 channels) must not straddle the sources
    const int64_t cmax = p.C1 > p.C2 ? p.C1 : p.C2;
    if (cmax * p.H * p.W * 8 >= (1ll << 31) - 8) return false;     // per-image 32-bit byte offsets
    if ((int64_t)p.Kpad * p.Mpad * 8 >= (1ll << 31)) return false;
    // the epilogue addresses one image's output / skip-add planes with 32-bit byte offsets
    if (((int64_t)p.out_ctot + 64) * p.out_H * p.out_W * 8 >= (1ll << 31)) return false;
    if (p.add && ((int64_t)p.Cout + 64) * p.AH * p.AW * 8 >= (1ll << 31)) return false;
    if ((int64_t)p.B * ((p.OH + 7) / 8) * ((p.OW + 31) / 32) * (p.Mpad / 16) >= (1ll << 31)) return false;
    return true;
}

int iiseg::iiseg_launch_conv_halo_f64(hipStream_t s, const ConvParams64& p, bool unpool) {
    if (p.pool && p.Cout <= 16) return IISEG_ERR_UNSUPPORTED;
    static const int th16 = getenv("IISEG_F64_HALO16_TH") ? atoi(getenv("IISEG_F64_HALO16_TH")) : 8;
    if (p.Cout <= 16) return th16 == 16 ? launch_h64<16, 16>(s, p, unpool) : launch_h64<16, 8>(s, p, unpool);
    return launch_h64<64, 8>(s, p, unpool);
}
