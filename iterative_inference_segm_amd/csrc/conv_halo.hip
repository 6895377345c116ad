// Halo-tile direct 3x3 convolution for gfx950 (CDNA4), fp32 MFMA: the kernel for the SHALLOW 3x3
// layers of the hot path (few channels, 113^2..422^2 maps), where conv_taps.hip is bound by its
// gather (one buffer_load per (channel, tap, pixel)) and Winograd by its transforms' HBM traffic.
//
// A workgroup owns a TH x 32 pixel tile of one image and BM output channels.  Per k-tile (4 input
// channels = 36 k) it stages the (TH+2) x 34 input PATCH of those channels into LDS once --
// 9x fewer global loads and LDS writes than materialising the im2col tile -- and the MFMA B
// operand is read straight out of the patch: element (k = (c, ky, kx), pixel (y, x)) lives at
// patch[c][y + ky][x + kx], i.e. lane address = lane base + a compile-time constant per k.
// The accumulation order over k is the same (channel-major, tap-minor, sequential 32x32x2 steps)
// as conv_taps.hip / conv_igemm.hip, so results are bit-identical to those kernels.
//
// Fusions: two-source channel concat (h first; needs C1 % 4 == 0), DePool2D equality-mask unpool
// as the patch load (3 loads + compare per patch element instead of per im2col element:
// layers/mylayers.py:88-115), bias / skip-add with crop / ReLU / window / placement epilogue.
// Same Lasagne Conv2DLayer call sites as conv_taps.hip (models/fcn8.py:34-45,
// models/fcn_down.py:102-104, models/fcn_up.py:83-86).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "iiseg.h"
#include "common.h"
#include "conv_common.h"

using namespace iiseg;

namespace {

constexpr int RSRC_W3 = 0x00027000;
constexpr unsigned OOB = 0x80000000u;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const float* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, RSRC_W3);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc_b(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, RSRC_W3);
}
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

__device__ __forceinline__ unsigned buf_ld_u8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b8(r, (int)voff, (int)soff, 0);
}

// MASKIN (with UNPOOL): the DePool2D mask comes as bytes (ConvParams::mask_in) instead of being
// formed from pre == pooled: 2 loads (1 + 4 bytes) per patch element instead of 3 (12 bytes)
template <int BM, int TH, int WM, int WN, bool UNPOOL, bool MASKIN = false>
__global__ __launch_bounds__(256, 2) void conv_halo_f32_kernel(const ConvParams p, const int tiles_y,
                                                               const int tiles_x) {
    constexpr int CPT = 4, BK = 9 * CPT, NCH = BK / 2;
    constexpr int TW = 32, PH = TH + 2, PW = TW + 2, PP = PH * PW;
    constexpr int PE = CPT * PP;              // patch elements per k-tile
    constexpr int NE = (PE + 255) / 256;      // ... per thread
    constexpr int WTM = BM / WM, TM = WTM / 32;
    constexpr int RW = TH / WN, TN = RW;      // output rows per wave = 32-pixel MFMA column tiles
    constexpr int WVEC = BK * BM / 4, WPT = (WVEC + 255) / 256;
    static_assert(WM * WN == 4 && TH % WN == 0 && BM % (WM * 32) == 0 && WPT <= 5, "tile config");
    // MASKIN: a thread stages POOLED positions (`up` value + mask byte, 2 loads) and writes the up
    // to four patch elements of the 2x2 block -- a third of the loads of the per-pixel form
    constexpr int QH = PH / 2 + 1, QW = PW / 2 + 1, QP = QH * QW, QE = CPT * QP;
    constexpr int NQ = (QE + 255) / 256;
    static_assert(NQ <= NE, "staging registers");

    __shared__ __attribute__((aligned(16))) float Ws[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Ps[2][NE * 256];

    int pt, mt;
    tile_of_block(blockIdx.x, gridDim.x, p.n_ptiles, p.n_mtiles, pt, mt);
    const int m0 = mt * BM;
    const int tpi = tiles_y * tiles_x;
    const int b = pt / tpi;
    const int tr = pt - b * tpi;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int wy0 = ty * TH, wx0 = tx * TW;                        // tile origin, window coords
    const int iy0 = p.oy0 + wy0 - p.pad, ix0 = p.ox0 + wx0 - p.pad;  // patch origin, input coords

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int HW = p.H * p.W, hw2 = p.h2 * p.w2;
    const int C1 = p.C1, Ctot = p.C1 + p.C2;

    // ---- patch staging: element e = i*256 + tid  ->  (channel of the k-tile, patch y, patch x) ----
    unsigned voff[NE], voff2[UNPOOL ? NE : 1];
    int cl[NE], bsel[UNPOOL ? NE : 1];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = i * 256 + tid;
        const int c = e / PP, rr = e - c * PP;
        const int py = rr / PW, px = rr - py * PW;
        const int iy = iy0 + py, ix = ix0 + px;
        bool ok = e < PE && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        cl[i] = c;
        voff[i] = ok ? 4u * (unsigned)(c * HW + iy * p.W + ix) : OOB;
        if constexpr (UNPOOL) {
            // DePool2D (layers/mylayers.py:95-114): only the 2h x 2w region has pooling windows
            ok = ok && iy < 2 * p.h2 && ix < 2 * p.w2;
            voff2[i] = ok ? 4u * (unsigned)(c * hw2 + (iy >> 1) * p.w2 + (ix >> 1)) : OOB;
            bsel[i] = ((iy & 1) << 1) | (ix & 1);
        }
    }
    unsigned qv[MASKIN ? NQ : 1];
    int qs[MASKIN ? NQ : 1][4], qc[MASKIN ? NQ : 1];
    if constexpr (MASKIN) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int e = i * 256 + tid;
            const int c = e / QP, r = e - c * QP;
            const int qy = r / QW, qx = r - qy * QW;
            const int Y2 = (iy0 >> 1) + qy, X2 = (ix0 >> 1) + qx;       // (arithmetic shifts: floor)
            const bool in = e < QE;
            qc[i] = c;
            // DePool2D (layers/mylayers.py:95-114): outside the h2 x w2 pooled map there is no
            // window (padding, the odd trailing row / column): the elements written there are zero
            qv[i] = (in && (unsigned)Y2 < (unsigned)p.h2 && (unsigned)X2 < (unsigned)p.w2)
                        ? 4u * (unsigned)(c * hw2 + Y2 * p.w2 + X2) : OOB;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int py = 2 * Y2 + (sl >> 1) - iy0, px = 2 * X2 + (sl & 1) - ix0;
                qs[i][sl] = (in && (unsigned)py < (unsigned)PH && (unsigned)px < (unsigned)PW)
                                ? c * PP + py * PW + px : -1;
            }
        }
    }
    // one image per tile: descriptors start at image b of each source
    const unsigned char* basem = MASKIN ? p.mask_in + (size_t)b * C1 * hw2 : nullptr;
    const float* base1 = (UNPOOL && !MASKIN) ? p.pre + (size_t)b * C1 * HW : p.x1 + (size_t)b * C1 * HW;
    const float* base2 = p.C2 > 0 ? p.x2 + (size_t)b * p.C2 * HW : base1;
    const int n1 = C1 * HW * 4, n2 = p.C2 > 0 ? p.C2 * HW * 4 : n1;
    const float* baseq = UNPOOL ? p.pooled + (size_t)b * C1 * hw2 : nullptr;
    const float* baseu = UNPOOL ? p.x1 + (size_t)b * C1 * hw2 : nullptr;
    const int nq = C1 * hw2 * 4;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float xv[UNPOOL ? NE : 1], xq[UNPOOL ? NE : 1], xu[UNPOOL ? NE : 1];
    const int wrow0 = tid / (BM / 4);
    constexpr int RPJ = 256 / (BM / 4);
    const int wc4 = tid % (BM / 4);
    const __amdgpu_buffer_rsrc_t wrsrc = mk_rsrc(p.wp, p.Kpad * p.Mpad * 4);

    // channels [c0, c0+4) of the logical (concatenated) input; the source is tile-uniform
#define HALO_LOAD_X(KT, BUF)                                                                       \
    {                                                                                              \
        const int c0 = (KT) * CPT;                                                                 \
        if constexpr (MASKIN) {                                                                    \
            const int crem = C1 - c0;                                                              \
            static_for<0, NQ>([&](auto I) __attribute__((always_inline)) {                         \
                constexpr int i = decltype(I)::value;                                              \
                const unsigned vo2 = qc[i] < crem ? qv[i] : OOB;                                   \
                xq[i] = __builtin_bit_cast(float, buf_ld_u8(mk_rsrc_b(basem, nq >> 2),             \
                                                            vo2 == OOB ? OOB : vo2 >> 2,           \
                                                            (unsigned)(c0 * hw2)));                \
                xu[i] = buf_ld(mk_rsrc(baseu, nq), vo2, (unsigned)(c0 * hw2) * 4u);                \
            });                                                                                    \
        } else if constexpr (UNPOOL) {                                                             \
            const int crem = C1 - c0;                                                              \
            static_for<0, NE>([&](auto I) __attribute__((always_inline)) {                         \
                constexpr int i = decltype(I)::value;                                              \
                const bool cok = cl[i] < crem;                                                     \
                const unsigned vo = cok ? voff[i] : OOB, vo2 = cok ? voff2[i] : OOB;               \
                if constexpr (MASKIN) {                                                            \
                    xq[i] = __builtin_bit_cast(float, buf_ld_u8(mk_rsrc_b(basem, nq >> 2),         \
                                                                vo2 == OOB ? OOB : vo2 >> 2,       \
                                                                (unsigned)(c0 * hw2)));            \
                } else {                                                                           \
                    xv[i] = buf_ld(mk_rsrc(base1, n1), vo, (unsigned)(c0 * HW) * 4u);              \
                    xq[i] = buf_ld(mk_rsrc(baseq, nq), vo2, (unsigned)(c0 * hw2) * 4u);            \
                }                                                                                  \
                xu[i] = buf_ld(mk_rsrc(baseu, nq), vo2, (unsigned)(c0 * hw2) * 4u);                \
            });                                                                                    \
        } else {                                                                                   \
            const bool s1 = c0 < C1;                                                               \
            const int crem = (s1 ? C1 : Ctot) - c0;                                                \
            const unsigned so = (unsigned)((s1 ? c0 : c0 - C1) * HW) * 4u;                         \
            static_for<0, NE>([&](auto I) __attribute__((always_inline)) {                         \
                constexpr int i = decltype(I)::value;                                              \
                const unsigned vo = cl[i] < crem ? voff[i] : OOB;                                  \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                          \
                    mk_rsrc(s1 ? base1 : base2, s1 ? n1 : n2),                                     \
                    (__attribute__((address_space(3))) void*)(&Ps[BUF][i * 256 + wave * 64]), 4,   \
                    (int)vo, (int)so, 0, 0);                                                       \
            });                                                                                    \
        }                                                                                          \
    }
#define HALO_STORE_X(BUF)                                                                          \
    if constexpr (MASKIN) {                                                                        \
        static_for<0, NQ>([&](auto I) __attribute__((always_inline)) {                             \
            constexpr int i = decltype(I)::value;                                                  \
            /* bit (row & 1) * 2 + (col & 1) of the window's byte: pre == pooled */                \
            _Pragma("unroll") for (int sl = 0; sl < 4; ++sl)                                       \
                if (qs[i][sl] >= 0)                                                                \
                    Ps[BUF][qs[i][sl]] =                                                           \
                        ((__builtin_bit_cast(unsigned, xq[i]) >> sl) & 1u) ? xu[i] : 0.f;          \
        });                                                                                        \
    } else if constexpr (UNPOOL) {                                                                 \
        static_for<0, NE>([&](auto I) __attribute__((always_inline)) {                             \
            constexpr int i = decltype(I)::value;                                                  \
            /* padding / odd trailing row+col read 0 == 0 -> up, which is also 0 there */          \
            if constexpr (MASKIN)                                                                  \
                Ps[BUF][i * 256 + tid] =                                                           \
                    ((__builtin_bit_cast(unsigned, xq[i]) >> bsel[i]) & 1u) ? xu[i] : 0.f;         \
            else                                                                                   \
            Ps[BUF][i * 256 + tid] = (xv[i] == xq[i]) ? xu[i] : 0.f;                               \
        });                                                                                        \
    }
#define HALO_W_ON(j) ((j) < WPT && (((j) + 1) * 256 <= WVEC || tid + 256 * (j) < WVEC))
#define HALO_W_DMA(KT, BUF, j)                                                                     \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                      \
        wrsrc, (__attribute__((address_space(3))) void*)(&Ws[BUF][0][0] + (wave * 64 + 256 * (j)) * 4), \
        16, (int)(4u * (unsigned)(((KT) * BK + wrow0 + (j) * RPJ) * p.Mpad + m0 + wc4 * 4)), 0, 0, 0)
#define HALO_LOAD_W(KT, BUF)                                                                       \
    {                                                                                              \
        if (HALO_W_ON(0)) HALO_W_DMA(KT, BUF, 0);                                                  \
        if (HALO_W_ON(1)) HALO_W_DMA(KT, BUF, 1);                                                  \
        if (HALO_W_ON(2)) HALO_W_DMA(KT, BUF, 2);                                                  \
        if (HALO_W_ON(3)) HALO_W_DMA(KT, BUF, 3);                                                  \
        if (HALO_W_ON(4)) HALO_W_DMA(KT, BUF, 4);                                                  \
    }

    const int nkt = p.Kpad / BK;
    HALO_LOAD_X(0, 0)
    HALO_LOAD_W(0, 0)
    HALO_STORE_X(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int lbase = wn * RW * PW + l31;   // this lane's pixel inside the patch (tap 0,0)
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nkt;
        float a[2][TM], bq[2][TN];
        // k = 2*ch + lh  ->  patch offset of (channel, ky, kx) = compile-time constant per ch
#define HALO_KOFF(K) (((K) / 9) * PP + (((K) % 9) / 3) * PW + ((K) % 9) % 3)
        {
            const int kb = lh ? HALO_KOFF(1) : HALO_KOFF(0);
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = Ws[buf][lh][wm * WTM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) bq[0][j] = Ps[buf][lbase + kb + j * PW];
        }
        static_for<0, NCH>([&](auto CH) __attribute__((always_inline)) {
            constexpr int ch = decltype(CH)::value;
            if constexpr (ch + 1 < NCH) {
                constexpr int k0 = 2 * (ch + 1);
                const int kb = lh ? HALO_KOFF(k0 + 1) : HALO_KOFF(k0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[(ch + 1) & 1][i] = Ws[buf][k0 + lh][wm * WTM + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) bq[(ch + 1) & 1][j] = Ps[buf][lbase + kb + j * PW];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ch & 1][i], bq[ch & 1][j],
                                                                     acc[i][j], 0, 0, 0);
            if constexpr (ch == 0) {
                if (more) {
                    HALO_LOAD_W(kt + 1, buf ^ 1)
                    HALO_LOAD_X(kt + 1, buf ^ 1)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (more) HALO_STORE_X(buf ^ 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef HALO_KOFF
#undef HALO_LOAD_X
#undef HALO_STORE_X
#undef HALO_LOAD_W
#undef HALO_W_DMA
#undef HALO_W_ON

    // ---- epilogue: bias, skip add (center-cropped), ReLU, NCHW store, fused pool / mask bytes ----
    // C/D layout of the 32x32 MFMA: column = lane & 31 (pixel x), row = (r&3) + 8*(r>>2) + 4*lh.
    // Stored as it stands every store instruction writes 4 bytes per lane into two planes; stamped
    // in-kernel that epilogue was 16-35 % of a workgroup's cycles (64 stores per thread with 64-bit
    // address arithmetic and bounds branches each).  So the tile takes one trip through LDS (the
    // weight ring's bytes, CHR channels at a time): Cs[c][row][x] <- acc (conflict-free: lanes are
    // consecutive x), then every thread moves 16-byte pieces of whole rows through buffer descriptors
    // of image b with 32-bit offsets computed once per thread; out-of-range pieces get the
    // out-of-bounds offset instead of a branch.  Same values in the same order of operations.
    constexpr int WSB = 2 * BK * BM * 4;                          // bytes of the weight ring
    constexpr int CHR = WSB >= 32 * TH * 128 ? 32 : (WSB >= 16 * TH * 128 ? 16 : 8);
    static_assert(CHR * TH * 128 <= WSB && 256 % (TH * 8) == 0, "staging tile must fit the weight ring");
    constexpr int CPP = 256 / (TH * 8);                           // channels per store pass
    constexpr int NSP = CHR / CPP;                                // store passes per round
    constexpr int QPP = 256 / ((TH / 2) * 16);                    // channels per pool pass
    constexpr int NQP = CHR / QPP;
    float* Cs = &Ws[0][0][0];
    const int OPL = p.out_H * p.out_W, APL = p.AH * p.AW, PPL = p.pool_H * p.pool_W;
    const bool pooling = p.pool != nullptr;
    const __amdgpu_buffer_rsrc_t r_bias = mk_rsrc(p.bias, p.bias ? p.Cout * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_out =
        mk_rsrc(p.out ? p.out + (size_t)b * p.out_ctot * OPL : nullptr, p.out ? p.out_ctot * OPL * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_add =
        mk_rsrc(p.add ? p.add + (size_t)b * p.Cout * APL : nullptr, p.add ? p.Cout * APL * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_pool =
        mk_rsrc(pooling ? p.pool + (size_t)b * p.Cout * PPL : nullptr, pooling ? p.Cout * PPL * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_mask = mk_rsrc_b(
        p.mask_out ? p.mask_out + (size_t)b * p.Cout * PPL : nullptr, p.mask_out ? p.Cout * PPL : 0);
    const bool relu1 = p.relu && !p.add, relu2 = p.relu && p.add;   // (with a skip-add the ReLU comes
                                                                    //  after the sum)
    // store pass: thread -> (channel s_c of the pass, row, 4-pixel piece)
    const int s_c = tid / (TH * 8), s_rem = tid % (TH * 8);
    const int s_row = s_rem >> 3, s_x4 = (s_rem & 7) * 4;
    const int s_wy = wy0 + s_row, s_wx = wx0 + s_x4;
    const int s_nv = min(4, p.OW - s_wx);
    const bool s_ok = s_wy < p.OH && s_nv > 0;
    const unsigned s_out0 = 4u * (unsigned)((p.out_c0 + m0 + s_c) * OPL + (p.out_y0 + s_wy) * p.out_W +
                                            p.out_x0 + s_wx);
    const unsigned s_add0 = 4u * (unsigned)((m0 + s_c) * APL + (p.ay0 + s_wy) * p.AW + p.ax0 + s_wx);
    // pool pass: thread -> (channel q_c of the pass, pooled row, pooled column) of the staged tile
    const int q_c = tid / ((TH / 2) * 16), q_rem = tid % ((TH / 2) * 16);
    const int q_row = q_rem >> 4, q_col = q_rem & 15;
    const int q_wy = wy0 + 2 * q_row, q_wx = wx0 + 2 * q_col;
    const int q_py = (p.oy0 + q_wy) >> 1, q_px = (p.ox0 + q_wx) >> 1;
    const bool q_ok = pooling && q_wy + 1 < p.OH && q_wx + 1 < p.OW && q_py < p.pool_H && q_px < p.pool_W;
    const unsigned q_off0 = (unsigned)((m0 + q_c) * PPL + q_py * p.pool_W + q_px);   // elements
    const int lrow = wn * RW;
#pragma unroll
    for (int wmr = 0; wmr < WM; ++wmr)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int h = 0; h < 32 / CHR; ++h) {
                // channels [cb, cb + CHR) of the workgroup's BM: registers r with 8*(r>>2) in the part
                const int cb = wmr * WTM + i * 32 + h * CHR;
                constexpr int RN = CHR / 2;                       // registers per part (CHR=8: 4, 16: 8, 32: 16)
                float bv[RN];
#pragma unroll
                for (int rr = 0; rr < RN; ++rr) {
                    const int r = h * RN + rr;
                    bv[rr] = buf_ld(r_bias, 4u * (unsigned)(m0 + wmr * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh), 0);
                }
                // the skip-add pieces of the round's NSP store passes, in flight while the tile goes through
                // LDS (loaded inside the pass each of them followed the previous pass's store: a load cannot
                // move above an earlier store -- NSP memory round trips in a row)
                f32x4 a4r[NSP];
#pragma unroll
                for (int k = 0; k < NSP; ++k) {
                    const bool ok = s_ok && m0 + cb + CPP * k + s_c < p.Cout;
                    a4r[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        r_add, (int)((p.add && ok && s_nv == 4) ? s_add0 + 4u * (unsigned)((cb + CPP * k) * APL) : OOB), 0, 0));
                }
                __syncthreads();                   // previous users of these LDS bytes are done
                if (wm == wmr) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int rr = 0; rr < RN; ++rr) {
                            const int r = h * RN + rr;
                            const int c = (r & 3) + 8 * (r >> 2) + 4 * lh - h * CHR;   // inside the part
                            float v = acc[i][j][r] + bv[rr];
                            if (relu1) v = fmaxf(v, 0.f);
                            Cs[(c * TH + lrow + j) * 32 + l31] = v;
                        }
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < NSP; ++k) {
                    const int c = CPP * k + s_c;
                    const bool ok = s_ok && m0 + cb + c < p.Cout;
                    f32x4 v = *reinterpret_cast<const f32x4*>(Cs + (c * TH + s_row) * 32 + s_x4);
                    const unsigned oo = s_out0 + 4u * (unsigned)((cb + CPP * k) * OPL);
                    const unsigned ao = s_add0 + 4u * (unsigned)((cb + CPP * k) * APL);
                    if (p.add) v += a4r[k];
                    if (ok && s_nv < 4) {          // ragged right edge of the window: element by element
                        for (int e = 0; e < s_nv; ++e) {
                            float t = v[e];
                            if (p.add) t += buf_ld(r_add, ao + 4u * e, 0);
                            if (relu2) t = fmaxf(t, 0.f);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, t), r_out,
                                                                  (int)(oo + 4u * e), 0, 0);
                        }
                    }
                    if (relu2) {
                        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                        v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out,
                                                           (int)((ok && s_nv == 4) ? oo : OOB), 0, 0);
                }
                if (pooling) {
                    // fused 2x2 max-pool of these channels from the staged tile: window origin and TH
                    // are even, so the pairs (2m, 2m+1) of rows / columns are whole inside the tile; a
                    // trailing unpaired row / column of the map has no pooling window (ignore_border)
#pragma unroll
                    for (int k = 0; k < NQP; ++k) {
                        const int c = QPP * k + q_c;
                        const bool ok = q_ok && m0 + cb + c < p.Cout;
                        const float* c0 = Cs + (c * TH + 2 * q_row) * 32 + 2 * q_col;
                        const float m = fmaxf(fmaxf(c0[0], c0[1]), fmaxf(c0[32], c0[33]));
                        const unsigned po = q_off0 + (unsigned)((cb + QPP * k) * PPL);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, m), r_pool,
                                                              (int)(ok ? 4u * po : OOB), 0, 0);
                        // bit (row & 1) * 2 + (col & 1): pre == pooled  (no records without mask_out)
                        const unsigned bits = (c0[0] == m ? 1u : 0u) | (c0[1] == m ? 2u : 0u) |
                                              (c0[32] == m ? 4u : 0u) | (c0[33] == m ? 8u : 0u);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bits, r_mask,
                                                             (int)(ok ? po : OOB), 0, 0);
                    }
                }
            }
}

template <int BM, int TH, int WM, int WN>
int launch_halo(hipStream_t s, const ConvParams& cp, bool unpool) {
    ConvParams p = cp;
    const int tiles_y = (p.OH + TH - 1) / TH, tiles_x = (p.OW + 31) / 32;
    p.n_ptiles = p.B * tiles_y * tiles_x;
    p.n_mtiles = p.Mpad / BM;
    const int grid = p.n_ptiles * p.n_mtiles;
    if (unpool && p.mask_in)
        IISEG_LAUNCH((conv_halo_f32_kernel<BM, TH, WM, WN, true, true>), dim3(grid), dim3(256),
                           0, s, p, tiles_y, tiles_x);
    else if (unpool)
        IISEG_LAUNCH((conv_halo_f32_kernel<BM, TH, WM, WN, true>), dim3(grid), dim3(256), 0, s,
                           p, tiles_y, tiles_x);
    else
        IISEG_LAUNCH((conv_halo_f32_kernel<BM, TH, WM, WN, false>), dim3(grid), dim3(256), 0, s,
                           p, tiles_y, tiles_x);
    return iiseg_check_launch();
}

// ---- 16-channel variant (Cout <= 16: the DAE's last layer, FC-DenseNet's growth-rate-16 convs) ----
// Same patch staging; the matrix step is v_mfma_f32_16x16x4_f32 (16 output channels x 16 pixels
// x 4 k), so a layer with 11 or 16 output channels fills 11/16 or 16/16 of the MFMA rows instead of
// 11/32 or 16/32.  Lane = (pixel n = lane & 15, k sub-step kq = lane >> 4); the accumulation over k
// is still sequential in k (4 at a time), results agree with the 32-row kernels to fp32 rounding.

template <int TH, bool UNPOOL, int DIL, bool BNRELU = false, bool MASKIN = false>
__global__ __launch_bounds__(256, 2) void conv_halo16_f32_kernel(const ConvParams p, const int tiles_y,
                                                                 const int tiles_x) {
    constexpr int BM = 16, CPT = 4, BK = 9 * CPT, NS = BK / 4;
    // dilation DIL (DilatedConv2DLayer of the context module, models/contextmod_dae.py:78-101): the
    // taps sit DIL apart, so the patch has a DIL-wide halo; when DIL > TH the three row bands a
    // tile reads (one per ky) do not overlap and only those 3*TH rows are staged
    constexpr bool COMPACT = DIL > TH;
    constexpr int TW = 32, PH = COMPACT ? 3 * TH : TH + 2 * DIL, PW = TW + 2 * DIL, PP = PH * PW;
    constexpr int PE = CPT * PP, NE = (PE + 255) / 256;
    static_assert(!(UNPOOL && DIL != 1), "DePool2D input only for undilated layers");
    static_assert(!(UNPOOL && BNRELU), "one input fusion at a time");
    constexpr int RW = TH / 4, TN = 2 * RW;     // 16-pixel column tiles per wave
    constexpr int WVEC = BK * BM / 4;           // 144 float4 per weight tile
    static_assert(TH % 4 == 0 && WVEC <= 256, "tile config");
    // MASKIN: pooled-position staging as in conv_halo_f32_kernel (2 loads per 2x2 block, not 8)
    constexpr int QH = PH / 2 + 1, QW = PW / 2 + 1, QP = QH * QW, QE = CPT * QP;
    constexpr int NQ = (QE + 255) / 256;
    static_assert(!MASKIN || (NQ <= NE && DIL == 1), "staging registers");

    __shared__ __attribute__((aligned(16))) float Ws[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Ps[2][NE * 256];

    int pt, mt;
    tile_of_block(blockIdx.x, gridDim.x, p.n_ptiles, p.n_mtiles, pt, mt);
    const int m0 = mt * BM;
    const int tpi = tiles_y * tiles_x;
    const int b = pt / tpi;
    const int tr = pt - b * tpi;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int wy0 = ty * TH, wx0 = tx * TW;
    const int iy0 = p.oy0 + wy0 - p.pad, ix0 = p.ox0 + wx0 - p.pad;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int HW = p.H * p.W, hw2 = p.h2 * p.w2;
    const int C1 = p.C1, Ctot = p.C1 + p.C2;

    unsigned voff[NE], voff2[UNPOOL ? NE : 1];
    int cl[NE], bsel[UNPOOL ? NE : 1];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = i * 256 + tid;
        const int c = e / PP, rr = e - c * PP;
        const int py = rr / PW, px = rr - py * PW;
        const int iy = iy0 + (COMPACT ? (py / TH) * DIL + py % TH : py), ix = ix0 + px;
        bool ok = e < PE && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        cl[i] = c;
        voff[i] = ok ? 4u * (unsigned)(c * HW + iy * p.W + ix) : OOB;
        if constexpr (UNPOOL) {
            ok = ok && iy < 2 * p.h2 && ix < 2 * p.w2;
            voff2[i] = ok ? 4u * (unsigned)(c * hw2 + (iy >> 1) * p.w2 + (ix >> 1)) : OOB;
            bsel[i] = ((iy & 1) << 1) | (ix & 1);
        }
    }
    unsigned qv[MASKIN ? NQ : 1];
    int qs[MASKIN ? NQ : 1][4], qc[MASKIN ? NQ : 1];
    if constexpr (MASKIN) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int e = i * 256 + tid;
            const int c = e / QP, r = e - c * QP;
            const int qy = r / QW, qx = r - qy * QW;
            const int Y2 = (iy0 >> 1) + qy, X2 = (ix0 >> 1) + qx;       // (arithmetic shifts: floor)
            const bool in = e < QE;
            qc[i] = c;
            qv[i] = (in && (unsigned)Y2 < (unsigned)p.h2 && (unsigned)X2 < (unsigned)p.w2)
                        ? 4u * (unsigned)(c * hw2 + Y2 * p.w2 + X2) : OOB;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int py = 2 * Y2 + (sl >> 1) - iy0, px = 2 * X2 + (sl & 1) - ix0;
                qs[i][sl] = (in && (unsigned)py < (unsigned)PH && (unsigned)px < (unsigned)PW)
                                ? c * PP + py * PW + px : -1;
            }
        }
    }
    const unsigned char* basem = MASKIN ? p.mask_in + (size_t)b * C1 * hw2 : nullptr;
    const float* base1 = (UNPOOL && !MASKIN) ? p.pre + (size_t)b * C1 * HW
                                : p.x1 + (size_t)b * (p.in_bstride ? (size_t)p.in_bstride : (size_t)C1 * HW);
    const float* base2 = p.C2 > 0 ? p.x2 + (size_t)b * p.C2 * HW : base1;
    const int n1 = C1 * HW * 4, n2 = p.C2 > 0 ? p.C2 * HW * 4 : n1;
    const float* baseq = UNPOOL ? p.pooled + (size_t)b * C1 * hw2 : nullptr;
    const float* baseu = UNPOOL ? p.x1 + (size_t)b * C1 * hw2 : nullptr;
    const int nq = C1 * hw2 * 4;

    f32x4 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;

    float xv[(UNPOOL || BNRELU) ? NE : 1], xq[UNPOOL ? NE : 1], xu[UNPOOL ? NE : 1];
    int cbn[BNRELU ? NE : 1];   // BNRELU: channel of the staged element, -1 where it is padding
    const __amdgpu_buffer_rsrc_t wrsrc = mk_rsrc(p.wp, p.Kpad * p.Mpad * 4);
    const int wrow = tid / 4, wc4 = tid % 4;

#define H16_LOAD_X(KT, BUF)                                                                        \
    {                                                                                              \
        const int c0 = (KT) * CPT;                                                                 \
        if constexpr (MASKIN) {                                                                    \
            const int crem = C1 - c0;                                                              \
            static_for<0, NQ>([&](auto I) __attribute__((always_inline)) {                         \
                constexpr int i = decltype(I)::value;                                              \
                const unsigned vo2 = qc[i] < crem ? qv[i] : OOB;                                   \
                xq[i] = __builtin_bit_cast(float, buf_ld_u8(mk_rsrc_b(basem, nq >> 2),             \
                                                            vo2 == OOB ? OOB : vo2 >> 2,           \
                                                            (unsigned)(c0 * hw2)));                \
                xu[i] = buf_ld(mk_rsrc(baseu, nq), vo2, (unsigned)(c0 * hw2) * 4u);                \
            });                                                                                    \
        } else if constexpr (UNPOOL) {                                                             \
            const int crem = C1 - c0;                                                              \
            static_for<0, NE>([&](auto I) __attribute__((always_inline)) {                         \
                constexpr int i = decltype(I)::value;                                              \
                const bool cok = cl[i] < crem;                                                     \
                const unsigned vo = cok ? voff[i] : OOB, vo2 = cok ? voff2[i] : OOB;               \
                if constexpr (MASKIN) {                                                            \
                    xq[i] = __builtin_bit_cast(float, buf_ld_u8(mk_rsrc_b(basem, nq >> 2),         \
                                                                vo2 == OOB ? OOB : vo2 >> 2,       \
                                                                (unsigned)(c0 * hw2)));            \
                } else {                                                                           \
                    xv[i] = buf_ld(mk_rsrc(base1, n1), vo, (unsigned)(c0 * HW) * 4u);              \
                    xq[i] = buf_ld(mk_rsrc(baseq, nq), vo2, (unsigned)(c0 * hw2) * 4u);            \
                }                                                                                  \
                xu[i] = buf_ld(mk_rsrc(baseu, nq), vo2, (unsigned)(c0 * hw2) * 4u);                \
            });                                                                                    \
        } else if constexpr (BNRELU) {                                                             \
            const int crem = C1 - c0;                                                              \
            static_for<0, NE>([&](auto I) __attribute__((always_inline)) {                         \
                constexpr int i = decltype(I)::value;                                              \
                const bool ok = cl[i] < crem && voff[i] != OOB;                                    \
                xv[i] = buf_ld(mk_rsrc(base1, n1), ok ? voff[i] : OOB, (unsigned)(c0 * HW) * 4u);  \
                cbn[i] = ok ? c0 + cl[i] : -1;                                                     \
            });                                                                                    \
        } else {                                                                                   \
            const bool s1 = c0 < C1;                                                               \
            const int crem = (s1 ? C1 : Ctot) - c0;                                                \
            const unsigned so = (unsigned)((s1 ? c0 : c0 - C1) * HW) * 4u;                         \
            static_for<0, NE>([&](auto I) __attribute__((always_inline)) {                         \
                constexpr int i = decltype(I)::value;                                              \
                const unsigned vo = cl[i] < crem ? voff[i] : OOB;                                  \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                          \
                    mk_rsrc(s1 ? base1 : base2, s1 ? n1 : n2),                                     \
                    (__attribute__((address_space(3))) void*)(&Ps[BUF][i * 256 + wave * 64]), 4,   \
                    (int)vo, (int)so, 0, 0);                                                       \
            });                                                                                    \
        }                                                                                          \
    }
#define H16_STORE_X(BUF)                                                                           \
    if constexpr (BNRELU) {                                                                        \
        /* same arithmetic as bn_relu_kernel: (x - mean) * (gamma * inv_std) + beta, rectified;   */ \
        /* zero padding stays zero (it pads the NORMALISED map)                                    */ \
        static_for<0, NE>([&](auto I) __attribute__((always_inline)) {                             \
            constexpr int i = decltype(I)::value;                                                  \
            const int c = cbn[i] < 0 ? 0 : cbn[i];                                                 \
            const float g = p.bn_gamma[c] * p.bn_inv_std[c];                                       \
            const float v = (xv[i] - p.bn_mean[c]) * g + p.bn_beta[c];                             \
            Ps[BUF][i * 256 + tid] = (cbn[i] >= 0 && v > 0.f) ? v : 0.f;                           \
        });                                                                                        \
    }                                                                                              \
    if constexpr (MASKIN) {                                                                        \
        static_for<0, NQ>([&](auto I) __attribute__((always_inline)) {                             \
            constexpr int i = decltype(I)::value;                                                  \
            _Pragma("unroll") for (int sl = 0; sl < 4; ++sl)                                       \
                if (qs[i][sl] >= 0)                                                                \
                    Ps[BUF][qs[i][sl]] =                                                           \
                        ((__builtin_bit_cast(unsigned, xq[i]) >> sl) & 1u) ? xu[i] : 0.f;          \
        });                                                                                        \
    } else if constexpr (UNPOOL) {                                                                 \
        static_for<0, NE>([&](auto I) __attribute__((always_inline)) {                             \
            constexpr int i = decltype(I)::value;                                                  \
            if constexpr (MASKIN)                                                                  \
                Ps[BUF][i * 256 + tid] =                                                           \
                    ((__builtin_bit_cast(unsigned, xq[i]) >> bsel[i]) & 1u) ? xu[i] : 0.f;         \
            else                                                                                   \
            Ps[BUF][i * 256 + tid] = (xv[i] == xq[i]) ? xu[i] : 0.f;                               \
        });                                                                                        \
    }
#define H16_LOAD_W(KT, BUF)                                                                        \
    if (tid < WVEC)                                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                  \
            wrsrc, (__attribute__((address_space(3))) void*)(&Ws[BUF][0][0] + wave * 64 * 4), 16,   \
            (int)(4u * (unsigned)(((KT) * BK + wrow) * p.Mpad + m0 + wc4 * 4)), 0, 0, 0);

    const int nkt = p.Kpad / BK;
    H16_LOAD_X(0, 0)
    H16_LOAD_W(0, 0)
    H16_STORE_X(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // patch offset of this lane's k = 4*s + kq for every k-step s of a k-tile (loop-invariant)
    int koff[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = 4 * s + kq;
        const int c = k / 9, tap = k - 9 * c;
        koff[s] = c * PP + (tap / 3) * (COMPACT ? TH : DIL) * PW + (tap % 3) * DIL;
    }
    const int lbase = wave * RW * PW + n;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nkt;
        float a[2], bq[2][TN];
        a[0] = Ws[buf][kq][n];
#pragma unroll
        for (int j = 0; j < TN; ++j) bq[0][j] = Ps[buf][lbase + koff[0] + (j >> 1) * PW + (j & 1) * 16];
        static_for<0, NS>([&](auto S) __attribute__((always_inline)) {
            constexpr int st = decltype(S)::value;
            if constexpr (st + 1 < NS) {
                a[(st + 1) & 1] = Ws[buf][4 * (st + 1) + kq][n];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bq[(st + 1) & 1][j] = Ps[buf][lbase + koff[st + 1] + (j >> 1) * PW + (j & 1) * 16];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st & 1], bq[st & 1][j], acc[j], 0, 0, 0);
            if constexpr (st == 0) {
                if (more) {
                    H16_LOAD_W(kt + 1, buf ^ 1)
                    H16_LOAD_X(kt + 1, buf ^ 1)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (more) H16_STORE_X(buf ^ 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef H16_LOAD_X
#undef H16_STORE_X
#undef H16_LOAD_W

    // C/D layout of the 16x16 MFMA: column = lane & 15 (pixel), row = 4 * (lane >> 4) + r (channel)
    const size_t OPL = (size_t)p.out_H * p.out_W, APL = (size_t)p.AH * p.AW;
    // (the lane's four bias values before the first store, index clamped instead of branched: loaded inside
    // the loop each of them sat between two stores behind a full wait -- 4 TN memory round trips in a row)
    float bias_r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = m0 + 4 * kq + r;
        bias_r[r] = p.bias ? p.bias[co < p.Cout ? co : p.Cout - 1] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int wy = wy0 + wave * RW + (j >> 1), wx = wx0 + (j & 1) * 16 + n;
        if (wy >= p.OH || wx >= p.OW) continue;
        float* outp = p.out + ((size_t)b * p.out_ctot + p.out_c0) * OPL +
                      (size_t)(p.out_y0 + wy) * p.out_W + p.out_x0 + wx;
        const float* addp = p.add ? p.add + (size_t)b * p.Cout * APL +
                                        (size_t)(p.ay0 + wy) * p.AW + p.ax0 + wx
                                  : nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = m0 + 4 * kq + r;
            if (co < p.Cout) {
                float v = acc[j][r] + bias_r[r];
                if (addp) v += addp[(size_t)co * APL];
                if (p.relu) v = fmaxf(v, 0.f);
                outp[(size_t)co * OPL] = v;
            }
        }
    }
}

int launch_halo16(hipStream_t s, const ConvParams& cp, bool unpool) {
    ConvParams p = cp;
    constexpr int TH = 8;
    const int tiles_y = (p.OH + TH - 1) / TH, tiles_x = (p.OW + 31) / 32;
    p.n_ptiles = p.B * tiles_y * tiles_x;
    p.n_mtiles = (p.Cout + 15) / 16;
    const dim3 grid(p.n_ptiles * p.n_mtiles), block(256);
#define H16(U, D) IISEG_LAUNCH((conv_halo16_f32_kernel<TH, U, D>), grid, block, 0, s, p, tiles_y, tiles_x)
    if (p.bn_mean) {
        if (unpool || p.dil != 1 || p.C2 != 0) return IISEG_ERR_UNSUPPORTED;
        IISEG_LAUNCH((conv_halo16_f32_kernel<TH, false, 1, true>), grid, block, 0, s, p, tiles_y,
                           tiles_x);
    } else if (unpool && p.mask_in) {
        IISEG_LAUNCH((conv_halo16_f32_kernel<TH, true, 1, false, true>), grid, block, 0, s, p,
                           tiles_y, tiles_x);
    } else if (unpool) H16(true, 1);
    else switch (p.dil) {
        case 1: H16(false, 1); break;
        case 2: H16(false, 2); break;
        case 4: H16(false, 4); break;
        case 8: H16(false, 8); break;
        case 16: H16(false, 16); break;
        default: return IISEG_ERR_UNSUPPORTED;
    }
#undef H16
    return iiseg_check_launch();
}

}  // namespace

// 1 if the halo kernel can run this (already validated) 3x3 request
bool iiseg_conv_halo_ok(const ConvParams& p, int KH, int KW) {
    if (KH != 3 || KW != 3 || p.transposed) return false;
    // dilated layers only on the 16-channel variant (context module: 11 -> 11 channels)
    if (p.dil != 1 && !(p.Cout <= 16 && (p.dil == 2 || p.dil == 4 || p.dil == 8 || p.dil == 16)))
        return false;
    if (p.Kpad % 36) return false;
    if (p.C2 > 0 && p.C1 % 4) return false;  // a k-tile (4 channels) must not straddle the sources
    const int64_t cmax = p.C1 > p.C2 ? p.C1 : p.C2;
    if (cmax * p.H * p.W * 4 >= (1ll << 31)) return false;  // per-image 32-bit byte offsets
    if ((int64_t)p.B * ((p.OH + 3) / 4) * ((p.OW + 31) / 32) * (p.Mpad / 32) >= (1ll << 31)) return false;
    // the epilogue addresses one image's output / skip-add planes with 32-bit byte offsets
    if (((int64_t)p.out_ctot + 128) * p.out_H * p.out_W * 4 >= (1ll << 31)) return false;
    if (p.add && ((int64_t)p.Cout + 128) * p.AH * p.AW * 4 >= (1ll << 31)) return false;
    return true;
}

int iiseg_launch_conv_halo(hipStream_t s, const ConvParams& p, int bm, bool unpool) {
    static const int h16 = getenv("IISEG_CONV_HALO16") ? atoi(getenv("IISEG_CONV_HALO16")) : 1;
    if (h16 && p.Cout <= 16) return launch_halo16(s, p, unpool);
    switch (bm) {
        case 128: return launch_halo<128, 4, 2, 2>(s, p, unpool);
        case 64: return launch_halo<64, 8, 1, 4>(s, p, unpool);
        default: return launch_halo<32, 8, 1, 4>(s, p, unpool);
    }
}
