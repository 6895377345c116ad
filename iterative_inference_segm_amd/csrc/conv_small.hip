// 1x1 / (dilated) 3x3 convolution between at most 16 and at most 16 channels in fp32 on the VECTOR ALU: the
// layers of the context-module DAE (models/contextmod_dae.py:74-105: conv3x3 on [image, y] -> six
// DilatedConv2DLayer 11 -> 11, dilation 1, 2, 4, 8, 16, 1 -> 1x1; 50 refinement steps in BASELINE configs[4]).
//
// Why not the matrix pipe: with 11 channels on either side a 16x16x4 fp32 MFMA tile multiplies 16 rows for 11
// and 12 k for 11 -- half of the products are padding, the 16-row halo kernel ran these layers at 24-45 TFLOP/s
// nominal, 2.2-2.6 x their HBM time -- and a dilated layer's patch is 2.5-6 x its tile (three row bands 2 d apart,
// staged dword by dword).  The layers are bound by HBM traffic (read 11 planes, write 11 planes), not by
// arithmetic: here a thread owns four consecutive pixels and all output channels in registers, operands come
// straight from L1 / L2 in coalesced row segments (no LDS patch, no halo amplification whatever the dilation),
// the weights (Wp[k][Mpad] of iiseg_conv_pack_f32) are scalar loads -- SGPR operands of the FMAs -- and the
// products are v_pk_fma_f32 pairs of output channels.  Sum order: bias, then channel-major / tap-minor
// sequential fp32 FMAs.  (Measured on the way: one pixel per thread with the weights broadcast from LDS or as
// scalars runs at the speed of the halo kernel -- 99 dword loads per pixel keep the texture path busier than the
// FMAs keep the ALU.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "iiseg.h"
#include "common.h"
#include "conv_common.h"
#include "tail_math.h"

using namespace iiseg;

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
constexpr int RSRC_W3 = 0x00027000;
constexpr unsigned OOB = 0x80000000u;

// T taps (1 or 9); COP: output channels padded to a multiple of 4 (<= 16).  Tile = 16 rows x 64 columns: thread
// (row tid / 16, column group tid % 16) owns FOUR consecutive pixels of a row, so a tap of a channel is one
// 16-byte load for four pixels (a wave: four 256-byte row segments) and every weight pair multiplies four pixels
// -- the vector memory path is then behind the FMAs, not in front of them.  'valid' layers only (pad 0: every
// tap of a valid pixel lies inside its row): with zero padding the border groups would have to load element by
// element, and that form measured slower than the halo kernel (0.26 / 0.30 against 0.20 / 0.26 ms on the two
// padded layers of the module), which keeps them.
template <int T, int COP>
__global__ __launch_bounds__(256) void conv_small_f32_kernel(const ConvParams p, const int tiles_y, const int tiles_x) {
    constexpr int KW = T == 9 ? 3 : 1;
    const int tid = threadIdx.x, tx = tid & 15, tyy = tid >> 4;
    const int tpi = tiles_y * tiles_x;
    const int b = blockIdx.x / tpi;
    const int tr = blockIdx.x - b * tpi;
    const int ty = tr / tiles_x, txx = tr - ty * tiles_x;
    const int wy = ty * 16 + tyy, wx = txx * 64 + tx * 4;           // window coordinates of the first pixel
    const int Cin = p.C1, HW = p.H * p.W;
    const int nv = min(4, p.OW - wx);                               // valid pixels of the group (<= 0: none)
    const bool any_ok = wy < p.OH && nv > 0;
    const int ix0 = p.ox0 + wx - p.pad;
    // per tap row: byte offset of (iy, ix0) inside a channel plane, or OOB when the row is padding / unused
    unsigned roff[KW];
#pragma unroll
    for (int ky = 0; ky < KW; ++ky) {
        const int iy = p.oy0 + wy - p.pad + ky * p.dil;
        roff[ky] = (any_ok && (unsigned)iy < (unsigned)p.H) ? 4u * (unsigned)(iy * p.W + ix0) : OOB;
    }
    // (a ragged last group reads up to 12 bytes past its pixels -- the next row, or, at the very end of the image,
    // out of the descriptor's range: raw buffer loads are range-checked dword by dword)
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x1 + (size_t)b * Cin * HW), 0, Cin * HW * 4, RSRC_W3);
    f32x2 acc[4][COP / 2];
#pragma unroll
    for (int j = 0; j < COP / 2; ++j) {
        const f32x2 bv = {(p.bias && 2 * j < p.Cout) ? p.bias[2 * j] : 0.f,
                          (p.bias && 2 * j + 1 < p.Cout) ? p.bias[2 * j + 1] : 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e][j] = bv;
    }
    if (p.add) {
        // (no bias then: the chain CONTINUES from the addend -- four pixels of a channel are one 16-byte load)
        const int APL = p.AH * p.AW;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.add + (size_t)b * p.Cout * APL), 0, p.Cout * APL * 4, RSRC_W3);
        const unsigned a0 = any_ok ? 4u * (unsigned)((p.ay0 + wy) * p.AW + p.ax0 + wx) : OOB;
#pragma unroll
        for (int c = 0; c < COP; ++c) {
            const f32x4 av = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                ra, (int)((c < p.Cout && a0 != OOB) ? a0 + 4u * (unsigned)(c * APL) : OOB), 0, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e][c / 2][c & 1] = av[e];
        }
    }
    constexpr int CT = T == 9 ? 2 : 3;     // channels per trip: CT T weight rows = a multiple of the 3-row ring
    f32x4 xv[CT][T];
    auto load = [&](int c, int s) __attribute__((always_inline)) {
        const int so = c * HW * 4;
#pragma unroll
        for (int t = 0; t < T; ++t)
            xv[s][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rx, (int)(roff[t / KW] == OOB ? OOB : roff[t / KW] + 4u * (unsigned)((t % KW) * p.dil)), so, 0));
    };
    // weight row k = c T + tap of Wp[Kpad][Mpad]: wave-uniform address -> scalar loads, the weights are SGPR
    // operands of the FMAs; row k + 2 is fetched while row k multiplies (a scalar load issued right in front of
    // its use parks the wave for the scalar cache's latency, 99 times per pixel group; left to itself hipcc either
    // does that or hoists a dozen rows and spills scalar registers: the order is pinned row by row)
    f32x4 wq[3][COP / 4];
    auto wload = [&](int k, auto S) __attribute__((always_inline)) {
        constexpr int sw = decltype(S)::value;
        const float* wr = p.wp + (size_t)k * p.Mpad;
#pragma unroll
        for (int q = 0; q < COP / 4; ++q) wq[sw][q] = *reinterpret_cast<const f32x4*>(wr + q * 4);
    };
    load(0, 0);
    const int nk = Cin * T;
    wload(0, ic<0>{});
    wload(min(1, nk - 1), ic<1>{});
    for (int c = 0; c < Cin; c += CT) {
        // CT channels per trip: the loads of the next channel are in flight under the FMAs of this one
        static_for<0, CT>([&](auto Hh) __attribute__((always_inline)) {
            constexpr int h = decltype(Hh)::value;
            const int cc = c + h;
            if (cc < Cin) {
                if (cc + 1 < Cin) load(cc + 1, (h + 1) % CT);
                static_for<0, T>([&](auto Tt) __attribute__((always_inline)) {
                    constexpr int t = decltype(Tt)::value;
                    constexpr int ws = (h * T + t) % 3;           // (two channels = 2 T rows; T = 1 or 9: see the trip count)
                    const int kn = min(cc * T + t + 2, nk - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    wload(kn, ic<(ws + 2) % 3>{});
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < COP / 4; ++q) {
                        const f32x2 wa = {wq[ws][q][0], wq[ws][q][1]}, wb = {wq[ws][q][2], wq[ws][q][3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f32x2 xx = {xv[h][t][e], xv[h][t][e]};
                            acc[e][2 * q] = __builtin_elementwise_fma(xx, wa, acc[e][2 * q]);
                            acc[e][2 * q + 1] = __builtin_elementwise_fma(xx, wb, acc[e][2 * q + 1]);
                        }
                    }
                });
            }
        });
    }
    if (!any_ok) return;
    const int OPL = p.out_H * p.out_W;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.out + (size_t)b * p.out_ctot * OPL), 0, p.out_ctot * OPL * 4, RSRC_W3);
    const unsigned o0 = 4u * (unsigned)(p.out_c0 * OPL + (p.out_y0 + wy) * p.out_W + p.out_x0 + wx);
#pragma unroll
    for (int j = 0; j < COP / 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int co = 2 * j + k;
            if (co >= p.Cout) continue;
            float v0 = acc[0][j][k], v1 = acc[1][j][k], v2 = acc[2][j][k], v3 = acc[3][j][k];
            if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            const unsigned oo = o0 + 4u * (unsigned)(co * OPL);
            if (nv == 4) {
                const f32x4 v = {v0, v1, v2, v3};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), ro, (int)oo, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v0), ro, (int)oo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v1), ro, (int)(nv > 1 ? oo + 4u : OOB), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v2), ro, (int)(nv > 2 ? oo + 8u : OOB), 0, 0);
            }
        }
}

template <int T>
int launch_small(hipStream_t s, const ConvParams& p) {
    const int tiles_y = (p.OH + 15) / 16, tiles_x = (p.OW + 63) / 64;
    const dim3 grid(p.B * tiles_y * tiles_x), block(256);
    const int cop = (p.Cout + 3) / 4 * 4;
    switch (cop) {
        case 4: IISEG_LAUNCH((conv_small_f32_kernel<T, 4>), grid, block, 0, s, p, tiles_y, tiles_x); break;
        case 8: IISEG_LAUNCH((conv_small_f32_kernel<T, 8>), grid, block, 0, s, p, tiles_y, tiles_x); break;
        case 12: IISEG_LAUNCH((conv_small_f32_kernel<T, 12>), grid, block, 0, s, p, tiles_y, tiles_x); break;
        default: IISEG_LAUNCH((conv_small_f32_kernel<T, 16>), grid, block, 0, s, p, tiles_y, tiles_x); break;
    }
    return iiseg_check_launch();
}

// ---- the context module's last layers and the refinement update as ONE launch -------------------------------
// models/contextmod_dae.py:98-105 (dilconv6: 3x3 'valid', ReLU; dilconv7: 1x1, linear; softmax) followed by
// iterative_inference.py:203-204, 270-277 (de = y - r, y <- clip(y - step de, 0, 1), ||de||_2 partials).  A
// thread of conv_small_f32_kernel already owns ALL output channels of its four pixels, so the 1x1 layer is C x C
// more FMAs on registers and the softmax / update needs no exchange: two launches, one write + one read of the
// (B, C, H, W) map for each of them and the per-step copy of y into the concat buffer disappear (the updated y
// is stored twice: into y and into its channels of the [image, y] buffer the next step's first layer reads).
// Every value goes through the same FMA chains (3x3: bias, channel-major / tap-minor; 1x1: bias, channel order)
// and the same tail arithmetic (tail_math.h) as in the separate kernels: y comes out bit-identical; the norm
// partials are summed per 16 x 64 tile instead of per 256 pixels (last_norm may differ in its last bits).
struct CtxTailParams {
    const float* x;          // (B, C, H + 2, W + 2): the ReLU output of the layer before
    const float* wp6;        // Wp[9 C][Mpad6] of iiseg_conv_pack_f32 (3x3)
    const float* b6;
    const float* wp7;        // Wp[C][Mpad7] (1x1)
    const float* b7;
    float* y;                // (B, C, H, W) in / out
    const int* active;
    double* partial;         // (B, tiles)
    float* ycat;             // mirror of y: channels [cat_c0, cat_c0 + C) of (B, cat_ctot, cat_H, cat_W) at (cat_y0, cat_x0), or NULL
    int cat_ctot, cat_c0, cat_H, cat_W, cat_y0, cat_x0;
    int B, C, H, W, Mpad6, Mpad7;
    float step;
};

template <int COP>
__global__ __launch_bounds__(256, COP <= 12 ? 3 : 2) void ctx_tail_kernel(const CtxTailParams p, const int tiles_y, const int tiles_x) {
    constexpr int T = 9, KW = 3, CT = 2;
    __shared__ double red[4];
    const int tid = threadIdx.x, tx = tid & 15, tyy = tid >> 4;
    const int tpi = tiles_y * tiles_x;
    const int b = blockIdx.x / tpi;
    const int tr = blockIdx.x - b * tpi;
    const int ty = tr / tiles_x, txx = tr - ty * tiles_x;
    const int wy = ty * 16 + tyy, wx = txx * 64 + tx * 4;
    const int C = p.C, XW = p.W + 2, XHW = (p.H + 2) * XW, HW = p.H * p.W;
    const int nv = min(4, p.W - wx);
    const bool any_ok = wy < p.H && nv > 0;
    unsigned roff[KW];
#pragma unroll
    for (int ky = 0; ky < KW; ++ky) roff[ky] = any_ok ? 4u * (unsigned)((wy + ky) * XW + wx) : OOB;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * C * XHW), 0, C * XHW * 4, RSRC_W3);
    f32x2 acc[4][COP / 2];
#pragma unroll
    for (int j = 0; j < COP / 2; ++j) {
        const f32x2 bv = {(p.b6 && 2 * j < C) ? p.b6[2 * j] : 0.f, (p.b6 && 2 * j + 1 < C) ? p.b6[2 * j + 1] : 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e][j] = bv;
    }
    // ---- dilconv6: the loop of conv_small_f32_kernel<9, COP> ----
    f32x4 xv[CT][T];
    auto load = [&](int c, int s) __attribute__((always_inline)) {
        const int so = c * XHW * 4;
#pragma unroll
        for (int t = 0; t < T; ++t)
            xv[s][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rx, (int)(roff[t / KW] == OOB ? OOB : roff[t / KW] + 4u * (unsigned)(t % KW)), so, 0));
    };
    f32x4 wq[3][COP / 4];
    auto wload = [&](const float* wp, int mpad, int k, auto S) __attribute__((always_inline)) {
        constexpr int sw = decltype(S)::value;
        const float* wr = wp + (size_t)k * mpad;
#pragma unroll
        for (int q = 0; q < COP / 4; ++q) wq[sw][q] = *reinterpret_cast<const f32x4*>(wr + q * 4);
    };
    load(0, 0);
    const int nk = C * T;
    wload(p.wp6, p.Mpad6, 0, ic<0>{});
    wload(p.wp6, p.Mpad6, min(1, nk - 1), ic<1>{});
    for (int c = 0; c < C; c += CT) {
        static_for<0, CT>([&](auto Hh) __attribute__((always_inline)) {
            constexpr int h = decltype(Hh)::value;
            const int cc = c + h;
            if (cc < C) {
                if (cc + 1 < C) load(cc + 1, (h + 1) % CT);
                static_for<0, T>([&](auto Tt) __attribute__((always_inline)) {
                    constexpr int t = decltype(Tt)::value;
                    constexpr int ws = (h * T + t) % 3;
                    const int kn = min(cc * T + t + 2, nk - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    wload(p.wp6, p.Mpad6, kn, ic<(ws + 2) % 3>{});
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < COP / 4; ++q) {
                        const f32x2 wa = {wq[ws][q][0], wq[ws][q][1]}, wb = {wq[ws][q][2], wq[ws][q][3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f32x2 xx = {xv[h][t][e], xv[h][t][e]};
                            acc[e][2 * q] = __builtin_elementwise_fma(xx, wa, acc[e][2 * q]);
                            acc[e][2 * q + 1] = __builtin_elementwise_fma(xx, wb, acc[e][2 * q + 1]);
                        }
                    }
                });
            }
        });
    }
    // the y columns of the four pixels (one 16-byte load per channel): in flight under the 1x1 layer
    const bool act = p.active[b] != 0;
    const __amdgpu_buffer_rsrc_t ry =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (size_t)b * C * HW), 0, C * HW * 4, RSRC_W3);
    const unsigned yo = any_ok ? 4u * (unsigned)(wy * p.W + wx) : OOB;
    f32x4 yq[COP];
#pragma unroll
    for (int c = 0; c < COP; ++c)
        yq[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            ry, (int)((c < C && yo != OOB) ? yo + 4u * (unsigned)(c * HW) : OOB), 0, 0));
    // ---- ReLU, then dilconv7 (1x1, linear): the FMA chain of conv_small_f32_kernel<1, COP> ----
    float t6[4][COP];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < COP / 2; ++j) {
            t6[e][2 * j] = fmaxf(acc[e][j][0], 0.f);
            t6[e][2 * j + 1] = fmaxf(acc[e][j][1], 0.f);
        }
#pragma unroll
    for (int j = 0; j < COP / 2; ++j) {
        const f32x2 bv = {(p.b7 && 2 * j < C) ? p.b7[2 * j] : 0.f, (p.b7 && 2 * j + 1 < C) ? p.b7[2 * j + 1] : 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e][j] = bv;
    }
    wload(p.wp7, p.Mpad7, 0, ic<0>{});
    wload(p.wp7, p.Mpad7, min(1, C - 1), ic<1>{});
    static_for<0, COP>([&](auto Cc) __attribute__((always_inline)) {
        constexpr int c = decltype(Cc)::value;
        constexpr int ws = c % 3;
        if (c < C) {
            __builtin_amdgcn_sched_barrier(0);
            wload(p.wp7, p.Mpad7, min(c + 2, C - 1), ic<(ws + 2) % 3>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < COP / 4; ++q) {
                const f32x2 wa = {wq[ws][q][0], wq[ws][q][1]}, wb = {wq[ws][q][2], wq[ws][q][3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 xx = {t6[e][c], t6[e][c]};
                    acc[e][2 * q] = __builtin_elementwise_fma(xx, wa, acc[e][2 * q]);
                    acc[e][2 * q + 1] = __builtin_elementwise_fma(xx, wb, acc[e][2 * q + 1]);
                }
            }
        }
    });
    // ---- softmax + update per pixel (tail_math.h: the arithmetic of refine_update_kernel) ----
    double nsum = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float r[16], yv[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            r[c] = c < COP ? acc[e][c / 2][c & 1] : 0.f;
            yv[c] = c < COP ? yq[c < COP ? c : 0][e] : 0.f;
        }
        const float ss = refine_pixel<16, float>(C, r, yv, act, p.step);
        nsum += (any_ok && e < nv) ? (double)sqrtf(ss) : 0.0;
#pragma unroll
        for (int c = 0; c < COP; ++c) yq[c][e] = yv[c];
    }
    // stores: y in place and its mirror in the concat buffer (active images only: a stopped image's y stays)
    const bool st = any_ok && act;
    const int CPL = p.cat_H * p.cat_W;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.ycat ? p.ycat + (size_t)b * p.cat_ctot * CPL : nullptr), 0, p.ycat ? p.cat_ctot * CPL * 4 : 0, RSRC_W3);
    const unsigned co0 = 4u * (unsigned)(p.cat_c0 * CPL + (p.cat_y0 + wy) * p.cat_W + p.cat_x0 + wx);
#pragma unroll
    for (int c = 0; c < COP; ++c) {
        if (c >= C) continue;
        const unsigned oy = st ? yo + 4u * (unsigned)(c * HW) : OOB;
        const unsigned oc = st ? co0 + 4u * (unsigned)(c * CPL) : OOB;
        if (nv >= 4) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, yq[c]), ry, (int)oy, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, yq[c]), rc, (int)oc, 0, 0);
        } else {
            const float v0 = yq[c][0], v1 = yq[c][1], v2 = yq[c][2];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v0), ry, (int)oy, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v1), ry, (int)((st && nv > 1) ? oy + 4u : OOB), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v2), ry, (int)((st && nv > 2) ? oy + 8u : OOB), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v0), rc, (int)oc, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v1), rc, (int)((st && nv > 1) ? oc + 4u : OOB), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v2), rc, (int)((st && nv > 2) ? oc + 8u : OOB), 0, 0);
        }
    }
    double d = wave_sum(nsum);
    if ((tid & 63) == 0) red[tid >> 6] = d;
    __syncthreads();
    if (tid == 0) p.partial[(size_t)b * tpi + tr] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

// single-source, plain (no DePool2D input, skip-add, pool, masks, BatchNorm), 'valid' (pad 0) 1x1 / 3x3 request
// between at most 16 channels on either side, any dilation
bool iiseg_conv_small_ok(const ConvParams& p, int KH, int KW) {
    static const int on = getenv("IISEG_CONV_SMALL") ? atoi(getenv("IISEG_CONV_SMALL")) : 1;
    if (!on || p.transposed || p.pad != 0 || p.C2 != 0 || p.C1 > 16 || p.Cout > 16 || p.pool || p.mask_in ||
        p.mask_out || p.bn_mean || !p.out)
        return false;
    // a skip-add only as the START of the FMA chain (no bias): the continuation of another launch's chain -- the
    // y half of the context module's first layer starts from the cached image half (contextmod.py)
    if (p.add && (p.bias || (int64_t)p.Cout * p.AH * p.AW * 4 >= (1ll << 31) - 16)) return false;
    if (!((KH == 1 && KW == 1) || (KH == 3 && KW == 3))) return false;
    if (p.Mpad < 16 || (p.Mpad & 3)) return false;
    if ((int64_t)p.C1 * p.H * p.W * 4 >= (1ll << 31) - 4) return false;      // per-image 32-bit byte offsets
    if ((int64_t)p.B * ((p.OH + 3) / 4) * ((p.OW + 63) / 64) >= (1ll << 31)) return false;
    return true;
}

int iiseg_launch_conv_small(hipStream_t s, const ConvParams& p, int KH) {
    return KH == 3 ? launch_small<9>(s, p) : launch_small<1>(s, p);
}

extern "C" int iiseg_ctx_tail_partials(int32_t H, int32_t W) {
    if (H <= 0 || W <= 0) return 0;
    return ((H + 15) / 16) * ((W + 63) / 64);
}

extern "C" int iiseg_ctx_tail_f32(void* stream, const float* x, const float* wp6, int32_t Mpad6, const float* b6,
                                  const float* wp7, int32_t Mpad7, const float* b7, float* y, const int32_t* active,
                                  double* partial, float* ycat, int32_t cat_ctot, int32_t cat_c0, int32_t cat_H,
                                  int32_t cat_W, int32_t cat_y0, int32_t cat_x0, int32_t B, int32_t C, int32_t H,
                                  int32_t W, float step) {
    if (!x || !wp6 || !wp7 || !y || !active || !partial) return IISEG_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return IISEG_ERR_SHAPE;
    const int cop = (C + 3) / 4 * 4;
    if (C > 16 || cop < 12 || Mpad6 < cop || Mpad7 < cop || (Mpad6 & 3) || (Mpad7 & 3)) return IISEG_ERR_UNSUPPORTED;
    if (((uintptr_t)wp6 | (uintptr_t)wp7) & 15) return IISEG_ERR_ALIGN;
    if ((int64_t)C * (H + 2) * (W + 2) * 4 >= (1ll << 31) - 16) return IISEG_ERR_UNSUPPORTED;
    if (ycat) {
        if (cat_c0 < 0 || cat_c0 + C > cat_ctot || cat_y0 < 0 || cat_x0 < 0 || cat_y0 + H > cat_H || cat_x0 + W > cat_W)
            return IISEG_ERR_SHAPE;
        if ((int64_t)cat_ctot * cat_H * cat_W * 4 >= (1ll << 31) - 16) return IISEG_ERR_UNSUPPORTED;
    }
    const int tiles_y = (H + 15) / 16, tiles_x = (W + 63) / 64;
    if ((int64_t)B * tiles_y * tiles_x >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    CtxTailParams p = {};
    p.x = x; p.wp6 = wp6; p.b6 = b6; p.wp7 = wp7; p.b7 = b7; p.y = y; p.active = active; p.partial = partial;
    p.ycat = ycat; p.cat_ctot = cat_ctot; p.cat_c0 = cat_c0; p.cat_H = cat_H; p.cat_W = cat_W;
    p.cat_y0 = cat_y0; p.cat_x0 = cat_x0;
    p.B = B; p.C = C; p.H = H; p.W = W; p.Mpad6 = Mpad6; p.Mpad7 = Mpad7; p.step = step;
    const dim3 grid(B * tiles_y * tiles_x), block(256);
    if (cop == 12)
        IISEG_LAUNCH((ctx_tail_kernel<12>), grid, block, 0, (hipStream_t)stream, p, tiles_y, tiles_x);
    else
        IISEG_LAUNCH((ctx_tail_kernel<16>), grid, block, 0, (hipStream_t)stream, p, tiles_y, tiles_x);
    return iiseg_check_launch();
}
