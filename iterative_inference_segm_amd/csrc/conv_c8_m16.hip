// Direct 3x3 convolution on bf16 C8 activations for layers with AT MOST 16 OUTPUT CHANNELS (gfx950,
// v_mfma_f32_16x16x32_bf16, fp32 accumulate): the DAE's class-score layer (up_conv1, 64 -> 11 at 224^2:
// models/fcn_up.py:83-86) and FC-DenseNet's growth-rate-16 dense-block layers (BN -> ReLU -> conv 3x3 ->
// 16 new channels appended to the stack: models/FCDenseNet.py:61-146, FC_DenseNet.layers.BN_ReLU_Conv).
// On the 32x32 MFMA of conv_c8_bf16.hip such a layer keeps 11 or 16 of 32 matrix rows busy and, far from
// the matrix pipe's limit anyway, pays the 64-channel kernel's prologue / staging / epilogue for it.
// Here:
//   * M = 16 channels, N = 16 pixels, K = 32 = two taps x 16 input channels per MFMA (taps (2m, 2m + 1)
//     for m = 0..3, tap 8 with a zero second half): 20 MFMAs of 16 cycles per 16-channel k-tile and
//     wave instead of 18 of 32;
//   * 256 threads = 4 waves x 128 pixels, a th x tw tile (<= 512 pixels, shape chosen per window);
//   * LDS 50 KB, < 128 registers: three workgroups and twelve waves per CU.  These layers are bound by
//     the latency of a k-tile's staging (a k-tile is 40 short MFMAs per wave: the next k-tile's loads,
//     issued one k-tile ahead, are not back when they are needed), so what counts is work per staging
//     round trip x workgroups in flight: 512-pixel tiles took 20 % off 256-pixel ones; two register sets
//     with the loads issued two k-tiles ahead were built and measured SLOWER with the per-element staging
//     (0.154 vs 0.132 ms on the class-score layer: 190-210 registers, two workgroups per CU instead of
//     three); with DePool2D staged by pooled positions a set is 12 registers and the two-ahead form is what
//     runs (140 registers, still three workgroups; 0.1250 -> 0.1220 ms over three alternating repeats: the
//     layer issues 1140 vector instructions per 160 MFMAs and wave, that is what bounds it), and for the
//     BatchNorm + ReLU mode as well (24 registers per set, 163-167 in all: FC-DenseNet103 forward 6.63 -> 6.46 ms
//     of kernels over three alternating repeats);
//   * input staging per 8-channel half (the half is a compile-time constant of a piece): LDS-DMA, or
//     through registers for DePool2D (up chunk + 8 mask bytes, layers/mylayers.py:88-115) and for
//     BatchNorm + ReLU applied on the way in (per-channel scale / shift read as scalars; the padding
//     ring stays exact zeros, as a zero-padded convolution of the normalised map has it);
//   * output: fp32 NCHW (class scores), or bf16 C8 into a 16-channel slice of a wider C8 tensor (the
//     dense block's stack: no concat copy).
// The accumulators start from the bias; every output is one fixed-order sum (k-tile, tap pair), so
// results do not depend on the tile shape or the window a launch covers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "iiseg.h"
#include "common.h"
#include "conv_common.h"
#include "c8_common.h"

using namespace iiseg;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int RSRC_W3 = 0x00027000;
constexpr unsigned OOB = 0x80000000u;
constexpr int NB = 8;              // 16-pixel blocks per wave: 4 waves x 128 pixels = 512-pixel tiles
constexpr int NPC = 3;             // 256-chunk staging pieces per 8-channel half
constexpr int HCAP = 640;          // patch chunks per half (whole 64-chunk DMA pieces: 256 + 256 + 128)
constexpr int WROWS = 18 * 16;     // weight chunks per k-tile: (tap, half) x 16 channels
constexpr int WCAP = 320;          // ... rounded up to whole 64-chunk DMA pieces

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, RSRC_W3);
}
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

struct M16Params {
    const void* x1;               // C8 input (B, C1/8, H, W, 8); UNPOOL: `up` (B, C1/8, h2, w2, 8)
    const unsigned char* mask_in; // UNPOOL: (B, C1/8, h2, w2, 8) mask bytes
    const void* wp;               // Wp16[kt][tap][h][Mpad][8] (iiseg_conv_halo_bf16_pack, Mpad = 32)
    const float* bias;
    const float* bn_a;            // BNRELU: x <- max(a[c] x + b[c], 0), one pair per input channel
    const float* bn_b;
    void* out;
    int B, C1, H, W, h2, w2;
    int in_c8tot;                 // chunk planes per image of x1 (>= C1 / 8: the first channels of a wider tensor)
    int Cout, OH, OW, oy0, ox0, pad;
    int nkt, Mpad;
    int out_ctot, out_c0, out_H, out_W, out_y0, out_x0;
    int relu;
    int n_ptiles, tiles_y, tiles_x, th, tw;
    unsigned pw_magic, tw_magic;
    // split-K (small maps: few tiles, long channel loops), two launches: phase 1 -- workgroup (tile, slice) sums
    // k-tiles [nkt slice / nsplit, nkt (slice + 1) / nsplit) from zero (slice 0: from the bias) into an fp32 slab;
    // phase 2 -- one workgroup per tile adds the slabs in slice order and runs the epilogue.  phase 0: no split
    int nsplit, phase;
    float* slabs;                 // [n_ptiles][nsplit][NB * 256 * 4]
    // batch statistics of the 16 produced channels (bf16 C8 output): per-tile sums / sums of squares of the
    // STORED values in double, stat_ws[chunk 0 / 1][tile][16]; bn_stats_c8_final_kernel adds the tiles in order
    double* stat_ws;
};

enum { M16_PLAIN = 0, M16_UNPOOL = 1, M16_BNRELU = 2 };

template <int MODE, bool OUTF32>
__global__ __launch_bounds__(256, 3) void conv_c8_m16_kernel(const M16Params p) {
    // ONE LDS array: weight ring Ws[2][WCAP], patch ring Ps[2][2 halves][HCAP]
    __shared__ __attribute__((aligned(16))) uint4 smem[2 * WCAP + 4 * HCAP];
    uint4 (*Ws)[WCAP] = reinterpret_cast<uint4 (*)[WCAP]>(smem);
    uint4 (*Ps)[2][HCAP] = reinterpret_cast<uint4 (*)[2][HCAP]>(smem + 2 * WCAP);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int HW = p.H * p.W, hw2 = p.h2 * p.w2;
    const int CC = p.C1 >> 3;

    // ---- tile: th x tw pixels of one image, blocks dealt round-robin to the XCDs in runs ------------
    int pt, mt;
    const int S = p.nsplit;
    int sidx = 0;
    if (p.phase == 1) {
        pt = (int)blockIdx.x / S;
        sidx = (int)blockIdx.x - pt * S;
    } else {
        tile_of_block(blockIdx.x, gridDim.x, p.n_ptiles, 1, pt, mt);
    }
    const int tpi = p.tiles_y * p.tiles_x;
    const int tb = pt / tpi;
    const int tr = pt - tb * tpi;
    const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
    const int wy0 = ty * p.th, wx0 = tx * p.tw;
    const int PWs = p.tw + 2;
    const int half = (p.th + 2) * PWs;           // patch chunks of one 8-channel half (<= HCAP)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // piece i of a half (chunks [256 i + 64 wave, + 64)): only if the patch reaches it
    bool piece[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) piece[i] = i * 256 + wave_u * 64 < half && i * 256 + wave_u * 64 < HCAP;

    // ---- patch staging offsets: chunk e = i * 256 + tid of a half -> (patch row, patch column) ------
    unsigned voff[NPC];                     // byte offset in chunk plane 0 of the image, or OOB
    unsigned voffm[MODE == M16_UNPOOL ? NPC : 1];
    int bsel[MODE == M16_UNPOOL ? NPC : 1];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int e = i * 256 + tid;
        const int prow = (int)(((unsigned)e * p.pw_magic) >> 20), pcol = e - prow * PWs;     // e / PWs
        const int iy = p.oy0 + wy0 - p.pad + prow, ix = p.ox0 + wx0 - p.pad + pcol;
        bool ok = e < half && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        if constexpr (MODE == M16_UNPOOL) {
            ok = ok && iy < 2 * p.h2 && ix < 2 * p.w2;     // DePool2D: only where pooling windows are
            const unsigned pq = (unsigned)((iy >> 1) * p.w2 + (ix >> 1));
            voff[i] = ok ? pq * 16u : OOB;
            voffm[i] = ok ? pq * 8u : OOB;
            bsel[i] = ((iy & 1) << 1) | (ix & 1);
        } else {
            voff[i] = ok ? (unsigned)(iy * p.W + ix) * 16u : OOB;
        }
    }
    // M16_UNPOOL: staged by POOLED positions (one `up` chunk + its 8 mask bytes per 2x2 window, up to four patch
    // elements written from them) instead of per patch element: a third of the loads of the layer that is bound
    // by exactly those (the class-score layer: 42 % of its wave cycles parked on them, profiles/r04_pmc_layers.md).
    // One pooled position per thread and half: (th + 2) / 2 + 1 rows x (tw + 2) / 2 + 1 columns <= 256 (host).
    unsigned qoff = OOB, qoffm = OOB;
    int qlds[4] = {-1, -1, -1, -1};
    if constexpr (MODE == M16_UNPOOL) {
        const int iy0 = p.oy0 + wy0 - p.pad, ix0 = p.ox0 + wx0 - p.pad;      // patch origin, input coordinates
        const int PH = p.th + 2;
        const int Y20 = iy0 >> 1, X20 = ix0 >> 1;                            // (arithmetic shifts: floor)
        const int QH = ((iy0 + PH - 1) >> 1) - Y20 + 1, QW = ((ix0 + PWs - 1) >> 1) - X20 + 1;
        const unsigned qw_magic = ((1u << 20) + (unsigned)QW - 1u) / (unsigned)QW;
        const int qy = (int)(((unsigned)tid * qw_magic) >> 20), qx = tid - qy * QW;
        const int Y2 = Y20 + qy, X2 = X20 + qx;
        const bool in = tid < QH * QW;
        const bool valid = in && (unsigned)Y2 < (unsigned)p.h2 && (unsigned)X2 < (unsigned)p.w2;
        qoff = valid ? (unsigned)(Y2 * p.w2 + X2) * 16u : OOB;
        qoffm = valid ? (unsigned)(Y2 * p.w2 + X2) * 8u : OOB;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const int py = 2 * Y2 + (sl >> 1) - iy0, px = 2 * X2 + (sl & 1) - ix0;
            qlds[sl] = (in && (unsigned)py < (unsigned)PH && (unsigned)px < (unsigned)PWs) ? py * PWs + px : -1;
        }
    }
    const int plane = MODE == M16_UNPOOL ? hw2 : HW;
    const char* base1 = (const char*)p.x1 + (size_t)tb * p.in_c8tot * plane * 16;
    const unsigned n1 = (unsigned)(CC * plane) * 16u;
    const __amdgpu_buffer_rsrc_t r_x1 = mk_rsrc(base1, n1);
    const i32x4s s_x1 = mk_srsrc(base1, n1);
    const __amdgpu_buffer_rsrc_t r_m =
        mk_rsrc(MODE == M16_UNPOOL ? p.mask_in + (size_t)tb * CC * hw2 * 8 : nullptr,
                MODE == M16_UNPOOL ? (unsigned)(CC * hw2) * 8u : 0u);
    const i32x4s s_w = mk_srsrc(p.wp, (unsigned)(p.nkt * 18 * p.Mpad) * 16u);
    const unsigned lds_w = __builtin_amdgcn_readfirstlane(lds_addr(&Ws[0][0]) + (unsigned)wave * 1024u);
    const unsigned lds_p = __builtin_amdgcn_readfirstlane(lds_addr(&Ps[0][0][0]) + (unsigned)wave * 1024u);
    // weights of a k-tile: chunk f = j * 256 + tid -> row (tap, half) f / 16, channel f % 16
    unsigned woff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = j * 256 + tid;
        woff[j] = f < WROWS ? 16u * (unsigned)((f >> 4) * p.Mpad + (f & 15)) : OOB;
    }

    // ---- this lane's pixels: block k of the wave = pixels (wave 64 + 16 k + l15) of the tile, row-major
    int bpos[NB], ey[NB], ex[NB];
    bool eok[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int n = (wave * NB + k) * 16 + l15;
        int ly = (int)(((unsigned)n * p.tw_magic) >> 20), lx = n - ly * p.tw;               // n / tw
        const bool inb = ly < p.th;
        if (!inb) { ly = 0; lx = 0; }
        ey[k] = wy0 + ly; ex[k] = wx0 + lx;
        eok[k] = inb && ey[k] < p.OH && ex[k] < p.OW;
        bpos[k] = ly * PWs + lx;
    }
    // K = 32 of an MFMA: lane group g -> (tap 2m + (g >> 1), half g & 1) in step m < 4; in step 4 the
    // groups 0, 1 hold tap 8 and the groups 2, 3 a zero A operand
    const int hsel = g & 1, tsel = g >> 1;
    int aoff[5], boff[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int tap = m < 4 ? 2 * m + tsel : 8;
        const int ky = tap / 3, kx = tap - 3 * ky;
        aoff[m] = (tap * 2 + hsel) * 16 + l15;
        boff[m] = hsel * HCAP + ky * PWs + kx;
    }

    // accumulators start from the bias: C/D row = 4 g + q (channel), column = l15 (pixel)
    f32x4 acc[NB];
    {
        const __amdgpu_buffer_rsrc_t r_bias = mk_rsrc(p.bias, p.bias ? (unsigned)p.Cout * 4u : 0u);
        f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            r_bias, (int)(16u * (unsigned)g), 0, 0));
        if (sidx != 0) bv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NB; ++k) acc[k] = bv;
    }

    // ---- staging of one k-tile (chunks KC, KC + 1 of the input; packed weight k-tile WT) ------------
    auto dma_w = [&](int wt, int buf) __attribute__((always_inline)) {
        const unsigned so_w = __builtin_amdgcn_readfirstlane((unsigned)(wt * 18 * p.Mpad) * 16u);
        dma16(s_w, lds_w + (unsigned)(buf * WCAP) * 16u, woff[0], so_w);
        if (wave_u == 0) dma16(s_w, lds_w + (unsigned)(buf * WCAP + 256) * 16u, woff[1], so_w);
    };
    auto dma_x = [&](int kc, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((kc + h) * plane) * 16u);
#pragma unroll
            for (int i = 0; i < NPC; ++i)
                if (piece[i]) dma16(s_x1, lds_p + (unsigned)((buf * 2 + h) * HCAP + i * 256) * 16u, voff[i], so);
        }
    };
    // Register staging (UNPOOL / BNRELU): loaded here, transformed and written to LDS after the MFMAs.  TWO k-tiles
    // are kept in registers (DePool2D: one `up` chunk + 8 mask bytes per thread and half = 12 registers per set;
    // BatchNorm + ReLU: up to three chunks per half = 24): the loads of k-tile kt + 2 are issued at the top of step
    // kt, so when step kt + 1 expands them into the LDS they are a whole step old -- the workgroup-wide wait in
    // front of the barrier has already covered them and there is no wait between the MFMAs and the LDS writes.
    // (Round 4 measured the same idea slower with DePool2D staged per patch element: 48 registers per set, a
    // workgroup less per CU.  Now 140-144 registers, still three workgroups.)
    constexpr int NRP = MODE == M16_UNPOOL ? 1 : NPC;
    u32x4 xu[2][2][NRP];
    u32x2 xm[2][2];
    auto load_x = [&](auto SET, int kc) __attribute__((always_inline)) {
        constexpr int s = decltype(SET)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int so = (int)((unsigned)((kc + h) * plane) * 16u);
            if constexpr (MODE == M16_UNPOOL) {
                xu[s][h][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_x1, (int)qoff, so, 0));
                xm[s][h] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r_m, (int)qoffm, so >> 1, 0));
            } else {
#pragma unroll
                for (int i = 0; i < NRP; ++i) {
                    if (!piece[i]) continue;
                    xu[s][h][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_x1, (int)voff[i], so, 0));
                }
            }
        }
    };
    auto store_x = [&](auto SET, int kc, int buf) __attribute__((always_inline)) {
        constexpr int s = decltype(SET)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (MODE == M16_UNPOOL) {
                // the four pixels of the thread's pooling window: bit sl of byte j of the mask pair says
                // pre == pooled for channel j there (layers/mylayers.py:111-114)
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    const unsigned t0 = (xm[s][h][0] >> sl) & 0x01010101u;
                    const unsigned t1 = (xm[s][h][1] >> sl) & 0x01010101u;
                    const unsigned b0 = (t0 << 8) - t0, b1 = (t1 << 8) - t1;
                    uint4 v;
                    v.x = xu[s][h][0][0] & __builtin_amdgcn_perm(b0, b0, 0x01010000u);
                    v.y = xu[s][h][0][1] & __builtin_amdgcn_perm(b0, b0, 0x03030202u);
                    v.z = xu[s][h][0][2] & __builtin_amdgcn_perm(b1, b1, 0x01010000u);
                    v.w = xu[s][h][0][3] & __builtin_amdgcn_perm(b1, b1, 0x03030202u);
                    if (qlds[sl] >= 0) Ps[buf][h][qlds[sl]] = v;
                }
            } else {
                // the 8 channels of this half: wave-uniform addresses, read through the CONSTANT address space so
                // that they ARE scalar loads (s_load_dwordx8, lgkmcnt): through the generic pointers hipcc made them
                // four global_load_dwordx4 with a full vmcnt wait in front of the LDS writes of every k-tile (the
                // (a, b) table is written by an earlier kernel, never here).  FC-DenseNet103 forward 7.78 -> 7.32 ms;
                // issued earlier still (with the activation loads, live across the MFMAs) they spill scalar
                // registers: 7.32-7.62 ms
                float sa[8], sb[8];
                typedef const float __attribute__((address_space(4))) cfloat;
                const int c0 = __builtin_amdgcn_readfirstlane((kc + h) * 8);
                cfloat* ca = (cfloat*)(uintptr_t)p.bn_a;
                cfloat* cb = (cfloat*)(uintptr_t)p.bn_b;
#pragma unroll
                for (int q = 0; q < 8; ++q) { sa[q] = ca[c0 + q]; sb[q] = cb[c0 + q]; }
                // (an FMA takes one scalar operand: the shifts go to vector registers ONCE per half -- left to
                // itself hipcc re-materialises them with a v_mov in front of every FMA of every piece: 144 moves)
                float vb[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_mov_b32 %0, %1" : "=v"(vb[q]) : "s"(sb[q]));
#pragma unroll
                for (int i = 0; i < NRP; ++i) {
                    if (!piece[i]) continue;
                    // BatchNorm + ReLU of the stored bf16 values, rounded to bf16 once more; chunks of
                    // the zero-padding ring (out-of-range offset: the load returned zeros) stay zero
                    const bool inside = voff[i] != OOB;
                    unsigned w[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned u = xu[s][h][i][q];
                        const float lo = fmaxf(__builtin_fmaf(bf_lo(u), sa[2 * q], vb[2 * q]), 0.f);
                        const float hi = fmaxf(__builtin_fmaf(bf_hi(u), sa[2 * q + 1], vb[2 * q + 1]), 0.f);
                        w[q] = inside ? pack_bf16(lo, hi) : 0u;
                    }
                    if (i * 256 + tid < half) Ps[buf][h][i * 256 + tid] = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
    };

    const int kt0 = p.phase == 1 ? p.nkt * sidx / S : 0;
    const int kt1 = p.phase == 1 ? p.nkt * (sidx + 1) / S : (p.phase == 2 ? 0 : p.nkt);
    auto mfma_tile = [&](int buf) __attribute__((always_inline)) {
        const uint4* Wb = &Ws[buf][0];
        const uint4* Pb = &Ps[buf][0][0];
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            uint4 a = Wb[aoff[m]];
            if (m == 4 && g >= 2) a = make_uint4(0u, 0u, 0u, 0u);
            uint4 b[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) b[k] = Pb[boff[m] + bpos[k]];
#pragma unroll
            for (int k = 0; k < NB; ++k)
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                                   __builtin_bit_cast(bf16x8, b[k]), acc[k], 0, 0, 0);
        }
    };
    if constexpr (MODE != M16_PLAIN) {
        typedef std::integral_constant<int, 0> S0;
        typedef std::integral_constant<int, 1> S1;
        if (p.phase != 2 && kt0 < kt1) {
            load_x(S0{}, 2 * kt0);
            dma_w(kt0, 0);
            if (kt0 + 1 < kt1) load_x(S1{}, 2 * (kt0 + 1));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            store_x(S0{}, 2 * kt0, 0);
        }
        // step kt: k-tile kt is in LDS buffer SET, k-tile kt + 1 in register set SET ^ 1 (loaded or in flight)
        auto step = [&](auto SET, int kt) __attribute__((always_inline)) {
            constexpr int s = decltype(SET)::value;
            // own DMA pieces, register loads and LDS writes retired, then the barrier publishes them and tells that
            // every wave is done reading what the previous step read
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 1 < kt1) dma_w(kt + 1, s ^ 1);
            if (kt + 2 < kt1) load_x(SET, 2 * (kt + 2));
            mfma_tile(s);
            if (kt + 1 < kt1) store_x(std::integral_constant<int, s ^ 1>{}, 2 * (kt + 1), s ^ 1);
        };
        for (int kt = kt0; kt < kt1; kt += 2) {
            step(S0{}, kt);
            if (kt + 1 < kt1) step(S1{}, kt + 1);
        }
    } else {
        if (p.phase != 2) {
            dma_x(2 * kt0, 0);
            dma_w(kt0, 0);
        }
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            // own DMA pieces retired, then the barrier publishes them and tells that every wave is done reading
            // what the previous step read
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 1 < kt1) {
                dma_x(2 * (kt + 1), buf ^ 1);
                dma_w(kt + 1, buf ^ 1);
            }
            mfma_tile(buf);
        }
    }

    if (p.phase == 1) {
        // partial sums of this slice -> its slab (only the lanes that own an output pixel)
        float* slab = p.slabs + ((size_t)pt * S + sidx) * (NB * 256 * 4);
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (eok[k]) *reinterpret_cast<f32x4*>(slab + (k * 256 + tid) * 4) = acc[k];
        return;
    }
    if (p.phase == 2) {
        // the slabs in slice order: a fixed association of the channel sum
        const float* slab0 = p.slabs + (size_t)pt * S * (NB * 256 * 4);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if (!eok[k]) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(slab0 + (k * 256 + tid) * 4);
            for (int q = 1; q < S; ++q)
                v += *reinterpret_cast<const f32x4*>(slab0 + (size_t)q * (NB * 256 * 4) + (k * 256 + tid) * 4);
            acc[k] = v;
        }
    }

    // ---- epilogue --------------------------------------------------------------------------------
    const int OPL = p.out_H * p.out_W;
    if constexpr (OUTF32) {
        // fp32 NCHW (B, out_ctot, out_H, out_W): channel 4 g + q of pixel l15 -- 64-byte runs per store
        const __amdgpu_buffer_rsrc_t r_out =
            mk_rsrc((const char*)p.out + (size_t)tb * p.out_ctot * OPL * 4, (unsigned)(p.out_ctot * OPL) * 4u);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const unsigned o0 = (unsigned)(p.out_c0 * OPL + (p.out_y0 + ey[k]) * p.out_W + p.out_x0 + ex[k]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = 4 * g + q;
                float v = acc[k][q];
                if (p.relu) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(
                    __builtin_bit_cast(int, v), r_out,
                    (int)((eok[k] && co < p.Cout) ? 4u * (o0 + (unsigned)(co * OPL)) : OOB), 0, 0);
            }
        }
    } else {
        // bf16 C8: the 16 channels are chunks (out_c0 / 8) + (0, 1) of the (B, out_ctot / 8, out_H, out_W, 8)
        // tensor; lane group g holds half (g & 1) of chunk (g >> 1); channels past Cout are exact zeros
        // (zero weight rows, no bias)
        const int oct8 = p.out_ctot >> 3;
        const __amdgpu_buffer_rsrc_t r_out =
            mk_rsrc((const char*)p.out + (size_t)tb * oct8 * OPL * 16, (unsigned)(oct8 * OPL) * 16u);
        const unsigned cplane = (unsigned)(((p.out_c0 >> 3) + (g >> 1)) * OPL) * 16u + 8u * (unsigned)(g & 1);
        double st_s[4] = {0.0, 0.0, 0.0, 0.0}, st_ss[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const unsigned opix = (unsigned)((p.out_y0 + ey[k]) * p.out_W + p.out_x0 + ex[k]);
            f32x4 v = acc[k];
            if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            u32x2 w2;
            w2[0] = pack_bf16(v[0], v[1]); w2[1] = pack_bf16(v[2], v[3]);
            __builtin_amdgcn_raw_buffer_store_b64(w2, r_out, (int)(eok[k] ? opix * 16u + cplane : OOB), 0, 0);
            if (p.stat_ws && eok[k]) {
                const float sv[4] = {bf_lo(w2[0]), bf_hi(w2[0]), bf_lo(w2[1]), bf_hi(w2[1])};
#pragma unroll
                for (int q = 0; q < 4; ++q) { st_s[q] += sv[q]; st_ss[q] += (double)sv[q] * sv[q]; }
            }
        }
        if (p.stat_ws) {
            // channel 4 g + q: the 16 lanes of a group in a fixed-order butterfly, then the four waves in order
            __shared__ double red[4][4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    st_s[q] += __shfl_xor(st_s[q], o, 64);
                    st_ss[q] += __shfl_xor(st_ss[q], o, 64);
                }
            if (l15 == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { red[wave][g][q] = st_s[q]; red[wave][g][4 + q] = st_ss[q]; }
            }
            __syncthreads();
            if (tid < 32) {
                const int c = tid & 15, which = tid >> 4;            // channel, 0: sum / 1: sum of squares
                const int gg = c >> 2, qq = c & 3;
                double t = red[0][gg][which * 4 + qq];
                t += red[1][gg][which * 4 + qq];
                t += red[2][gg][which * 4 + qq];
                t += red[3][gg][which * 4 + qq];
                p.stat_ws[((size_t)(c >> 3) * p.n_ptiles + pt) * 16 + which * 8 + (c & 7)] = t;
            }
        }
    }
}

// y = max(a x + b, 0) coefficients of one BatchNorm layer over the first n channels of a stack:
// a = gamma inv_std, b = beta - mean a (models/FCDenseNet.py BN_ReLU_Conv; batch statistics, P10)
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ beta, const float* __restrict__ gamma,
                                                      const float* __restrict__ mean, const float* __restrict__ inv_std,
                                                      float* __restrict__ a, float* __restrict__ b, int n) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < n) {
        const float s = gamma[c] * inv_std[c];
        a[c] = s;
        b[c] = beta[c] - mean[c] * s;
    }
}

// Batch statistics (mean, 1 / sqrt(var + eps), biased variance) of channels [c0, c0 + n) of a bf16 C8
// tensor (B, Ctot / 8, H, W, 8); c0, n multiples of 8.  Two deterministic stages: workgroup (chunk,
// slice) sums its share of the pixels of all images in double (per thread, then a fixed-order tree) into
// ws[chunk][slice][16]; the second kernel adds the slices in order.
constexpr int BN_SLICES = 128;

__global__ __launch_bounds__(256) void bn_stats_c8_kernel(const uint4* __restrict__ x, int B, int C8tot, int c8_0,
                                                          int HW, double* __restrict__ ws) {
    __shared__ double red[16][256];
    const int c8 = c8_0 + blockIdx.x, slice = blockIdx.y;
    double s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.0; ss[j] = 0.0; }
    const int64_t total = (int64_t)B * HW;
    for (int64_t t = (int64_t)slice * 256 + threadIdx.x; t < total; t += (int64_t)BN_SLICES * 256) {
        const int b = (int)(t / HW), i = (int)(t - (int64_t)b * HW);
        const uint4 u = x[((size_t)b * C8tot + c8) * HW + i];
        const float v[8] = {bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y), bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] += v[j]; ss[j] += (double)v[j] * v[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[j][threadIdx.x] = s[j]; red[8 + j][threadIdx.x] = ss[j]; }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int j = 0; j < 16; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x < 16) ws[((size_t)blockIdx.x * BN_SLICES + slice) * 16 + threadIdx.x] = red[threadIdx.x][0];
}

// Stage 2: one workgroup per chunk of 8 channels; thread (subset u = tid / 16, value j = tid % 16) adds slices
// u, u + 64, ... in order, then the 64 subsets are added in order -- a fixed association whatever nslices is.
// fold_*: (optional, single-chunk-pair launches of the dense-block layers) after the statistics of the new
// channels are in place, the (a, b) pair of the NEXT consumer's BatchNorm over its first fold_n channels
// (bn_fold_kernel's arithmetic) by workgroup 0 -- saves that launch.  Needs both chunks' statistics, so the
// workgroups of a folding launch are ONE workgroup looping over the chunks.
__global__ __launch_bounds__(1024) void bn_stats_c8_final_kernel(const double* __restrict__ ws, int nslices, int c8_0,
                                                                 int nchunks, double count, double eps,
                                                                 float* __restrict__ mean, float* __restrict__ inv_std,
                                                                 const float* __restrict__ fold_beta,
                                                                 const float* __restrict__ fold_gamma,
                                                                 float* __restrict__ fold_a, float* __restrict__ fold_b,
                                                                 int fold_n) {
    constexpr int NSUB = 64;
    __shared__ double part[NSUB][16];
    const int tid = threadIdx.x, j = tid & 15, u = tid >> 4;
    const int ch0 = fold_a ? 0 : blockIdx.x, ch1 = fold_a ? nchunks : blockIdx.x + 1;
    for (int ch = ch0; ch < ch1; ++ch) {
        const double* w = ws + (size_t)ch * nslices * 16 + j;
        double t = 0.0;
        int k = u;
        // (eight loads in flight, added in slice order)
        for (; k + 7 * NSUB < nslices; k += 8 * NSUB) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = w[(size_t)(k + q * NSUB) * 16];
#pragma unroll
            for (int q = 0; q < 8; ++q) t += v[q];
        }
        for (; k < nslices; k += NSUB) t += w[(size_t)k * 16];
        part[u][j] = t;
        __syncthreads();
        if (tid < 16) {
            double sm = 0.0;
            for (int q = 0; q < NSUB; ++q) sm += part[q][tid];
            part[0][tid] = sm;
        }
        __syncthreads();
        if (tid < 8) {
            const double m = part[0][tid] / count;
            double var = part[0][8 + tid] / count - m * m;
            if (var < 0.0) var = 0.0;
            mean[(c8_0 + ch) * 8 + tid] = (float)m;
            inv_std[(c8_0 + ch) * 8 + tid] = (float)(1.0 / sqrt(var + eps));
        }
        __syncthreads();
    }
    if (fold_a) {
        __threadfence_block();
        for (int c = tid; c < fold_n; c += 1024) {
            const float sc = fold_gamma[c] * inv_std[c];
            fold_a[c] = sc;
            fold_b[c] = fold_beta[c] - mean[c] * sc;
        }
    }
}

int m16_check(const iiseg_conv_desc* d) {
    if (!d) return IISEG_ERR_NULL;
    if (d->KH != 3 || d->KW != 3 || d->dil != 1 || (d->flags & (IISEG_CONV_TRANSPOSED2 | IISEG_CONV_X3)))
        return IISEG_ERR_UNSUPPORTED;
    if (d->B <= 0 || d->C1 <= 0 || d->C2 != 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 || d->Cout > 16 ||
        d->pad < 0 || d->OH <= 0 || d->OW <= 0 || d->oy0 < 0 || d->ox0 < 0)
        return d && d->Cout > 16 ? IISEG_ERR_UNSUPPORTED : IISEG_ERR_SHAPE;
    if (d->C1 % 16) return IISEG_ERR_UNSUPPORTED;
    const int fullH = d->H + 2 * d->pad - 2, fullW = d->W + 2 * d->pad - 2;
    if (d->oy0 + d->OH > fullH || d->ox0 + d->OW > fullW) return IISEG_ERR_SHAPE;
    if (d->out_ctot != 0 && (d->out_c0 < 0 || d->out_c0 + d->Cout > d->out_ctot)) return IISEG_ERR_SHAPE;
    if (d->out_H != 0 && (d->out_y0 < 0 || d->out_x0 < 0 || d->out_y0 + d->OH > d->out_H ||
                          d->out_x0 + d->OW > d->out_W))
        return IISEG_ERR_SHAPE;
    // one image of every tensor is addressed with 32-bit byte offsets
    if ((int64_t)d->C1 * d->H * d->W * 2 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    const int64_t octot = d->out_ctot ? d->out_ctot : 16;
    const int64_t opl = d->out_H ? (int64_t)d->out_H * d->out_W : (int64_t)d->OH * d->OW;
    if ((octot + 16) * opl * 4 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    if ((int64_t)(d->C1 / 16) * 18 * 32 * 16 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    return IISEG_OK;
}


// tile shape of a launch and its split-K factor.  The factor depends on the tiles PER IMAGE and the channel
// count only -- never on the batch -- so an image gets the same association of its channel sum alone and
// in a batch: 16 / tiles-per-image slices (one 7^2 or 14^2 map of FC-DenseNet's deep blocks is ONE tile with
// 41 .. 67 k-tiles in sequence, 32 workgroups on 256 CUs at batch 32), at least four k-tiles per slice.
void m16_tiling(const iiseg_conv_desc* d, int* th, int* tw, int* nsplit) {
    int64_t tiles;
    rect_shape(d->OH, d->OW, NB * 64, HCAP, false, th, tw, &tiles);
    static const char* shape_env = getenv("IISEG_M16_SHAPE");           // "th,tw": timing experiments
    if (shape_env) {
        int a = 0, b = 0;
        if (sscanf(shape_env, "%d,%d", &a, &b) == 2 && a > 0 && b > 0 && a * b <= NB * 64 && (a + 2) * (b + 2) <= HCAP) {
            *th = a; *tw = b;
        }
    }
    static const int split_env = getenv("IISEG_M16_SPLITK") ? atoi(getenv("IISEG_M16_SPLITK")) : 1;
    const int tpi = ((d->OH + *th - 1) / *th) * ((d->OW + *tw - 1) / *tw);
    int S = 16 / tpi;
    const int nkt = d->C1 / 16;
    if (S > nkt / 4) S = nkt / 4;
    *nsplit = (split_env && S > 1) ? S : 1;
}

}  // namespace

extern "C" int iiseg_conv_c8_m16_supported(const iiseg_conv_desc* d) { return m16_check(d) == IISEG_OK ? 1 : 0; }

extern "C" int64_t iiseg_conv_c8_m16_workspace_bytes(const iiseg_conv_desc* d) {
    if (m16_check(d) != IISEG_OK) return 0;
    int th, tw, S;
    m16_tiling(d, &th, &tw, &S);
    const int64_t tiles = (int64_t)d->B * ((d->OH + th - 1) / th) * ((d->OW + tw - 1) / tw);
    // per-tile statistics of the produced channels (2 chunks x 16 doubles), then the split-K slabs
    return tiles * 2 * 16 * 8 + (S > 1 ? tiles * S * (NB * 256 * 4) * 4 : 0);
}

extern "C" int iiseg_conv_c8_m16(void* stream, const iiseg_conv_desc* d, const void* x1, int in_ctot,
                                 const uint8_t* mask_in, const float* bn_a, const float* bn_b,
                                 const void* wp16, const float* bias, void* out, int out_kind) {
    return iiseg_conv_c8_m16_ws(stream, d, x1, in_ctot, mask_in, bn_a, bn_b, wp16, bias, out, out_kind, nullptr, 0,
                                nullptr, nullptr, 0.0, nullptr, nullptr, nullptr, nullptr, 0);
}

extern "C" int iiseg_conv_c8_m16_ws(void* stream, const iiseg_conv_desc* d, const void* x1, int in_ctot,
                                    const uint8_t* mask_in, const float* bn_a, const float* bn_b,
                                    const void* wp16, const float* bias, void* out, int out_kind,
                                    void* workspace, int64_t workspace_bytes, float* stat_mean,
                                    float* stat_inv_std, double stat_eps, const float* fold_beta,
                                    const float* fold_gamma, float* fold_a, float* fold_b, int fold_n) {
    const int st = m16_check(d);
    if (st) return st;
    if (in_ctot == 0) in_ctot = d->C1;
    if (in_ctot < d->C1 || in_ctot % 8) return IISEG_ERR_SHAPE;
    if ((d->flags & IISEG_CONV_UNPOOL) && in_ctot != d->C1) return IISEG_ERR_UNSUPPORTED;
    if ((int64_t)in_ctot * d->H * d->W * 2 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    if (!x1 || !wp16 || !out) return IISEG_ERR_NULL;
    const bool unpool = (d->flags & IISEG_CONV_UNPOOL) != 0;
    if (unpool != (mask_in != nullptr)) return IISEG_ERR_NULL;
    if ((bn_a != nullptr) != (bn_b != nullptr) || (bn_a && unpool)) return IISEG_ERR_UNSUPPORTED;
    if ((uintptr_t)wp16 & 15) return IISEG_ERR_ALIGN;
    if (out_kind != 1 && out_kind != 3) return IISEG_ERR_UNSUPPORTED;
    const int octot = d->out_ctot ? d->out_ctot : (out_kind == 3 ? d->Cout : 16);
    if (out_kind == 1 && (octot % 16 || d->out_c0 % 16)) return IISEG_ERR_UNSUPPORTED;
    M16Params p = {};
    p.x1 = x1; p.mask_in = mask_in; p.wp = wp16; p.bias = bias; p.bn_a = bn_a; p.bn_b = bn_b; p.out = out;
    p.B = d->B; p.C1 = d->C1; p.H = d->H; p.W = d->W; p.h2 = d->H / 2; p.w2 = d->W / 2;
    p.in_c8tot = in_ctot / 8;
    p.Cout = d->Cout; p.OH = d->OH; p.OW = d->OW; p.oy0 = d->oy0; p.ox0 = d->ox0; p.pad = d->pad;
    p.nkt = d->C1 / 16;
    p.Mpad = 32;                        // iiseg_conv_halo_bf16_pack's padding for Cout <= 32
    p.out_ctot = octot;
    p.out_c0 = d->out_ctot ? d->out_c0 : 0;
    p.out_H = d->out_H ? d->out_H : d->OH;
    p.out_W = d->out_H ? d->out_W : d->OW;
    p.out_y0 = d->out_H ? d->out_y0 : 0;
    p.out_x0 = d->out_H ? d->out_x0 : 0;
    p.relu = (d->flags & IISEG_CONV_RELU) ? 1 : 0;
    m16_tiling(d, &p.th, &p.tw, &p.nsplit);
    p.tiles_y = (d->OH + p.th - 1) / p.th;
    p.tiles_x = (d->OW + p.tw - 1) / p.tw;
    p.n_ptiles = d->B * p.tiles_y * p.tiles_x;
    p.pw_magic = magic20(p.tw + 2);
    p.tw_magic = magic20(p.tw);
    const int64_t stat_bytes = (int64_t)p.n_ptiles * 2 * 16 * 8;
    if (fold_a && (!stat_mean || !fold_b || !fold_beta || !fold_gamma || fold_n <= 0)) return IISEG_ERR_NULL;
    if (stat_mean) {
        if (!stat_inv_std || out_kind != 1 || !workspace || workspace_bytes < stat_bytes || ((uintptr_t)workspace & 15))
            return IISEG_ERR_UNSUPPORTED;
        p.stat_ws = (double*)workspace;
    }
    if (p.nsplit > 1) {
        const int64_t need = stat_bytes + (int64_t)p.n_ptiles * p.nsplit * (NB * 256 * 4) * 4;
        if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) p.nsplit = 1;
        else p.slabs = (float*)((char*)workspace + stat_bytes);
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 block(256);
    auto launch = [&](dim3 grid) {
        if (out_kind == 3) {
            if (unpool) IISEG_LAUNCH((conv_c8_m16_kernel<M16_UNPOOL, true>), grid, block, 0, s, p);
            else if (bn_a) IISEG_LAUNCH((conv_c8_m16_kernel<M16_BNRELU, true>), grid, block, 0, s, p);
            else IISEG_LAUNCH((conv_c8_m16_kernel<M16_PLAIN, true>), grid, block, 0, s, p);
        } else {
            if (unpool) IISEG_LAUNCH((conv_c8_m16_kernel<M16_UNPOOL, false>), grid, block, 0, s, p);
            else if (bn_a) IISEG_LAUNCH((conv_c8_m16_kernel<M16_BNRELU, false>), grid, block, 0, s, p);
            else IISEG_LAUNCH((conv_c8_m16_kernel<M16_PLAIN, false>), grid, block, 0, s, p);
        }
    };
    double* const stat_ws = p.stat_ws;
    if (p.nsplit > 1) {
        p.phase = 1;
        p.stat_ws = nullptr;
        launch(dim3(p.n_ptiles * p.nsplit));
        p.phase = 2;
        p.stat_ws = stat_ws;
        launch(dim3(p.n_ptiles));
    } else {
        p.phase = 0;
        launch(dim3(p.n_ptiles));
    }
    if (stat_ws)
        IISEG_LAUNCH(bn_stats_c8_final_kernel, dim3(fold_a ? 1 : 2), dim3(1024), 0, s, (const double*)stat_ws,
                     p.n_ptiles, p.out_c0 / 8, 2, (double)d->B * d->OH * d->OW, stat_eps, stat_mean, stat_inv_std,
                     fold_beta, fold_gamma, fold_a, fold_b, fold_n);
    return iiseg_check_launch();
}

extern "C" int iiseg_bn_fold_f32(void* stream, const float* beta, const float* gamma, const float* mean,
                                 const float* inv_std, float* a, float* b, int n) {
    if (!beta || !gamma || !mean || !inv_std || !a || !b) return IISEG_ERR_NULL;
    if (n <= 0) return IISEG_ERR_SHAPE;
    IISEG_LAUNCH(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, beta, gamma,
                       mean, inv_std, a, b, n);
    return iiseg_check_launch();
}

extern "C" int64_t iiseg_bn_stats_c8_workspace_elems(int n) { return (int64_t)((n + 7) / 8) * BN_SLICES * 16; }

extern "C" int iiseg_bn_stats_c8(void* stream, const void* x, int B, int Ctot, int c0, int n, int H, int W,
                                 double eps, float* mean, float* inv_std, double* workspace) {
    if (!x || !mean || !inv_std || !workspace) return IISEG_ERR_NULL;
    if (B <= 0 || Ctot <= 0 || n <= 0 || H <= 0 || W <= 0 || c0 < 0 || c0 + n > Ctot) return IISEG_ERR_SHAPE;
    if (Ctot % 8 || c0 % 8 || n % 8) return IISEG_ERR_UNSUPPORTED;
    IISEG_LAUNCH(bn_stats_c8_kernel, dim3(n / 8, BN_SLICES), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)x, B, Ctot / 8, c0 / 8, H * W, workspace);
    IISEG_LAUNCH(bn_stats_c8_final_kernel, dim3(n / 8), dim3(1024), 0, (hipStream_t)stream, workspace, BN_SLICES,
                 c0 / 8, n / 8, (double)B * H * W, eps, mean, inv_std, (const float*)nullptr, (const float*)nullptr,
                 (float*)nullptr, (float*)nullptr, 0);
    return iiseg_check_launch();
}
