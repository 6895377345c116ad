// Direct 3x3 convolution on bf16 "C8" activations (gfx950, v_mfma_f32_32x32x16_bf16, fp32 accumulate):
// the 16-bit MFMA path with the activations BETWEEN layers already in the matrix pipe's operand
// format.  A C8 tensor is (B, C/8, H, W, 8) bf16: the 8 channels one lane feeds to an MFMA are 16
// contiguous bytes ("k8 chunk"), so
//   * an input patch is staged global -> LDS by 16-byte LDS-DMA, one chunk per lane -- no VGPR pass,
//     no conversion, half the bytes of the fp32 NCHW form (conv_halo_bf16.hip spent most of a
//     k-tile issuing 24 dword loads + converts per thread: DESIGN 3.4 item 5);
//   * the MFMA B operand of tap (ky, kx) is one ds_read_b128 at patch[(y + ky) * PW + x + kx];
//   * the epilogue stores straight from the accumulators: in the 32x32 C/D layout a lane holds 4
//     consecutive channels of one pixel = 8 bytes of that pixel's chunk, lanes are consecutive
//     pixels, so one store instruction writes 512 contiguous bytes -- no LDS staging trip.
// Fusions (same call sites as conv_halo_bf16.hip: Conv2DLayer 3x3 of models/fcn8.py:34-71,
// models/fcn_down.py:102-104, models/fcn_up.py:83-86): bias, skip-add with crop (ElemwiseSumLayer),
// ReLU, output window / placement, 2x2 max-pool + DePool2D equality-mask BYTES taken from the fp32
// accumulators (layers/mylayers.py:111-114: same decisions as the fp32-activation form), DePool2D as
// the input staging of the decoder convs (up chunk + 8 mask bytes per patch element), the loop-
// invariant h half as an fp32 C8 addend, NCHW fp32 output for the class-score layer.
// Two pixel tilings: RECT (8 x 32 pixels of one image; fused pool) and FLAT (256 consecutive pixels
// of the window list of the whole batch: the 10^2..24^2 windows of the deep layers fill a tile that
// a 32-column tiling would leave 40-70 % empty).
// Pipeline: patch and weights double-buffered in ONE LDS array, both by LDS-DMA, one barrier per
// 16-channel k-tile (wait own DMA -> barrier -> issue next k-tile's DMA -> 36 MFMAs per wave).
// Numerics: statistical parity (8 significant bits per operand), like mma='bf16'; activations are
// rounded once, where an operand would be rounded anyway.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <tuple>
#include "iiseg.h"
#include "common.h"
#include "conv_common.h"
#include "c8_common.h"

using namespace iiseg;

namespace {

// Patch capacity of every variant: 5 DMA rounds of 256 chunks = 640 chunks per 8-channel half (two
// halves per 16-channel k-tile).  LDS per workgroup with 64 output channels: 2 x (18 KB weights +
// 20 KB patch) = 76 KB, two workgroups per CU.
#ifndef C8_STAGE_TAP
#define C8_STAGE_TAP 0          // -1: stage the next k-tile at the top of the step (the round-4 order)
#endif
constexpr int C8_NE = 5;
constexpr int C8_PCAP = C8_NE * 128;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int RSRC_W3 = 0x00027000;
constexpr unsigned OOB = 0x80000000u;

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
// lane ^ 1 of a 32-bit value (DPP quad_perm [1, 0, 3, 2])
__device__ __forceinline__ float dpp_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ unsigned dpp_xor1(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}

struct C8Params {
    const void* x1;               // C8 input (B, C1/8, H, W, 8); UNPOOL: `up` (B, C1/8, h2, w2, 8)
    const void* x2;               // second source of a channel concat (RECT only) or NULL
    const unsigned char* mask_in; // UNPOOL: (B, C1/8, h2, w2, 8) mask bytes
    const void* wp;               // Wp16[kt][tap][h][co][8] (iiseg_conv_halo_bf16_pack)
    const float* bias;
    const void* add;              // add_kind 1: C8 bf16 (B, Cout/8, AH, AW, 8); 2: C8 fp32
    void* out;                    // out_kind 0: none, 1: C8 bf16, 2: C8 fp32; OUTF32: float NCHW
    void* pool;                   // C8 bf16 (B, Cout/8, pool_H, pool_W, 8) or NULL
    unsigned char* mask_out;      // (B, Cout/8, pool_H, pool_W, 8) bytes or NULL
    int add_kind, out_kind;
    int B, C1, C2, H, W, h2, w2;
    int Cout, OH, OW, oy0, ox0, pad;
    int AH, AW, ay0, ax0;
    int nkt, Mpad;
    int out_ctot, out_c0, out_H, out_W, out_y0, out_x0;
    int pool_H, pool_W;
    int relu;
    int n_ptiles, n_mtiles, tiles_y, tiles_x;
    int N;                        // FLAT: B * OH * OW (quad: B * QH * QW quads)
    int PR, PWs;                  // FLAT: patch rows (capacity of a launch), patch row stride OW + 2
    int th, tw;                   // RECT: tile rows x columns, th * tw <= 128 TN pixels
    int quad;                     // pixel order inside a tile: 0 row-major, 1 by 2x2 pooling windows
    int QH, QW;                   // FLAT quad: pooling windows per image (ceil(OH / 2), ceil(OW / 2))
    unsigned pw_magic;            // ceil(2^20 / PWs): row of a patch chunk without a division
    unsigned tw_magic;            // RECT: ceil(2^20 / (quad ? tw / 2 : tw))
    int x3;                       // split-operand mode (template X3)
    int debug;
    int in_ct8;                   // chunk planes per image of the x1 TENSOR (>= C1 / 8: x1 may be a channel
                                  // slice of a wider C8 tensor -- a dense block's stack); X3: 2 C1 / 8
    int zins;                     // IISEG_CONV_ZINS: the logical input is the 2x zero-inserted x1,
                                  // z[2 i + 2][2 j + 2] = x1[i][j] on (2 h2 + 3) x (2 w2 + 3) (h2, w2 = x1's size)
};

// BM: output channels per workgroup (64; 32 for the class-score layer).  4 waves, each BM channels x
// TN 32-pixel columns: TN = 2 -> 8 x 32 pixel tiles (or 256 flat pixels), TN = 4 -> 16 x 32 pixel
// tiles: twice the MFMAs per k-tile between two barriers, per weight DMA and per workgroup prologue /
// epilogue -- the form of the large-window layers (few k-tiles, the fixed costs of a tile dominate).
//
// X3 ("bf16x3", the fp32-class mode of the 16-bit pipe): every activation is a PAIR of C8 tensors in
// one allocation, (B, 2 C/8, H, W, 8): chunks [0, C/8) of an image hold hi = bf16(v), chunks
// [C/8, 2 C/8) hold lo = bf16(v - hi) -- 16 significant bits.  The weights are split the same way on
// the host and packed as two k-groups [W_hi | W_lo]; per 16-channel tile the kernel runs three steps,
// x_lo W_hi, x_hi W_lo, x_hi W_hi, into ONE fp32 accumulation (the dropped x_lo W_lo term is 2^-18
// relative).  Same MFMA loop, same tiles, same epilogue; outputs / skip addends / pooled maps of
// kind 1 are hi / lo pairs, the DePool2D masks and the pool still come from the fp32 accumulators.
//
// NBUF: stages of the LDS ring.  2 = the pipelined form; 1 = layers with ONE k-tile (16 input channels:
// the first layers of both nets), where a second stage would never be filled -- half the LDS and, with
// 256-pixel tiles, few enough registers for three workgroups per CU instead of two (these layers are
// bound by latency and instruction issue around their 36 MFMAs, not by the matrix pipe).
//
// NW: waves per workgroup.  4 = the forms above; 8 (with TN = 2: again 512 pixels per workgroup, the
// weights of a k-tile shared by eight waves) halves the work and the registers of a wave against the
// TN = 4 form at the same LDS, so that FOUR waves share a SIMD instead of two: the shallow layers are
// bound by what happens around their few k-tiles (prologue, DMA round trip, pool / mask epilogue) and
// more resident waves fill those gaps.
//
// EPI: compile-time epilogue.  0 = the generic one (every combination of output format, addend format,
// pool, X3 decided at run time per chunk).  The frequent combinations have straight-line code of their own
// -- on the 1- to 8-k-tile layers, which are bound by instruction issue, the generic epilogue's scalar
// branches and addend plumbing are a fifth of a wave's instructions (conv1_1: 0.175 -> 0.124 ms):
//   1  ONLY the 2x2 max-pool (+ DePool2D mask bytes) of the ReLU of the result is stored (encoder layers)
//   2  bf16 C8 store, no addend, no pool (the plain layers of the FCN-8)
//   3  bf16 C8 store with a bf16 C8 skip addend (the decoder layers)
// Same values, same comparisons, same stores as the generic epilogue.
//   4  as 1, with an fp32 C8 addend summed in before the ReLU (the y half of the conv behind the h concat:
//      its loop-invariant h half is a cached fp32 map)
enum { EPI_GENERIC = 0, EPI_POOL = 1, EPI_STORE = 2, EPI_STORE_ADD = 3, EPI_POOL_ADD2 = 4 };
template <int BM, int TN, bool FLAT, bool UNPOOL, bool OUTF32, bool X3, int NBUF = 2, int NW = 4, int EPI = EPI_GENERIC>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : (NBUF == 1 ? (EPI != EPI_GENERIC ? 4 : 3) : 2)) void conv_c8_kernel(const C8Params p) {
    static_assert(EPI == EPI_GENERIC || !OUTF32, "specialised epilogues: bf16 C8 (or hi / lo pair) outputs");
    static_assert((EPI != EPI_POOL && EPI != EPI_POOL_ADD2) || !UNPOOL, "pool-only epilogue: encoder layers");
    static_assert(!FLAT || TN == 2, "flat tiles are 256 pixels");
    static_assert(NW == 4 || (NW == 8 && TN == 2 && !FLAT), "eight waves: 512-pixel rect tiles");
    constexpr int NT = 64 * NW;
    static_assert(NBUF == 2 || (!X3 && !UNPOOL), "single-stage form: plain one-k-tile layers");
    // patch buffer: C8_PCAP chunks per 8-channel half, staged as up to NE rounds of 256 chunks (a
    // wave issues only the pieces that hold patch chunks: wave-uniform test)
    constexpr int NCHK = C8_NE * 256;
    constexpr int NE = (NCHK + NT - 1) / NT;
    constexpr int TM = BM / 32;
    constexpr int WCH = 18 * BM;                             // weight chunks per k-tile
    constexpr int WPT = (WCH + NT - 1) / NT;
    static_assert(WCH % 64 == 0, "a wave's DMA piece is whole");

    // ONE LDS array: weight ring Ws[2][WCH], patch ring Ps[2][NCHK]
    __shared__ __attribute__((aligned(16))) uint4 smem[NBUF * WCH + NBUF * NCHK];
    uint4 (*Ws)[WCH] = reinterpret_cast<uint4 (*)[WCH]>(smem);
    uint4 (*Ps)[NCHK] = reinterpret_cast<uint4 (*)[NCHK]>(smem + NBUF * WCH);

    // pixel order of a tile: known at compile time under a specialised epilogue (pooling-window order
    // exactly when a pool is fused)
    const bool quad = (EPI == EPI_POOL || EPI == EPI_POOL_ADD2) ? true : (EPI == EPI_GENERIC ? p.quad != 0 : false);
    int pt, mt;
    tile_of_block(blockIdx.x, gridDim.x, p.n_ptiles, p.n_mtiles, pt, mt);
    const int m0 = mt * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int HW = p.H * p.W, hw2 = p.h2 * p.w2;
    const int OHW = p.OH * p.OW;
    const int CCh = p.C1 >> 3;                               // chunks of the C1 channels
    const int CC1 = X3 ? 2 * CCh : CCh;                      // chunks per image of source 1
    const int kt1 = p.C1 >> 4;                               // X3: k-tiles per k-group
    const int IC1 = p.in_ct8;                                // chunk planes per image of the x1 tensor (>= CC1)

    // ---- tile geometry -----------------------------------------------------------------------
    int tb = 0, wy0 = 0, wx0 = 0;     // RECT: image, tile origin in window coordinates
    int n0 = 0, vmin = 0;             // FLAT: first pixel (quad) of the tile, its virtual row
    int PWs, half;
    if constexpr (FLAT) {
        // The windows of all images stacked: virtual output row u = b * (OH + 2) + y (two unused
        // rows per image keep the 3-row halo of neighbouring images apart); the patch holds the
        // virtual INPUT rows [vmin, vmin + PR) x (OW + 2) columns.  A tile is 256 consecutive pixels
        // of that list, or (quad) 64 consecutive 2x2 pooling windows of it.
        if (quad) {
            const int QHW = p.QH * p.QW;
            n0 = pt * 64;
            const int b0 = n0 / QHW, r0 = n0 - b0 * QHW;
            vmin = b0 * (p.OH + 2) + 2 * (r0 / p.QW);
        } else {
            n0 = pt * 256;
            const int b0 = n0 / OHW, r0 = n0 - b0 * OHW;
            vmin = b0 * (p.OH + 2) + r0 / p.OW;
        }
        PWs = p.PWs;
        half = p.PR * PWs;
    } else {
        // RECT: th x tw pixels of one image (th * tw <= 128 TN; any shape whose patch fits)
        const int tpi = p.tiles_y * p.tiles_x;
        tb = pt / tpi;
        const int tr = pt - tb * tpi;
        const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
        wy0 = ty * p.th; wx0 = tx * p.tw;
        PWs = p.tw + 2;
        half = (p.th + 2) * PWs;
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // piece i of this wave holds chunks [i * 256 + wave * 64, + 64): staged only if the patch reaches it
    bool piece[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) piece[i] = i * NT + wave_u * 64 < 2 * half;

    constexpr bool UPQ = UNPOOL && !FLAT && !X3;      // DePool2D staged by pooled positions (below)
    // ---- patch staging offsets: chunk e = i * 256 + tid -> (half h, patch row, patch column) ----
    unsigned voff[NE];                 // byte offset of the chunk in its source (k-tile 0), or OOB
    unsigned voffm[UNPOOL ? NE : 1];   // UNPOOL: byte offset of its 8 mask bytes
    int bsel[UNPOOL ? NE : 1];         // UNPOOL: bit (y & 1) * 2 + (x & 1) of the mask byte
#pragma unroll
    for (int i = 0; i < (UPQ ? 0 : NE); ++i) {
        const int e = i * NT + tid;
        const int h = e >= half ? 1 : 0;
        const int rr = e - h * half;
        const int prow = (int)(((unsigned)rr * p.pw_magic) >> 20), pcol = rr - prow * PWs;   // rr / PWs
        bool ok = e < 2 * half;
        int b, iy;
        if constexpr (FLAT) {
            const int V = vmin + prow;
            b = V / (p.OH + 2);
            iy = p.oy0 - p.pad + (V - b * (p.OH + 2));
            ok = ok && b < p.B;
        } else {
            b = 0;                                            // per-image descriptors
            iy = p.oy0 + wy0 - p.pad + prow;
        }
        const int ix = p.ox0 + (FLAT ? 0 : wx0) - p.pad + pcol;
        ok = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        if constexpr (UNPOOL) {
            // DePool2D (layers/mylayers.py:95-114): only the 2 h2 x 2 w2 region has pooling windows
            ok = ok && iy < 2 * p.h2 && ix < 2 * p.w2;
            const unsigned pq = (unsigned)((iy >> 1) * p.w2 + (ix >> 1));
            voff[i] = ok ? ((unsigned)((b * IC1 + h) * hw2) + pq) * 16u : OOB;
            voffm[i] = ok ? ((unsigned)((b * CCh + h) * hw2) + pq) * 8u : OOB;
            bsel[i] = ((iy & 1) << 1) | (ix & 1);
        } else if (p.zins) {
            // FC-DenseNet's TransitionUp (models/FCDenseNet.py:118-121, a 3x3 stride-2 transposed convolution)
            // as a 'valid' correlation of the zero-inserted block: three of four patch chunks are zeros that
            // are never fetched (the out-of-range offset), the fourth comes straight from the block
            ok = ok && !((iy | ix) & 1) && iy >= 2 && ix >= 2 && iy <= 2 * p.h2 && ix <= 2 * p.w2;
            voff[i] = ok ? (unsigned)((b * IC1 + h) * hw2 + ((iy >> 1) - 1) * p.w2 + (ix >> 1) - 1) * 16u : OOB;
        } else {
            voff[i] = ok ? (unsigned)((b * IC1 + h) * HW + iy * p.W + ix) * 16u : OOB;
        }
    }
    // UPQ (DePool2D input on RECT tiles): a thread stages POOLED positions -- one `up` chunk + its 8 mask bytes,
    // two loads -- and writes the up to four patch elements of the 2x2 window (one select each, the mask bit a
    // compile-time shift); the per-element form above fetches the same pooled chunk four times (3.4 x the vector
    // memory instructions and their address arithmetic per k-tile, 30 instead of 12 staging registers).
    constexpr int NQ = 2;              // pooled positions per thread and k-tile: 2 halves x at most 256
    unsigned qoff[UPQ ? NQ : 1], qoffm[UPQ ? NQ : 1];
    int qlds[UPQ ? NQ : 1][4];
    bool pieceq[UPQ ? NQ : 1];
    if constexpr (UPQ) {
        const int iy0 = p.oy0 + wy0 - p.pad, ix0 = p.ox0 + wx0 - p.pad;      // patch origin, input coordinates
        const int PH = p.th + 2;
        const int Y20 = iy0 >> 1, X20 = ix0 >> 1;                            // (arithmetic shifts: floor)
        const int QH = ((iy0 + PH - 1) >> 1) - Y20 + 1, QW = ((ix0 + PWs - 1) >> 1) - X20 + 1, QP = QH * QW;
        const unsigned qw_magic = ((1u << 20) + (unsigned)QW - 1u) / (unsigned)QW;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int e = i * NT + tid;
            const int h = e >= QP ? 1 : 0;
            const int rr = e - h * QP;
            const int qy = (int)(((unsigned)rr * qw_magic) >> 20), qx = rr - qy * QW;
            const int Y2 = Y20 + qy, X2 = X20 + qx;
            const bool in = e < 2 * QP;
            // outside the h2 x w2 pooled map (padding, the odd trailing row / column) there is no window: the
            // loads return zeros and the patch elements written from them are zeros (layers/mylayers.py:95-114)
            const bool valid = in && (unsigned)Y2 < (unsigned)p.h2 && (unsigned)X2 < (unsigned)p.w2;
            const unsigned pq = (unsigned)(h * hw2 + Y2 * p.w2 + X2);
            qoff[i] = valid ? pq * 16u : OOB;
            qoffm[i] = valid ? pq * 8u : OOB;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int py = 2 * Y2 + (sl >> 1) - iy0, px = 2 * X2 + (sl & 1) - ix0;
                qlds[i][sl] = (in && (unsigned)py < (unsigned)PH && (unsigned)px < (unsigned)PWs)
                                  ? h * half + py * PWs + px : -1;
            }
            pieceq[i] = i * NT + wave_u * 64 < 2 * QP;
        }
    }
    // sources: RECT = image tb of each tensor (32-bit offsets inside one image), FLAT = whole tensor
    const int plane = (UNPOOL || p.zins) ? hw2 : HW;
    const char* base1 = (const char*)p.x1 + (FLAT ? (size_t)0 : (size_t)tb * IC1 * plane * 16);
    // (a channel slice: the descriptor ends with the slice's last plane of the last image it covers)
    const unsigned n1 = (unsigned)(((FLAT ? p.B : 1) - 1) * IC1 * plane + CC1 * plane) * 16u;
    const char* base2 = p.C2 > 0 ? (const char*)p.x2 + (size_t)tb * (p.C2 >> 3) * plane * 16 : base1;
    const unsigned n2 = p.C2 > 0 ? (unsigned)((p.C2 >> 3) * plane) * 16u : n1;
    const __amdgpu_buffer_rsrc_t r_x1 = mk_rsrc(base1, (p.debug & 2) ? 0u : n1);
    const i32x4s s_x1 = mk_srsrc(base1, (p.debug & 2) ? 0u : n1);
    const i32x4s s_x2 = mk_srsrc(base2, (p.debug & 2) ? 0u : n2);
    const __amdgpu_buffer_rsrc_t r_m =
        mk_rsrc(UNPOOL ? p.mask_in + (FLAT ? (size_t)0 : (size_t)tb * CCh * hw2 * 8) : nullptr,
                UNPOOL ? (unsigned)((FLAT ? p.B : 1) * CCh * hw2) * 8u : 0u);
    const i32x4s s_w = mk_srsrc(p.wp, (p.debug & 1) ? 0u : (unsigned)(p.nkt * 18 * p.Mpad) * 16u);
    const unsigned lds_w = __builtin_amdgcn_readfirstlane(lds_addr(&Ws[0][0]) + (unsigned)wave * 1024u);
    const unsigned lds_p = __builtin_amdgcn_readfirstlane(lds_addr(&Ps[0][0]) + (unsigned)wave * 1024u);

    // ---- MFMA B-operand positions of this lane's 32-pixel columns ----------------------------------
    // Pixel of (wave, j, l31) inside the tile: row-major (pixel n = (wave TN + j) 32 + l31 of the th x tw
    // tile, or of the flat list), or by pooling windows (quad): window q = (wave TN / 2 + j / 2) 16 +
    // l31 / 2, pixel (2 qy + (j & 1), 2 qx + (l31 & 1)) -- a lane's accumulators j, j + 1 are then the
    // two rows and lane ^ 1 the other column of one 2x2 window, whatever the tile's shape.
    int bpos[TN];
    int eb[TN], ey[TN], ex[TN];        // epilogue: image, window row, window column of the lane's pixel
    bool eok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if constexpr (FLAT) {
            if (quad) {
                const int q = n0 + wave * 16 + (l31 >> 1);
                const bool qok = q < p.N;
                const int qq = qok ? q : p.N - 1;
                const int QHW = p.QH * p.QW;
                eb[j] = qq / QHW;
                const int r = qq - eb[j] * QHW;
                const int qy = r / p.QW, qx = r - qy * p.QW;
                ey[j] = 2 * qy + j; ex[j] = 2 * qx + (l31 & 1);
                eok[j] = qok && ey[j] < p.OH && ex[j] < p.OW;
            } else {
                const int n = n0 + wave * 64 + j * 32 + l31;
                eok[j] = n < p.N;
                const int nn = eok[j] ? n : p.N - 1;
                eb[j] = nn / OHW;
                const int r = nn - eb[j] * OHW;
                ey[j] = r / p.OW; ex[j] = r - ey[j] * p.OW;
            }
            bpos[j] = (eb[j] * (p.OH + 2) + ey[j] - vmin) * PWs + ex[j];
        } else {
            eb[j] = 0;
            int ly, lx;
            bool inb;
            if (quad) {
                const int q = (wave * (TN / 2) + (j >> 1)) * 16 + (l31 >> 1);
                const int hw = p.tw >> 1;
                const int qy = (int)(((unsigned)q * p.tw_magic) >> 20), qx = q - qy * hw;      // q / hw
                ly = 2 * qy + (j & 1); lx = 2 * qx + (l31 & 1);
                inb = ly < p.th;
            } else {
                const int n = (wave * TN + j) * 32 + l31;
                ly = (int)(((unsigned)n * p.tw_magic) >> 20); lx = n - ly * p.tw;              // n / tw
                inb = ly < p.th;
            }
            if (!inb) { ly = 0; lx = 0; }
            ey[j] = wy0 + ly; ex[j] = wx0 + lx;
            eok[j] = inb && ey[j] < p.OH && ex[j] < p.OW;
            bpos[j] = ly * PWs + lx;
        }
    }

    // The accumulators start from the bias (C/D layout of the 32x32 MFMA: register r of a lane is
    // channel (r & 3) + 8 (r >> 2) + 4 lh of its 32-channel block; channels past Cout read 0): the
    // epilogue has no bias pass, and these loads return under the first DMA wait.
    f32x16 acc[TM][TN];
    {
        const __amdgpu_buffer_rsrc_t r_bias0 = mk_rsrc(p.bias, p.bias ? (unsigned)p.Cout * 4u : 0u);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    r_bias0, (int)(4u * (unsigned)(m0 + i * 32 + g * 8 + 4 * lh)), 0, 0));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[i][j][g * 4 + q] = bv[q];
            }
    }

    // A staging step moves the 16 channels that start at chunk KC of the source (tile-uniform: C1 % 16
    // == 0; chunks >= CC1 are the second source of a concat) and the packed weight k-tile WT.
#define C8_SRC(KC)                                                                                 \
    const bool s1 = X3 || UNPOOL || (KC) < CC1;                                                    \
    const unsigned so = (unsigned)(((KC) - (s1 ? 0 : CC1)) * plane) * 16u;
    // patch by LDS-DMA: lane -> one chunk, a wave's 64 chunks land contiguously
#define C8_DMA_X(KC, BUF)                                                                          \
    {                                                                                              \
        C8_SRC(KC)                                                                                 \
        const unsigned so_u = __builtin_amdgcn_readfirstlane(so);                                  \
        static_for<0, NE>([&](auto I) __attribute__((always_inline)) {                             \
            constexpr int i = decltype(I)::value;                                                  \
            if (piece[i])                                                                          \
                dma16(s1 ? s_x1 : s_x2, lds_p + (unsigned)((BUF) * NCHK + i * NT) * 16u, voff[i], so_u); \
        });                                                                                        \
    }
    // weights of packed k-tile WT, channels [m0, m0 + BM): rows (tap, h) of BM chunks each
    unsigned woff[WPT];
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const int f = j * NT + tid;
        woff[j] = f < WCH ? 16u * (unsigned)((f / BM) * p.Mpad + m0 + f % BM) : OOB;
    }
#define C8_DMA_W(WT, BUF)                                                                          \
    {                                                                                              \
        const unsigned so_w = __builtin_amdgcn_readfirstlane((unsigned)((WT) * 18 * p.Mpad) * 16u); \
        static_for<0, WPT>([&](auto J) __attribute__((always_inline)) {                            \
            constexpr int j = decltype(J)::value;                                                  \
            if ((j + 1) * NT <= WCH || j * NT + wave * 64 < WCH)                                   \
                dma16(s_w, lds_w + (unsigned)((BUF) * WCH + j * NT) * 16u, woff[j], so_w);         \
        });                                                                                        \
    }
    // UNPOOL: up chunk (chunk KC of `up`) + mask bytes (chunk KM of the mask) into registers, selected
    // and written to LDS later
    u32x4 xu[UNPOOL ? (UPQ ? NQ : NE) : 1];
    u32x2 xm[UNPOOL ? (UPQ ? NQ : NE) : 1];
#define C8_LOAD_U(KC, KM)                                                                          \
    if constexpr (UPQ) {                                                                           \
        const unsigned so = (unsigned)((KC) * hw2) * 16u;                                          \
        const unsigned som = (unsigned)((KM) * hw2) * 8u;                                          \
        static_for<0, NQ>([&](auto I) __attribute__((always_inline)) {                             \
            constexpr int i = decltype(I)::value;                                                  \
            if (i == 0 || pieceq[i]) {                                                             \
                xu[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(           \
                    r_x1, (int)qoff[i], (int)so, 0));                                              \
                xm[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(            \
                    r_m, (int)qoffm[i], (int)som, 0));                                             \
            }                                                                                      \
        });                                                                                        \
    } else {                                                                                       \
        const unsigned so = (unsigned)((KC) * hw2) * 16u;                                          \
        const unsigned som = (unsigned)((KM) * hw2) * 8u;                                          \
        static_for<0, NE>([&](auto I) __attribute__((always_inline)) {                             \
            constexpr int i = decltype(I)::value;                                                  \
            if (i < 3 || piece[i]) {   /* (no branch around the pieces every launch has) */       \
                xu[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(           \
                    r_x1, (int)voff[i], (int)so, 0));                                              \
                xm[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(            \
                    r_m, (int)voffm[i], (int)som, 0));                                             \
            }                                                                                      \
        });                                                                                        \
    }
#define C8_STORE_U(BUF)                                                                            \
    if constexpr (UPQ) {                                                                           \
        static_for<0, NQ * 4>([&](auto I) __attribute__((always_inline)) {                         \
            constexpr int i = decltype(I)::value / 4, sl = decltype(I)::value % 4;                 \
            if (i > 0 && !pieceq[i]) return;                                                       \
            /* bit sl of byte j of the mask pair: pre == pooled for channel j at the window's pixel sl */ \
            const unsigned t0 = (xm[i][0] >> sl) & 0x01010101u;                                    \
            const unsigned t1 = (xm[i][1] >> sl) & 0x01010101u;                                    \
            const unsigned b0 = (t0 << 8) - t0, b1 = (t1 << 8) - t1;                               \
            uint4 v;                                                                               \
            v.x = xu[i][0] & __builtin_amdgcn_perm(b0, b0, 0x01010000u);                           \
            v.y = xu[i][1] & __builtin_amdgcn_perm(b0, b0, 0x03030202u);                           \
            v.z = xu[i][2] & __builtin_amdgcn_perm(b1, b1, 0x01010000u);                           \
            v.w = xu[i][3] & __builtin_amdgcn_perm(b1, b1, 0x03030202u);                           \
            if (qlds[i][sl] >= 0) Ps[BUF][qlds[i][sl]] = v;                                        \
        });                                                                                        \
    } else                                                                                         \
    static_for<0, NE>([&](auto I) __attribute__((always_inline)) {                                 \
        constexpr int i = decltype(I)::value;                                                      \
        if (i >= 3 && !piece[i]) return;                                                           \
        /* byte j of the mask pair, bit bsel: pre == pooled at this pixel for channel j.  The four  \
           bits of a dword become four 0x00 / 0xff bytes (t * 255 without a multiply), v_perm_b32   \
           doubles each byte into the 16-bit lane of its channel */                                 \
        const unsigned t0 = (xm[i][0] >> bsel[i]) & 0x01010101u;                                   \
        const unsigned t1 = (xm[i][1] >> bsel[i]) & 0x01010101u;                                   \
        const unsigned b0 = (t0 << 8) - t0, b1 = (t1 << 8) - t1;                                   \
        uint4 v;                                                                                   \
        v.x = xu[i][0] & __builtin_amdgcn_perm(b0, b0, 0x01010000u);                               \
        v.y = xu[i][1] & __builtin_amdgcn_perm(b0, b0, 0x03030202u);                               \
        v.z = xu[i][2] & __builtin_amdgcn_perm(b1, b1, 0x01010000u);                               \
        v.w = xu[i][3] & __builtin_amdgcn_perm(b1, b1, 0x03030202u);                               \
        if (i * NT + tid < 2 * half) Ps[BUF][i * NT + tid] = v;                                    \
    });

    // The k-loop.  Plain: step kt = k-tile kt, both rings alternate (buffer kt & 1), the step's top
    // issues k-tile kt + 1 into the other buffers.
    // X3: channel tile c = steps 3c .. 3c + 2, in the order
    //     r = 0: x_lo_c (Ps[0]) x W_hi_c (Ws[c & 1])      top: stage x_hi_c -> Ps[1], W_lo_c -> Ws[~c & 1]
    //     r = 1: x_hi_c (Ps[1]) x W_lo_c (Ws[~c & 1])     top: stage x_lo_{c+1} -> Ps[0]
    //     r = 2: x_hi_c (Ps[1]) x W_hi_c (Ws[c & 1])      top: stage W_hi_{c+1} -> Ws[~c & 1]
    // -- x_hi staged for the W_lo product stays in LDS for the W_hi product, W_hi staged for the x_lo
    // product stays for the x_hi product: two patches and two weight tiles per channel tile instead of
    // three and three, every buffer written only after the barrier that follows its last read.
    // Packed weights: k-tiles [0, kt1) = W_hi, [kt1, 2 kt1) = W_lo.
    const int nsteps = X3 ? 3 * kt1 : p.nkt;
    if constexpr (UNPOOL) {
        C8_LOAD_U(X3 ? CCh : 0, 0)
        C8_DMA_W(0, 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C8_STORE_U(0)
    } else {
        C8_DMA_X(X3 ? CCh : 0, 0)
        C8_DMA_W(0, 0)
    }

    int xc = 0, xr = 0;                // X3: channel tile, product of the step
    for (int kt = 0; kt < nsteps; ++kt) {
        const int pb = X3 ? (xr ? 1 : 0) : (kt & 1);                          // patch buffer read by this step
        const int wb = X3 ? (xr == 1 ? ((xc & 1) ^ 1) : (xc & 1)) : (kt & 1);   // weight buffer
        const bool more = X3 ? xc + 1 < kt1 : kt + 1 < nsteps;
        // the step's operands have landed: every wave retires its own DMA pieces (and, UNPOOL, its LDS
        // writes), then the barrier publishes them -- and tells that every wave is done READING what
        // the previous step read, which the DMA issued from here on may overwrite
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int stored = -1;               // UNPOOL: patch buffer the registers loaded here go to
        if constexpr (X3) {
            if (xr == 0) {
                if constexpr (UNPOOL) { C8_LOAD_U(2 * xc, 2 * xc) stored = 1; } else { C8_DMA_X(2 * xc, 1) }
                C8_DMA_W(kt1 + xc, (xc & 1) ^ 1)
            } else if (xr == 1) {
                if (more) {
                    if constexpr (UNPOOL) { C8_LOAD_U(CCh + 2 * (xc + 1), 2 * (xc + 1)) stored = 0; }
                    else { C8_DMA_X(CCh + 2 * (xc + 1), 0) }
                }
            } else if (more) {
                C8_DMA_W(xc + 1, (xc & 1) ^ 1)
            }
        }
        // Plain mode: the next k-tile is staged BEHIND the MFMAs of tap 0 (below), not here: after the barrier a
        // wave first fetches its tap-0 operands and starts the matrix pipe, and issues the ~20 DMA / load
        // instructions of the next k-tile (address set-up included) while those MFMAs run -- at the top of the
        // step they stood between the barrier and the first MFMA of every k-tile.
        auto stage_next = [&]() __attribute__((always_inline)) {
            if constexpr (!X3 && !UNPOOL) {
                if (more && !(p.debug & 16)) {     // (debug 16: timing experiment, no DMA issued at all)
                    C8_DMA_X(2 * (kt + 1), pb ^ 1)
                    C8_DMA_W(kt + 1, wb ^ 1)
                }
            } else if constexpr (!X3) {
                if (more) {
                    C8_LOAD_U(2 * (kt + 1), 2 * (kt + 1))
                    C8_DMA_W(kt + 1, wb ^ 1)
                    stored = pb ^ 1;
                }
            }
        };
        constexpr int STAGE_TAP = NBUF == 1 ? -1 : C8_STAGE_TAP;     // (one-k-tile layers stage nothing)
        if constexpr (STAGE_TAP < 0) stage_next();
        // operands of tap t+1 are read from LDS while the MFMAs of tap t run (two register sets)
        uint4 a[2][TM], bq[2][TN];
        auto lds_operands = [&](auto TAP, auto SET) __attribute__((always_inline)) {
            constexpr int tap = decltype(TAP)::value, set = decltype(SET)::value;
            constexpr int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[set][i] = Ws[wb][(tap * 2 + lh) * BM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bq[set][j] = Ps[pb][lh * half + bpos[j] + ky * PWs + kx];
        };
        lds_operands(ic<0>{}, ic<0>{});
        static_for<0, 9>([&](auto TAP) __attribute__((always_inline)) {
            constexpr int tap = decltype(TAP)::value;
            if constexpr (tap + 1 < 9) lds_operands(ic<tap + 1>{}, ic<(tap + 1) & 1>{});
            // (the reads of tap t+1 stay IN FRONT of the MFMAs of tap t: left to itself hipcc sinks them
            // behind three of the four MFMAs and then waits for them one MFMA later)
            __builtin_amdgcn_sched_barrier(0);
            {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, a[tap & 1][i]),
                            __builtin_bit_cast(bf16x8, bq[tap & 1][j]), acc[i][j], 0, 0, 0);
            }
            if constexpr (tap == STAGE_TAP) stage_next();
            __builtin_amdgcn_sched_barrier(0);             // keep that order tap by tap
        });
        if constexpr (UNPOOL) {
            // (the target buffer is not read by this step and was last read before the barrier above)
            if (stored >= 0) { C8_STORE_U(stored) }
        }
        if constexpr (X3) {
            if (++xr == 3) { xr = 0; ++xc; }
        }
    }
#undef C8_SRC
#undef C8_DMA_X
#undef C8_DMA_W
#undef C8_LOAD_U
#undef C8_STORE_U

    // ---- epilogue, straight from the accumulators ------------------------------------------------
    // C/D layout of the 32x32 MFMA: column = lane & 31 (pixel), row = (r & 3) + 8 * (r >> 2) + 4 * lh:
    // register group g = r >> 2 holds channels 8 g + 4 lh + (0..3) = one half of chunk g.
    if ((p.debug & 8) && acc[0][0][0] != 12345.f) return;
    const int OPL = p.out_H * p.out_W, APL = p.AH * p.AW, PPL = p.pool_H * p.pool_W;
    const int ib = FLAT ? 0 : tb;                 // image folded into the descriptor base (RECT)
    if constexpr (EPI == EPI_STORE || EPI == EPI_STORE_ADD) {
        // (X3: outputs and addends are hi / lo pairs -- hi chunks, then lo chunks, per image)
        constexpr int PR2 = X3 ? 2 : 1;
        const int oct8 = p.out_ctot >> 3, co8 = ((p.Cout + 15) >> 4) << 1;
        const int octT = PR2 * oct8, co8A = PR2 * co8;
        const __amdgpu_buffer_rsrc_t r_out =
            mk_rsrc((const char*)p.out + (size_t)ib * octT * OPL * 16, (unsigned)((FLAT ? p.B : 1) * octT * OPL) * 16u);
        const __amdgpu_buffer_rsrc_t r_add =
            mk_rsrc(EPI == EPI_STORE_ADD ? (const char*)p.add + (size_t)ib * co8A * APL * 16 : nullptr,
                    EPI == EPI_STORE_ADD ? (unsigned)((FLAT ? p.B : 1) * co8A * APL) * 16u : 0u);
        unsigned ob[TN], ab[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned opix = (unsigned)(eb[j] * octT * OPL + (p.out_y0 + ey[j]) * p.out_W + p.out_x0 + ex[j]);
            const unsigned apix = (unsigned)(eb[j] * co8A * APL + (p.ay0 + ey[j]) * p.AW + p.ax0 + ex[j]);
            ob[j] = eok[j] ? opix * 16u + 8u * lh : OOB;
            ab[j] = eok[j] ? apix * 16u + 8u * lh : OOB;
        }
        const float rfloor = p.relu ? 0.f : -__builtin_inff();
        // (every load first: a buffer load cannot move above an earlier buffer store.  Plain mode: the skip
        // pieces of BOTH 32-channel blocks before the first store -- one memory round trip per workgroup instead
        // of one per block; the pair mode loads per block, twice the pieces)
        constexpr int AB = X3 ? 1 : TM;              // blocks whose addend pieces are in flight together
        u32x2 adA[AB][4][TN], adl[X3 ? 4 : 1][TN];
        auto load_add = [&](int i, int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c8 = ((m0 + i * 32) >> 3) + g;
                const int so_a = (int)((unsigned)(c8 * APL) * 16u);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    adA[slot][g][j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
                        r_add, (int)(c8 < co8 ? ab[j] : OOB), so_a, 0));
                    if constexpr (X3)
                        adl[g][j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
                            r_add, (int)(c8 < co8 ? ab[j] : OOB), so_a + (int)((unsigned)(co8 * APL) * 16u), 0));
                }
            }
        };
        if constexpr (EPI == EPI_STORE_ADD && !X3) {
#pragma unroll
            for (int i = 0; i < TM; ++i) load_add(i, i);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (EPI == EPI_STORE_ADD && X3) load_add(i, 0);
            auto& ad = adA[X3 ? 0 : i];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c8 = ((m0 + i * 32) >> 3) + g;
                if (c8 >= co8) continue;               // wave-uniform: chunks past the padded channels
                const int so_o = (int)((unsigned)(((p.out_c0 >> 3) + c8) * OPL) * 16u);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = acc[i][j][g * 4 + q];
                    if constexpr (EPI == EPI_STORE_ADD) {
                        f32x4 a = f32x4{bf_lo(ad[g][j][0]), bf_hi(ad[g][j][0]), bf_lo(ad[g][j][1]), bf_hi(ad[g][j][1])};
                        if constexpr (X3)      // hi + lo: the 16-bit value the producer stored
                            a += f32x4{bf_lo(adl[g][j][0]), bf_hi(adl[g][j][0]), bf_lo(adl[g][j][1]), bf_hi(adl[g][j][1])};
                        v += a;
                    }
                    v[0] = fmaxf(v[0], rfloor); v[1] = fmaxf(v[1], rfloor);
                    v[2] = fmaxf(v[2], rfloor); v[3] = fmaxf(v[3], rfloor);
                    u32x2 w2;
                    w2[0] = pack_bf16(v[0], v[1]); w2[1] = pack_bf16(v[2], v[3]);
                    __builtin_amdgcn_raw_buffer_store_b64(w2, r_out, (int)ob[j], so_o, 0);
                    if constexpr (X3) {
                        u32x2 l2;
                        l2[0] = pack_bf16(v[0] - bf_lo(w2[0]), v[1] - bf_hi(w2[0]));
                        l2[1] = pack_bf16(v[2] - bf_lo(w2[1]), v[3] - bf_hi(w2[1]));
                        __builtin_amdgcn_raw_buffer_store_b64(l2, r_out, (int)ob[j],
                                                              so_o + (int)((unsigned)(oct8 * OPL) * 16u), 0);
                    }
                }
            }
        }
    } else if constexpr (EPI == EPI_POOL || EPI == EPI_POOL_ADD2) {
        constexpr bool ADD2 = EPI == EPI_POOL_ADD2;
        const int co8 = ((p.Cout + 15) >> 4) << 1;
        // ADD2: the fp32 C8 addend (B, Cout / 8, AH, AW, 8) floats: a lane's four channels are 16 bytes
        const __amdgpu_buffer_rsrc_t r_add2 =
            mk_rsrc(ADD2 ? (const char*)p.add + (size_t)ib * co8 * APL * 32 : nullptr,
                    ADD2 ? (unsigned)((FLAT ? p.B : 1) * co8 * APL) * 32u : 0u);
        unsigned ab2[ADD2 ? TN : 1];
        if constexpr (ADD2) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned apix = (unsigned)(eb[j] * co8 * APL + (p.ay0 + ey[j]) * p.AW + p.ax0 + ex[j]);
                ab2[j] = eok[j] ? apix * 32u + 16u * lh : OOB;
            }
        }
        constexpr int GHP = TN == 4 ? 2 : 4;       // chunks whose addend loads are in flight together
        const int co8P = X3 ? 2 * co8 : co8;       // X3: the pooled map is a hi / lo pair, the masks are not
        const __amdgpu_buffer_rsrc_t r_pool =
            mk_rsrc((const char*)p.pool + (size_t)ib * co8P * PPL * 16, (unsigned)((FLAT ? p.B : 1) * co8P * PPL) * 16u);
        const __amdgpu_buffer_rsrc_t r_mask =
            mk_rsrc(p.mask_out ? p.mask_out + (size_t)ib * co8 * PPL * 8 : nullptr,
                    p.mask_out ? (unsigned)((FLAT ? p.B : 1) * co8 * PPL) * 8u : 0u);
        unsigned qb[TN / 2], qm[TN / 2];
#pragma unroll
        for (int jp = 0; jp < TN / 2; ++jp) {
            const int q_wy = ey[2 * jp], q_wx = ex[2 * jp];
            const int q_py = (p.oy0 + q_wy) >> 1, q_px = (p.ox0 + q_wx) >> 1;
            const bool q_ok = eok[2 * jp] && !(l31 & 1) && q_wy + 1 < p.OH && q_wx + 1 < p.OW &&
                              q_py < p.pool_H && q_px < p.pool_W;
            const unsigned qpix = (unsigned)(q_py * p.pool_W + q_px);
            qb[jp] = q_ok ? ((unsigned)(eb[2 * jp] * co8P * PPL) + qpix) * 16u + 8u * lh : OOB;
            qm[jp] = q_ok ? ((unsigned)(eb[2 * jp] * co8 * PPL) + qpix) * 8u + 4u * lh : OOB;
        }
        // The mask byte of a channel without compares: every value a of a window is <= its maximum m, so the
        // sign of a - m is set exactly when a is NOT the maximum (a != m -> a - m != 0: denormals are kept;
        // a == m -> +0: after the ReLU -- the host picks this epilogue for ReLU layers only -- no value is
        // -0).  v_alignbit funnels the signs into one word, the upper row's with 6 bits of the difference,
        // the lower row's with 2: signs at bits 8 q + 5 / 8 q + 7, junk (exponent bits) between them; one
        // shift, one v_bfi (~x & pattern) later the word is what the compare / select / or chain built --
        // 2 instead of 3.5 vector instructions per value and no VCC round trips (RECT forms: the shallow,
        // issue-bound layers; conv1_1 0.111 -> 0.109 ms).
        const unsigned xsh = (unsigned)(l31 & 1);
        const unsigned nsh = 5u - xsh, pat = 0x05050505u << xsh;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c8 = ((m0 + i * 32) >> 3) + g;
                if constexpr (ADD2) {
                    // (every load of a group of chunks before its first store: a buffer load cannot move above
                    // an earlier buffer store; the sum acc + addend is the generic epilogue's)
                    if (g % GHP == 0) {
#pragma unroll
                        for (int gg = g; gg < g + GHP; ++gg) {
                            const int cg = ((m0 + i * 32) >> 3) + gg;
                            const int so_a = (int)((unsigned)(cg * APL) * 32u);
#pragma unroll
                            for (int j = 0; j < TN; ++j) {
                                const f32x4 av = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    r_add2, (int)(cg < co8 ? ab2[j] : OOB), so_a, 0));
#pragma unroll
                                for (int q = 0; q < 4; ++q) acc[i][j][gg * 4 + q] += av[q];
                            }
                        }
                    }
                }
                if (c8 >= co8) continue;               // wave-uniform: chunks past the padded channels
                const int so_p = (int)((unsigned)(c8 * PPL) * 16u);
#pragma unroll
                for (int jp = 0; jp < TN / 2; ++jp) {
                    f32x4 m, a0, a1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        a0[q] = fmaxf(acc[i][2 * jp][g * 4 + q], 0.f);
                        a1[q] = fmaxf(acc[i][2 * jp + 1][g * 4 + q], 0.f);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float mv = fmaxf(a0[q], a1[q]);
                        m[q] = fmaxf(mv, dpp_xor1(mv));
                    }
                    unsigned own;
                    if constexpr (!FLAT) {
                        // (two chains of four, joined by one shift-or: half the dependent depth)
                        unsigned nbh = 0, nbl = 0;
#pragma unroll
                        for (int q = 1; q >= 0; --q) {
                            nbh = __builtin_amdgcn_alignbit(nbh, __builtin_bit_cast(unsigned, a1[q + 2] - m[q + 2]), 30);
                            nbl = __builtin_amdgcn_alignbit(nbl, __builtin_bit_cast(unsigned, a1[q] - m[q]), 30);
                            nbh = __builtin_amdgcn_alignbit(nbh, __builtin_bit_cast(unsigned, a0[q + 2] - m[q + 2]), 26);
                            nbl = __builtin_amdgcn_alignbit(nbl, __builtin_bit_cast(unsigned, a0[q] - m[q]), 26);
                        }
                        own = ~(((nbh << 16) | nbl) >> nsh) & pat;
                    } else {
                        // (the flat form -- deep layers, two waves per SIMD next to a partner workgroup's MFMAs --
                        // measured 1-2 % FASTER with the compare / select chain than with the denser code above:
                        // fcn conv4_3 0.378 against 0.381 ms, dae conv6_1 0.279 against 0.284)
                        own = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            own |= ((a0[q] == m[q] ? 1u : 0u) | (a1[q] == m[q] ? 4u : 0u)) << (8 * q);
                        own <<= xsh;
                    }
                    const unsigned mb = own | dpp_xor1(own);
                    u32x2 w2;
                    w2[0] = pack_bf16(m[0], m[1]); w2[1] = pack_bf16(m[2], m[3]);
                    __builtin_amdgcn_raw_buffer_store_b64(w2, r_pool, (int)qb[jp], so_p, 0);
                    if constexpr (X3) {
                        u32x2 l2;
                        l2[0] = pack_bf16(m[0] - bf_lo(w2[0]), m[1] - bf_hi(w2[0]));
                        l2[1] = pack_bf16(m[2] - bf_lo(w2[1]), m[3] - bf_hi(w2[1]));
                        __builtin_amdgcn_raw_buffer_store_b64(l2, r_pool, (int)qb[jp],
                                                              so_p + (int)((unsigned)(co8 * PPL) * 16u), 0);
                    }
                    __builtin_amdgcn_raw_buffer_store_b32((int)mb, r_mask, (int)qm[jp], so_p >> 1, 0);
                }
            }
    } else if constexpr (OUTF32) {
        // class-score layer: NCHW fp32, Cout <= 32
        static_assert(TM == 1, "NCHW output: one 32-channel block");
        const __amdgpu_buffer_rsrc_t r_out =
            mk_rsrc((const char*)p.out + (size_t)ib * p.out_ctot * OPL * 4,
                    (unsigned)((FLAT ? p.B : 1) * p.out_ctot * OPL) * 4u);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned o0 = (unsigned)((eb[j] * p.out_ctot + p.out_c0) * OPL +
                                           (p.out_y0 + ey[j]) * p.out_W + p.out_x0 + ex[j]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[0][j][r];
                if (p.relu) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(
                    __builtin_bit_cast(int, v), r_out,
                    (int)((eok[j] && co < p.Cout) ? 4u * (o0 + (unsigned)(co * OPL)) : OOB), 0, 0);
            }
        }
    } else {
        // chunks of the Cout output channels: whole 16-channel groups (the layout every C8 consumer
        // expects); channels past Cout come out as exact zeros (zero weight rows, no bias)
        const int oct8 = p.out_ctot >> 3, co8 = ((p.Cout + 15) >> 4) << 1;
        // X3: a kind-1 tensor holds hi chunks, then lo chunks, per image
        const int octT = (X3 && p.out_kind == 1) ? 2 * oct8 : oct8;
        const int co8A = (X3 && p.add_kind == 1) ? 2 * co8 : co8;
        const int co8P = X3 ? 2 * co8 : co8;
        const unsigned osz = p.out_kind == 2 ? 32u : 16u;     // bytes per output chunk
        const __amdgpu_buffer_rsrc_t r_out =
            mk_rsrc(p.out ? (const char*)p.out + (size_t)ib * octT * OPL * osz : nullptr,
                    p.out ? (unsigned)((FLAT ? p.B : 1) * octT * OPL) * osz : 0u);
        const unsigned asz = p.add_kind == 2 ? 32u : 16u;
        const __amdgpu_buffer_rsrc_t r_add =
            mk_rsrc(p.add ? (const char*)p.add + (size_t)ib * co8A * APL * asz : nullptr,
                    p.add ? (unsigned)((FLAT ? p.B : 1) * co8A * APL) * asz : 0u);
        const bool pooling = p.pool != nullptr;       // (host: only with the quad pixel order)
        const __amdgpu_buffer_rsrc_t r_pool =
            mk_rsrc(pooling ? (const char*)p.pool + (size_t)ib * co8P * PPL * 16 : nullptr,
                    pooling ? (unsigned)((FLAT ? p.B : 1) * co8P * PPL) * 16u : 0u);
        const __amdgpu_buffer_rsrc_t r_mask =
            mk_rsrc(pooling && p.mask_out ? p.mask_out + (size_t)ib * co8 * PPL * 8 : nullptr,
                    pooling && p.mask_out ? (unsigned)((FLAT ? p.B : 1) * co8 * PPL) * 8u : 0u);
        // byte offsets of the lane's pixels (its half of a chunk) inside chunk plane 0 of their image,
        // or the out-of-bounds offset; the chunk plane goes into the instructions' scalar offset
        unsigned ob[TN], ab[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned opix = (unsigned)(eb[j] * octT * OPL + (p.out_y0 + ey[j]) * p.out_W + p.out_x0 + ex[j]);
            const unsigned apix = (unsigned)(eb[j] * co8A * APL + (p.ay0 + ey[j]) * p.AW + p.ax0 + ex[j]);
            ob[j] = eok[j] ? opix * osz + (osz >> 1) * lh : OOB;
            ab[j] = eok[j] ? apix * asz + (asz >> 1) * lh : OOB;
        }
        // fused pool (quad pixel order): accumulators (2 jp, 2 jp + 1) are the two rows of a pooling
        // window, lane ^ 1 its other column; the even lane stores
        unsigned qb[TN / 2], qm[TN / 2];  // byte offsets of the pooled pixel's half chunk / mask dword, or OOB
#pragma unroll
        for (int jp = 0; jp < TN / 2; ++jp) {
            const int q_wy = ey[2 * jp], q_wx = ex[2 * jp];
            const int q_py = (p.oy0 + q_wy) >> 1, q_px = (p.ox0 + q_wx) >> 1;
            const bool q_ok = pooling && eok[2 * jp] && !(l31 & 1) && q_wy + 1 < p.OH && q_wx + 1 < p.OW &&
                              q_py < p.pool_H && q_px < p.pool_W;
            const unsigned qpix = (unsigned)(q_py * p.pool_W + q_px);
            qb[jp] = q_ok ? ((unsigned)(eb[2 * jp] * co8P * PPL) + qpix) * 16u + 8u * lh : OOB;
            qm[jp] = q_ok ? ((unsigned)(eb[2 * jp] * co8 * PPL) + qpix) * 8u + 4u * lh : OOB;
        }
        // Every load of the epilogue first (bias, skip-add; out-of-range pieces get the out-of-bounds
        // offset instead of a branch), then the stores: a buffer load that follows a buffer store in
        // program order cannot be moved above it (they may alias as far as hipcc knows), so a
        // {load, add, store} body per chunk is one memory round trip per chunk -- eight in a row
        // (hoisted per 32-channel block i: one register set of 4 x TN pieces serves both add formats)
        const bool has_add1 = p.add_kind == 1, has_add2 = p.add_kind == 2;
        const float rfloor = p.relu ? 0.f : -__builtin_inff();
        constexpr int GH = TN == 4 ? 2 : 4;      // chunks whose loads are hoisted together (registers)
#pragma unroll
        for (int ig = 0; ig < TM * (4 / GH); ++ig) {
            const int i = ig / (4 / GH), g0 = (ig % (4 / GH)) * GH;
            u32x4 adr[4][TN];                 // raw addend pieces (bf16 hi [, lo] halves or 4 floats)
#pragma unroll
            for (int g = g0; g < g0 + GH; ++g) {
                const int c8 = ((m0 + i * 32) >> 3) + g;
                const bool cok = c8 < co8;             // wave-uniform: chunks past the padded channels
                const int so_a = (int)((unsigned)(c8 * APL) * asz);
                if (cok && has_add1) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const u32x2 a2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
                            r_add, (int)ab[j], so_a, 0));
                        adr[g][j][0] = a2[0]; adr[g][j][1] = a2[1];
                        if constexpr (X3) {
                            const u32x2 a3 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
                                r_add, (int)ab[j], so_a + (int)((unsigned)(co8 * APL) * 16u), 0));
                            adr[g][j][2] = a3[0]; adr[g][j][3] = a3[1];
                        }
                    }
                } else if (cok && has_add2) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        adr[g][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            r_add, (int)ab[j], so_a, 0));
                }
            }
            // the addend as four floats per piece, whatever its format (zeros without one): the value
            // loop below has no branch
            f32x4 addv[4][TN];
#pragma unroll
            for (int g = g0; g < g0 + GH; ++g) {
                const bool cok = ((m0 + i * 32) >> 3) + g < co8;
                if (cok && has_add1) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        addv[g][j] = f32x4{bf_lo(adr[g][j][0]), bf_hi(adr[g][j][0]),
                                           bf_lo(adr[g][j][1]), bf_hi(adr[g][j][1])};
                        if constexpr (X3)      // hi + lo: the 16-bit value the producer stored
                            addv[g][j] += f32x4{bf_lo(adr[g][j][2]), bf_hi(adr[g][j][2]),
                                                bf_lo(adr[g][j][3]), bf_hi(adr[g][j][3])};
                    }
                } else if (cok && has_add2) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) addv[g][j] = __builtin_bit_cast(f32x4, adr[g][j]);
                } else {
#pragma unroll
                    for (int j = 0; j < TN; ++j) addv[g][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int g = g0; g < g0 + GH; ++g) {
                const int c8 = ((m0 + i * 32) >> 3) + g;            // chunk of the output channels
                const bool cok = c8 < co8;
                f32x4 v[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[j][q] = acc[i][j][g * 4 + q];
                    if (has_add1 || has_add2) v[j] += addv[g][j];     // (wave-uniform)
                    // (ReLU without a branch per piece: max with 0 or with -inf)
                    v[j][0] = fmaxf(v[j][0], rfloor); v[j][1] = fmaxf(v[j][1], rfloor);
                    v[j][2] = fmaxf(v[j][2], rfloor); v[j][3] = fmaxf(v[j][3], rfloor);
                }
                // stores: the pixel's byte offset is per lane and fixed, the chunk plane a scalar offset
                const int so_o = (int)((unsigned)(((p.out_c0 >> 3) + c8) * OPL) * osz);
                if (cok && p.out_kind == 1) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        u32x2 w2;
                        w2[0] = pack_bf16(v[j][0], v[j][1]); w2[1] = pack_bf16(v[j][2], v[j][3]);
                        __builtin_amdgcn_raw_buffer_store_b64(w2, r_out, (int)ob[j], so_o, 0);
                        if constexpr (X3) {
                            u32x2 l2;
                            l2[0] = pack_bf16(v[j][0] - bf_lo(w2[0]), v[j][1] - bf_hi(w2[0]));
                            l2[1] = pack_bf16(v[j][2] - bf_lo(w2[1]), v[j][3] - bf_hi(w2[1]));
                            __builtin_amdgcn_raw_buffer_store_b64(
                                l2, r_out, (int)ob[j], so_o + (int)((unsigned)(oct8 * OPL) * 16u), 0);
                        }
                    }
                } else if (cok && p.out_kind == 2) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[j]), r_out,
                                                               (int)ob[j], so_o, 0);
                }
                {
                    if (pooling && cok) {
                        // 2x2 max-pool of fp32 values + DePool2D mask bits (y & 1) * 2 + (x & 1):
                        // pre == pooled (layers/mylayers.py:111-114), window = rows (j = 0, 1) x
                        // columns (lane, lane ^ 1); the even lane stores
#pragma unroll
                        for (int jp = 0; jp < TN / 2; ++jp) {
                        // (instruction count matters: on the 1- and 4-k-tile pooled layers this epilogue
                        // is most of a workgroup -- rocprofv3 on the 16 -> 64 first layer: 1240 VALU +
                        // 740 SALU instructions per wave around 36 MFMAs.  Vertical max first, ONE lane
                        // exchange for the column partner's max, the four channels' bit pairs packed
                        // into one dword before the single exchange of the partner's bits.)
                        f32x4 m;
                        unsigned own = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float a0 = v[2 * jp][q], a1 = v[2 * jp + 1][q];
                            const float mv = fmaxf(a0, a1);
                            m[q] = fmaxf(mv, dpp_xor1(mv));
                            own |= ((a0 == m[q] ? 1u : 0u) | (a1 == m[q] ? 4u : 0u)) << (8 * q);
                        }
                        // byte q of `own`: bit 0 = row 0, bit 2 = row 1 of THIS lane's column; shifted
                        // to the column's bit position (x & 1) and merged with the partner's
                        own <<= (unsigned)(l31 & 1);
                        const unsigned mb = own | dpp_xor1(own);
                        const int so_p = (int)((unsigned)(c8 * PPL) * 16u);
                        u32x2 w2;
                        w2[0] = pack_bf16(m[0], m[1]); w2[1] = pack_bf16(m[2], m[3]);
                        __builtin_amdgcn_raw_buffer_store_b64(w2, r_pool, (int)qb[jp], so_p, 0);
                        if constexpr (X3) {
                            u32x2 l2;
                            l2[0] = pack_bf16(m[0] - bf_lo(w2[0]), m[1] - bf_hi(w2[0]));
                            l2[1] = pack_bf16(m[2] - bf_lo(w2[1]), m[3] - bf_hi(w2[1]));
                            __builtin_amdgcn_raw_buffer_store_b64(
                                l2, r_pool, (int)qb[jp], so_p + (int)((unsigned)(co8 * PPL) * 16u), 0);
                        }
                        __builtin_amdgcn_raw_buffer_store_b32((int)mb, r_mask, (int)qm[jp], so_p >> 1, 0);
                        }
                    }
                }
            }
        }
    }
}

// x (B, C, H, W) fp32 -> C8 (B, C8n, H, W, 8) bf16, channels >= C zero; X3: the hi / lo pair
// (B, 2 C8n, H, W, 8), lo = bf16(x - hi)
// (non-X3: `out` may be a wider C8 tensor of C8tot chunk planes per image, the C8n planes written start at
// plane c8_0; dense: C8tot = C8n, c8_0 = 0)
template <bool X3>
__global__ __launch_bounds__(256) void nchw_to_c8_kernel(const float* __restrict__ x, uint4* __restrict__ out,
                                                         int C, int HW, int C8n, int64_t total, int C8tot = 0,
                                                         int c8_0 = 0) {
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int pix = (int)(t % HW);
        const int64_t r = t / HW;
        const int c8 = (int)(r % C8n);
        const int64_t b = r / C8n;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // (unconditional loads, channels past C clamped and zeroed afterwards: a conditional load sits in a
            // basic block of its own behind a full wait -- eight memory round trips in a row)
            const int c = c8 * 8 + j;
            const float t8 = x[(b * C + (c < C ? c : C - 1)) * HW + pix];
            v[j] = c < C ? t8 : 0.f;
        }
        const uint4 hi = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]),
                                    pack_bf16(v[6], v[7]));
        if constexpr (X3) {
            const int64_t o = ((b * 2 * C8n + c8) * HW) + pix;
            out[o] = hi;
            out[o + (int64_t)C8n * HW] =
                make_uint4(pack_bf16(v[0] - bf_lo(hi.x), v[1] - bf_hi(hi.x)),
                           pack_bf16(v[2] - bf_lo(hi.y), v[3] - bf_hi(hi.y)),
                           pack_bf16(v[4] - bf_lo(hi.z), v[5] - bf_hi(hi.z)),
                           pack_bf16(v[6] - bf_lo(hi.w), v[7] - bf_hi(hi.w)));
        } else {
            out[C8tot ? ((b * C8tot + c8_0 + c8) * HW + pix) : t] = hi;
        }
    }
}

// C8 (B, C8n, H, W, 8) bf16 -> (B, C, H, W) fp32 (first C channels); X3: from the hi / lo pair
// (B, 2 C8n, H, W, 8), value = hi + lo (exact in fp32)
// (non-X3: `x` may be a wider C8 tensor of C8tot chunk planes per image, read from plane c8_0)
template <bool X3>
__global__ __launch_bounds__(256) void c8_to_nchw_kernel(const uint4* __restrict__ x, float* __restrict__ out,
                                                         int C, int HW, int C8n, int64_t total, int C8tot = 0,
                                                         int c8_0 = 0) {
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int pix = (int)(t % HW);
        const int64_t r = t / HW;
        const int c8 = (int)(r % C8n);
        const int64_t b = r / C8n;
        const uint4 u = x[X3 ? ((b * 2 * C8n + c8) * HW + pix) : C8tot ? ((b * C8tot + c8_0 + c8) * HW + pix) : t];
        float v[8] = {bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y),
                      bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w)};
        if constexpr (X3) {
            const uint4 w = x[(b * 2 * C8n + C8n + c8) * HW + pix];
            v[0] += bf_lo(w.x); v[1] += bf_hi(w.x); v[2] += bf_lo(w.y); v[3] += bf_hi(w.y);
            v[4] += bf_lo(w.z); v[5] += bf_hi(w.z); v[6] += bf_lo(w.w); v[7] += bf_hi(w.w);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c8 * 8 + j;
            if (c < C) out[(b * C + c) * HW + pix] = v[j];
        }
    }
}

// 2x2 max-pool (ignore_border) of the window (y0, x0, wh, ww) -- pooled coordinates -- of a C8 map
// `pre` (B, C8n, PH, PW, 8; F32: fp32 chunks, the unrounded conv results, else bf16), whose top-left
// corner sits at (py0, px0) of the full (H, W) map: pooled (B, C8n, H/2, W/2, 8) bf16 and, if
// mask != NULL, the DePool2D mask bytes.  With F32 the comparisons are those of the fused epilogue of
// conv_c8_kernel (fp32 values, rounded after the max), so a level gives the same masks whichever
// pixel tiling its conv ran on.
// x3_c8n > 0 (F32 only): pooled is a hi / lo pair (B, 2 x3_c8n, H/2, W/2, 8), the mask is not.
template <bool F32>
__global__ __launch_bounds__(256) void pool_mask_c8_kernel(const uint4* __restrict__ pre, uint4* __restrict__ pooled,
                                                           uint2* __restrict__ mask, int PH, int PW,
                                                           int py0, int px0, int h2, int w2, int y0, int x0,
                                                           int wh, int ww, int x3_c8n, int64_t total) {
    constexpr int Q = F32 ? 2 : 1;                  // uint4 per chunk
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int qx = (int)(t % ww);
        int64_t r = t / ww;
        const int qy = (int)(r % wh);
        r /= wh;                                    // b * C8n + c8
        const int Y = y0 + qy, X = x0 + qx;
        const uint4* s = pre + ((r * PH + (2 * Y - py0)) * PW + (2 * X - px0)) * Q;
        float v[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4* sk = s + ((k >> 1) * PW + (k & 1)) * Q;
            if constexpr (F32) {
                const uint4 a = sk[0], b = sk[1];
                v[k][0] = __builtin_bit_cast(float, a.x); v[k][1] = __builtin_bit_cast(float, a.y);
                v[k][2] = __builtin_bit_cast(float, a.z); v[k][3] = __builtin_bit_cast(float, a.w);
                v[k][4] = __builtin_bit_cast(float, b.x); v[k][5] = __builtin_bit_cast(float, b.y);
                v[k][6] = __builtin_bit_cast(float, b.z); v[k][7] = __builtin_bit_cast(float, b.w);
            } else {
                const uint4 u = sk[0];
                v[k][0] = bf_lo(u.x); v[k][1] = bf_hi(u.x); v[k][2] = bf_lo(u.y); v[k][3] = bf_hi(u.y);
                v[k][4] = bf_lo(u.z); v[k][5] = bf_hi(u.z); v[k][6] = bf_lo(u.w); v[k][7] = bf_hi(u.w);
            }
        }
        float m[8];
        unsigned bits[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            m[j] = fmaxf(fmaxf(v[0][j], v[1][j]), fmaxf(v[2][j], v[3][j]));
            bits[j] = (v[0][j] == m[j] ? 1u : 0u) | (v[1][j] == m[j] ? 2u : 0u) |
                      (v[2][j] == m[j] ? 4u : 0u) | (v[3][j] == m[j] ? 8u : 0u);
        }
        const int64_t o = (r * h2 + Y) * w2 + X;
        const uint4 hi = make_uint4(pack_bf16(m[0], m[1]), pack_bf16(m[2], m[3]), pack_bf16(m[4], m[5]),
                                    pack_bf16(m[6], m[7]));
        if (x3_c8n > 0) {
            const int64_t bb = r / x3_c8n, c8 = r - bb * x3_c8n;
            const int64_t oh = (((bb * 2 * x3_c8n + c8) * h2) + Y) * w2 + X;
            pooled[oh] = hi;
            pooled[oh + (int64_t)x3_c8n * h2 * w2] =
                make_uint4(pack_bf16(m[0] - bf_lo(hi.x), m[1] - bf_hi(hi.x)),
                           pack_bf16(m[2] - bf_lo(hi.y), m[3] - bf_hi(hi.y)),
                           pack_bf16(m[4] - bf_lo(hi.z), m[5] - bf_hi(hi.z)),
                           pack_bf16(m[6] - bf_lo(hi.w), m[7] - bf_hi(hi.w)));
        } else {
            pooled[o] = hi;
        }
        if (mask)
            mask[o] = make_uint2(bits[0] | (bits[1] << 8) | (bits[2] << 16) | (bits[3] << 24),
                                 bits[4] | (bits[5] << 8) | (bits[6] << 16) | (bits[7] << 24));
    }
}

// DePool2D materialised on C8 tensors (layers/mylayers.py:88-115): out[2 qy + dy][2 qx + dx] = up[qy][qx] where
// bit dy * 2 + dx of the mask byte of the channel is set, else 0, for the pooled-coordinate window
// (y0, x0, wh, ww); `out` (BC8, H, W, 8) full-size planes (rows / columns >= 2 h2 / 2 w2 are never written: the
// caller keeps them zero).  One thread = one pooled element of a chunk: a 16-byte load + 8 mask bytes in, four
// 16-byte stores out.  For the DEEP decoder levels (>= 1024 input channels, 10^2 - 19^2 windows): their conv then
// stages its patch by LDS-DMA like any plain layer instead of selecting every chunk through registers once per
// output-channel tile -- the unpooled map of such a level is a few tens of MB, the register staging costs the
// conv 15-25 % (profiles/r05_c8_unpool_materialise.txt).
__global__ __launch_bounds__(256) void unpool_c8_kernel(const uint4* __restrict__ up, const uint2* __restrict__ mask,
                                                        uint4* __restrict__ out, int H, int W, int h2, int w2,
                                                        int y0, int x0, int wh, int ww, int64_t total) {
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int qx = x0 + (int)(t % ww);
        int64_t r = t / ww;
        const int qy = y0 + (int)(r % wh);
        r /= wh;                                    // b * C8n + c8
        const int64_t pi = (r * h2 + qy) * w2 + qx;
        const uint4 u = up[pi];
        const uint2 m = mask[pi];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // (the select of conv_c8_kernel's DePool2D staging: bit k of each mask byte -> 0x0000 / 0xffff per channel)
            const unsigned t0 = (m.x >> k) & 0x01010101u, t1 = (m.y >> k) & 0x01010101u;
            const unsigned b0 = (t0 << 8) - t0, b1 = (t1 << 8) - t1;
            uint4 v;
            v.x = u.x & __builtin_amdgcn_perm(b0, b0, 0x01010000u);
            v.y = u.y & __builtin_amdgcn_perm(b0, b0, 0x03030202u);
            v.z = u.z & __builtin_amdgcn_perm(b1, b1, 0x01010000u);
            v.w = u.w & __builtin_amdgcn_perm(b1, b1, 0x03030202u);
            out[(r * H + 2 * qy + (k >> 1)) * W + 2 * qx + (k & 1)] = v;
        }
    }
}

// IISEG_CONV_X3 weights: w[co][c][3][3] (strides so, sc) -> out (Cout, 2 Cp, 3, 3) fp32 = [W_hi | W_lo],
// W_hi = bf16(w) and W_lo = bf16(w - W_hi) as floats, channels >= Cin of each group zero -- the filter
// iiseg_conv_halo_bf16_pack then packs unchanged (every value is a bf16 number)
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, int64_t so, int64_t sc,
                                                            float* __restrict__ out, int Cin, int Cp,
                                                            int64_t total) {
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int tap = (int)(t % 9);
        const int64_t r = t / 9;
        const int c = (int)(r % Cp);
        const int64_t co = r / Cp;
        const float v = c < Cin ? w[co * so + c * sc + tap] : 0.f;
        const float hi = (float)(__bf16)v;
        const float lo = (float)(__bf16)(v - hi);
        out[(co * 2 * Cp + c) * 9 + tap] = hi;
        out[(co * 2 * Cp + Cp + c) * 9 + tap] = lo;
    }
}

// FLAT tiling: patch rows a tile can need (worst case over tile positions).  Row-major: tile t starts
// at pixel 256 t of the stacked window list; quad: at pooling window 64 t of the stacked list of
// QH x QW windows per image (its pixels: rows 2 qy, 2 qy + 1).  The first / last unit of a tile fix the
// virtual rows it spans.  The start offsets inside an image repeat with period units / gcd(step, units)
// tiles, so at most that many (and never more than the launch has) are looked at; cached.
int flat_patch_rows(int B, int OH, int OW, bool quad) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, int> cache;
    const auto key = std::make_tuple(B, OH, OW, quad ? 1 : 0);
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    const int UW = quad ? (OW + 1) / 2 : OW;                  // units per row, rows of units per image
    const int UH = quad ? (OH + 1) / 2 : OH;
    const int64_t step = quad ? 64 : 256;
    const int64_t upi = (int64_t)UH * UW, N = (int64_t)B * upi;
    const int64_t npt = (N + step - 1) / step;
    int64_t a = step, b = upi;
    while (b) { const int64_t t = a % b; a = b; b = t; }      // gcd(step, units per image)
    const int64_t period = upi / a;
    int worst = 0;
    for (int64_t t = 0; t < npt; ++t) {
        const int64_t n0 = t * step, n1 = (n0 + step - 1 < N - 1) ? n0 + step - 1 : N - 1;
        const int b0 = (int)(n0 / upi), r0 = (int)(n0 % upi);
        const int b1 = (int)(n1 / upi), r1 = (int)(n1 % upi);
        const int rs = quad ? 2 : 1;                          // pixel rows per unit row
        const int v0 = b0 * (OH + 2) + rs * (r0 / UW), v1 = b1 * (OH + 2) + rs * (r1 / UW) + (rs - 1);
        if (v1 - v0 + 3 > worst) worst = v1 - v0 + 3;
        if (t >= period && t + 1 < npt) t = npt - 2;   // one full period seen: only the last tile is left
    }
    std::lock_guard<std::mutex> g(mu);
    cache[key] = worst;
    return worst;
}

// test hook (iiseg_conv_c8_force_tiling): kind -1 automatic, 0 rect-256, 1 rect-512, 2 flat; th, tw > 0
// fix the RECT shape (ignored when it cannot run the launch)
int g_force_kind = -1, g_force_th = 0, g_force_tw = 0;

struct C8Plan {
    bool flat;
    bool tall;      // RECT with 512-pixel tiles (TN = 4)
    bool quad;      // pixels of a tile ordered by 2x2 pooling windows (needed by a fused pool)
    bool single;    // one k-tile: the single-stage kernel (NBUF = 1, three workgroups per CU), 256-pixel tiles
    int th, tw;     // RECT tile shape
    int PR;
};

int c8_check(const iiseg_conv_desc* d, C8Plan* plan, bool pool = false) {
    if (!d) return IISEG_ERR_NULL;
    if (d->KH != 3 || d->KW != 3 || d->dil != 1 || (d->flags & IISEG_CONV_TRANSPOSED2))
        return IISEG_ERR_UNSUPPORTED;
    if (d->B <= 0 || d->C1 <= 0 || d->C2 < 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 ||
        d->pad < 0 || d->OH <= 0 || d->OW <= 0 || d->oy0 < 0 || d->ox0 < 0)
        return IISEG_ERR_SHAPE;
    if (d->C1 % 16 || d->C2 % 16) return IISEG_ERR_UNSUPPORTED;     // whole k-tiles per source
    const bool x3 = (d->flags & IISEG_CONV_X3) != 0;
    if (x3 && d->C2 != 0) return IISEG_ERR_UNSUPPORTED;
    const int fullH = d->H + 2 * d->pad - 2, fullW = d->W + 2 * d->pad - 2;
    if (d->oy0 + d->OH > fullH || d->ox0 + d->OW > fullW) return IISEG_ERR_SHAPE;
    if ((d->flags & IISEG_CONV_UNPOOL) && d->C2 != 0) return IISEG_ERR_UNSUPPORTED;
    if (d->out_ctot != 0 && (d->out_c0 < 0 || d->out_c0 + d->Cout > d->out_ctot)) return IISEG_ERR_SHAPE;
    if (d->out_H != 0 && (d->out_y0 < 0 || d->out_x0 < 0 || d->out_y0 + d->OH > d->out_H ||
                          d->out_x0 + d->OW > d->out_W))
        return IISEG_ERR_SHAPE;
    const int64_t cmax = (d->C1 > d->C2 ? d->C1 : d->C2) * (x3 ? 2 : 1);
    // RECT: one image of every tensor is addressed with 32-bit byte offsets
    if (cmax * d->H * d->W * 2 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    const int64_t octot = d->out_ctot ? d->out_ctot : d->Cout;
    const int64_t opl = d->out_H ? (int64_t)d->out_H * d->out_W : (int64_t)d->OH * d->OW;
    if ((octot + 64) * opl * 4 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;   // (x3 pair: 2 x 2 bytes)
    if (((int64_t)d->Cout + 64) * d->AH * d->AW * 4 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    const int bm = d->Cout > 32 ? 64 : 32, mpad = (d->Cout + bm - 1) / bm * bm;
    const int nkt = (d->C1 + d->C2) / 16 * (x3 ? 2 : 1);      // packed weight k-tiles
    if ((int64_t)nkt * 18 * mpad * 16 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    if (plan) {
        // Three candidates, priced by a small cost model (per-CU work: tiles are dealt to 256 CUs, a
        // tile costs its MFMA count plus a fixed set-up / epilogue share measured at about two k-tiles
        // of a 256-pixel tile, three of a 512-pixel one):
        //   RECT-256 / RECT-512  th x tw pixel tiles of one image, shape chosen for the fewest tiles
        //   FLAT                 256 consecutive pixels (64 consecutive pooling windows) of the whole batch:
        //                        no overhang at all, but a patch of whole window rows -- small windows
        // A fused pool needs the quad pixel order (even tile shapes / the quad flat list).
        const bool quad = pool;
        static const int force = getenv("IISEG_C8_TILING") ? atoi(getenv("IISEG_C8_TILING")) : 0;  // 1 rect, 2 flat
        static const int tall_env = getenv("IISEG_C8_TALL") ? atoi(getenv("IISEG_C8_TALL")) : -1;
        static const char* shape_env = getenv("IISEG_C8_SHAPE");          // "th,tw": tests
        const int64_t mt = (d->Cout + 63) / 64;
        auto per_cu = [&](int64_t tiles, double unit) { return (double)((tiles * mt + 255) / 256) * unit; };
        // staging share of a k-tile: patch chunks staged per output pixel beyond a compact tile's 1.3 cost
        // 2 % each by LDS-DMA, 8 % each through registers (DePool2D: select per chunk) -- fitted to
        // conv4_3 at 40^2 (flat 0.92 of rect) and up_conv3 at 58^2 (flat 1.07 of rect)
        const double gam = (d->flags & IISEG_CONV_UNPOOL) ? 0.08 : 0.02;
        auto stage = [&](int patch_half, int pixels) {
            const double ppp = 2.0 * patch_half / pixels - 1.3;
            return 1.0 + gam * (ppp > 0 ? ppp : 0);
        };
        int th2, tw2, th4, tw4;
        int64_t n2, n4;
        rect_shape(d->OH, d->OW, 256, C8_PCAP, quad, &th2, &tw2, &n2);
        rect_shape(d->OH, d->OW, 512, C8_PCAP, quad, &th4, &tw4, &n4);
        const double c2 = per_cu(n2 * d->B, 2.0 * (nkt * stage((th2 + 2) * (tw2 + 2), 256) + 2.0)),
                     c4 = per_cu(n4 * d->B, 4.0 * nkt * stage((th4 + 2) * (tw4 + 2), 512) + 6.0);
        static const int single_env = getenv("IISEG_C8_SINGLE") ? atoi(getenv("IISEG_C8_SINGLE")) : 1;
        plan->single = single_env && nkt == 1 && !x3 && !(d->flags & IISEG_CONV_UNPOOL) && d->Cout > 32;
        plan->flat = false;
        plan->quad = quad;
        plan->PR = 0;
        plan->tall = c4 < c2 && !plan->single;
        if (tall_env >= 0) plan->tall = tall_env != 0;
        if (g_force_kind == 0 || g_force_kind == 1) plan->tall = g_force_kind == 1;
        if (plan->tall) plan->single = false;
        plan->th = plan->tall ? th4 : th2;
        plan->tw = plan->tall ? tw4 : tw2;
        {
            int a = g_force_th, b = g_force_tw;
            if (shape_env && a <= 0) sscanf(shape_env, "%d,%d", &a, &b);
            if (a > 0 && b > 0 && (!quad || !((a | b) & 1)) && (a + 2) * (b + 2) <= C8_PCAP && a * b <= 512) {
                plan->th = a; plan->tw = b; plan->tall = a * b > 256;
                if (plan->tall) plan->single = false;
            }
        }
        // FLAT: single source, the whole tensors within 32-bit offsets, the patch within the buffer
        if (force != 1 && g_force_kind != 0 && g_force_kind != 1 && d->C2 == 0 &&
            (int64_t)d->B * cmax * d->H * d->W * 2 < (1ll << 31) &&
            (int64_t)d->B * (octot + 64) * opl * 4 < (1ll << 31) &&
            (int64_t)d->B * ((int64_t)d->Cout + 64) * d->AH * d->AW * 4 < (1ll << 31) &&
            (!pool || (int64_t)d->B * (d->Cout + 64) * (fullH / 2) * (fullW / 2) * 4 < (1ll << 31))) {
            const int pr = flat_patch_rows(d->B, d->OH, d->OW, quad);
            if (pr * (d->OW + 2) <= C8_PCAP) {
                const int64_t units = quad ? (int64_t)d->B * ((d->OH + 1) / 2) * ((d->OW + 1) / 2)
                                           : (int64_t)d->B * d->OH * d->OW;
                const int64_t nf = (units + (quad ? 63 : 255)) / (quad ? 64 : 256);
                const double cf = per_cu(nf, 2.0 * (nkt * stage(pr * (d->OW + 2), 256) + 2.5));
                const double cr = plan->tall ? c4 : c2;
                // (against 512-pixel tiles the flat list has to be 8 % ahead: 512 -> 512 at 37^2 runs 13 %
                // slower flat at equal modelled cost -- twice the weight DMA and barriers per pixel)
                if ((cf < (plan->tall ? 0.92 : 1.0) * cr && !plan->single) || force == 2 || g_force_kind == 2) {
                    plan->flat = true;
                    plan->tall = false;
                    plan->single = false;
                    plan->PR = pr;
                }
            }
        }
    }
    return IISEG_OK;
}

template <int BM, bool OUTF32>
int launch_c8(hipStream_t s, C8Params& p, const C8Plan& plan, bool unpool) {
    p.n_mtiles = p.Mpad / BM;
    p.quad = plan.quad ? 1 : 0;
    if (plan.flat) {
        p.QH = (p.OH + 1) / 2; p.QW = (p.OW + 1) / 2;
        p.N = plan.quad ? p.B * p.QH * p.QW : p.B * p.OH * p.OW;
        p.PR = plan.PR;
        p.PWs = p.OW + 2;
        p.n_ptiles = plan.quad ? (p.N + 63) / 64 : (p.N + 255) / 256;
        p.pw_magic = magic20(p.PWs);
        p.tw_magic = 0;
    } else {
        p.th = plan.th; p.tw = plan.tw;
        p.tiles_y = (p.OH + p.th - 1) / p.th;
        p.tiles_x = (p.OW + p.tw - 1) / p.tw;
        p.n_ptiles = p.B * p.tiles_y * p.tiles_x;
        p.pw_magic = magic20(p.tw + 2);
        p.tw_magic = magic20(plan.quad ? p.tw / 2 : p.tw);
    }
    const int grid = p.n_ptiles * p.n_mtiles;
    // the frequent epilogue combinations have straight-line code of their own (template EPI)
    static const int epi_env = getenv("IISEG_C8_EPI") ? atoi(getenv("IISEG_C8_EPI")) : 1;
    if constexpr (BM == 64 && !OUTF32) {
        int epi = EPI_GENERIC;
        if (epi_env) {
            if (p.pool && !p.out && !p.add && !unpool && p.relu) epi = EPI_POOL;
            else if (p.pool && !p.out && p.add && p.add_kind == 2 && !unpool && p.relu && !plan.single)
                epi = EPI_POOL_ADD2;
            else if (!p.pool && p.out && p.out_kind == 1 && !p.add && !unpool) epi = EPI_STORE;
            else if (!p.pool && p.out && p.out_kind == 1 && p.add && p.add_kind == 1) epi = EPI_STORE_ADD;
        }
#define C8_LAUNCH_EPI_X(E, UN, XX)                                                                 \
        do {                                                                                       \
            if (plan.flat)                                                                         \
                IISEG_LAUNCH((conv_c8_kernel<64, 2, true, UN, false, XX, 2, 4, E>), dim3(grid), dim3(256), 0, s, p); \
            else if (plan.tall)                                                                    \
                IISEG_LAUNCH((conv_c8_kernel<64, 4, false, UN, false, XX, 2, 4, E>), dim3(grid), dim3(256), 0, s, p); \
            else                                                                                   \
                IISEG_LAUNCH((conv_c8_kernel<64, 2, false, UN, false, XX, 2, 4, E>), dim3(grid), dim3(256), 0, s, p); \
            return iiseg_check_launch();                                                           \
        } while (0)
#define C8_LAUNCH_EPI(E, UN)                                                                       \
        do {                                                                                       \
            if (p.x3) C8_LAUNCH_EPI_X(E, UN, true); else C8_LAUNCH_EPI_X(E, UN, false);            \
        } while (0)
        if (epi == EPI_POOL) {
            if (plan.single) {
                IISEG_LAUNCH((conv_c8_kernel<64, 2, false, false, false, false, 1, 4, EPI_POOL>), dim3(grid), dim3(256), 0, s, p);
                return iiseg_check_launch();
            }
            C8_LAUNCH_EPI(EPI_POOL, false);
        } else if (epi == EPI_POOL_ADD2) {
            C8_LAUNCH_EPI(EPI_POOL_ADD2, false);
        } else if (epi == EPI_STORE) {
            if (plan.single) {
                IISEG_LAUNCH((conv_c8_kernel<64, 2, false, false, false, false, 1, 4, EPI_STORE>), dim3(grid), dim3(256), 0, s, p);
                return iiseg_check_launch();
            }
            C8_LAUNCH_EPI(EPI_STORE, false);
        } else if (epi == EPI_STORE_ADD) {
            if (unpool) C8_LAUNCH_EPI(EPI_STORE_ADD, true);
            else C8_LAUNCH_EPI(EPI_STORE_ADD, false);
        }
#undef C8_LAUNCH_EPI
#undef C8_LAUNCH_EPI_X
    }
#define C8_LAUNCH(TNV, FL, UN)                                                                     \
    do {                                                                                           \
        if (p.x3)                                                                                  \
            IISEG_LAUNCH((conv_c8_kernel<BM, TNV, FL, UN, OUTF32, true>), dim3(grid), dim3(256), 0, s, p); \
        else                                                                                       \
            IISEG_LAUNCH((conv_c8_kernel<BM, TNV, FL, UN, OUTF32, false>), dim3(grid), dim3(256), 0, s, p); \
    } while (0)
    if (plan.single) {
        if constexpr (BM == 64 && !OUTF32) {
            IISEG_LAUNCH((conv_c8_kernel<64, 2, false, false, false, false, 1>), dim3(grid), dim3(256), 0, s, p);
        } else {
            return IISEG_ERR_UNSUPPORTED;
        }
    } else if (plan.flat) {
        if (unpool) C8_LAUNCH(2, true, true); else C8_LAUNCH(2, true, false);
    } else if (plan.tall) {
        // (eight-wave workgroups -- TN = 2, NW = 8: half the work and registers per wave, four waves per
        // SIMD -- measured within 0.5 % of this form on every layer class, with spilled registers: opt-in)
        static const int w8 = getenv("IISEG_C8_W8") ? atoi(getenv("IISEG_C8_W8")) : 0;
        if constexpr (BM == 64 && !OUTF32) {
            if (w8 && !unpool && !p.x3) {
                IISEG_LAUNCH((conv_c8_kernel<64, 2, false, false, false, false, 2, 8>), dim3(grid), dim3(512),
                                   0, s, p);
                return iiseg_check_launch();
            }
        }
        if (unpool) C8_LAUNCH(4, false, true); else C8_LAUNCH(4, false, false);
    } else {
        if (unpool) C8_LAUNCH(2, false, true); else C8_LAUNCH(2, false, false);
    }
#undef C8_LAUNCH
    return iiseg_check_launch();
}

}  // namespace

extern "C" int iiseg_conv_c8_supported(const iiseg_conv_desc* d) {
    return c8_check(d, nullptr) == IISEG_OK ? 1 : 0;
}

extern "C" int iiseg_conv_c8_is_flat(const iiseg_conv_desc* d) {
    C8Plan plan;
    if (c8_check(d, &plan) != IISEG_OK) return 0;
    return plan.flat ? 1 : 0;
}

extern "C" int iiseg_conv_c8_force_tiling(int kind, int th, int tw) {
    if (kind < -1 || kind > 2 || th < 0 || tw < 0) return IISEG_ERR_SHAPE;
    g_force_kind = kind; g_force_th = th; g_force_tw = tw;
    return IISEG_OK;
}

extern "C" int iiseg_conv_c8_tiling(const iiseg_conv_desc* d, int pool, int32_t* out4) {
    C8Plan plan;
    const int st = c8_check(d, &plan, pool != 0);
    if (st) return st;
    if (!out4) return IISEG_ERR_NULL;
    out4[0] = plan.flat ? 2 : (plan.tall ? 1 : 0);
    out4[1] = plan.flat ? plan.PR : plan.th;
    out4[2] = plan.flat ? d->OW + 2 : plan.tw;
    out4[3] = plan.quad ? 1 : 0;
    return IISEG_OK;
}

extern "C" int iiseg_conv_c8(void* stream, const iiseg_conv_desc* d, const void* x1, const void* x2,
                             const uint8_t* mask_in, const void* wp16, const float* bias,
                             const void* add, int add_kind, void* out, int out_kind, void* pool_out,
                             uint8_t* mask_out) {
    return iiseg_conv_c8_slice(stream, d, x1, 0, x2, mask_in, wp16, bias, add, add_kind, out, out_kind, pool_out,
                               mask_out);
}

extern "C" int iiseg_conv_c8_slice(void* stream, const iiseg_conv_desc* d, const void* x1, int32_t x1_ctot,
                                   const void* x2, const uint8_t* mask_in, const void* wp16, const float* bias,
                                   const void* add, int add_kind, void* out, int out_kind, void* pool_out,
                                   uint8_t* mask_out) {
    C8Plan plan;
    const int st = c8_check(d, &plan, pool_out != nullptr);
    if (st) return st;
    if (!x1 || !wp16) return IISEG_ERR_NULL;
    if (d->C2 > 0 && !x2) return IISEG_ERR_NULL;
    const bool unpool = (d->flags & IISEG_CONV_UNPOOL) != 0;
    if (unpool != (mask_in != nullptr)) return IISEG_ERR_NULL;
    if ((uintptr_t)wp16 & 15) return IISEG_ERR_ALIGN;
    const bool zins = (d->flags & IISEG_CONV_ZINS) != 0;
    const bool x3f = (d->flags & IISEG_CONV_X3) != 0;
    // x1 as a channel slice of a wider tensor / zero-inserted: plain single-source bf16 launches
    if ((x1_ctot != 0 || zins) && (x3f || unpool)) return IISEG_ERR_UNSUPPORTED;
    if (x1_ctot != 0 && (x1_ctot % 8 || x1_ctot < d->C1)) return IISEG_ERR_SHAPE;
    if (zins && (d->C2 != 0 || d->pad != 0 || d->H < 5 || d->W < 5 || !(d->H & 1) || !(d->W & 1)))
        return IISEG_ERR_SHAPE;
    if (x1_ctot != 0 && (int64_t)d->B * x1_ctot * d->H * d->W * 2 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    if (out_kind < 0 || out_kind > 3 || add_kind < 0 || add_kind > 2) return IISEG_ERR_UNSUPPORTED;
    if ((out_kind != 0) != (out != nullptr)) return IISEG_ERR_NULL;
    if ((add_kind != 0) != (add != nullptr)) return IISEG_ERR_NULL;
    if (!out && !pool_out) return IISEG_ERR_NULL;
    if (mask_out && !pool_out) return IISEG_ERR_UNSUPPORTED;
    if (add && (d->AH < d->ay0 + d->OH || d->AW < d->ax0 + d->OW || d->ay0 < 0 || d->ax0 < 0))
        return IISEG_ERR_SHAPE;
    const int octot = d->out_ctot ? d->out_ctot : d->Cout;
    if (out_kind == 3) {                          // NCHW fp32: the class-score layer
        if (d->Cout > 32 || add || pool_out) return IISEG_ERR_UNSUPPORTED;
    } else if (d->out_ctot != 0 && (octot % 16 || d->out_c0 % 16)) {
        return IISEG_ERR_UNSUPPORTED;
    }
    const int fullH = d->H + 2 * d->pad - 2, fullW = d->W + 2 * d->pad - 2;
    if (pool_out) {
        // fused pool: whole pooling windows (even origin, even extent unless the window ends at the
        // map's last, unpaired row / column)
        if (((d->oy0 | d->ox0) & 1) || ((d->OH & 1) && d->oy0 + d->OH != fullH) ||
            ((d->OW & 1) && d->ox0 + d->OW != fullW))
            return IISEG_ERR_UNSUPPORTED;
    }
    C8Params p = {};
    p.x1 = x1; p.x2 = x2; p.mask_in = mask_in; p.wp = wp16; p.bias = bias;
    p.add = add; p.add_kind = add_kind; p.out = out; p.out_kind = out_kind;
    p.pool = pool_out; p.mask_out = mask_out;
    p.B = d->B; p.C1 = d->C1; p.C2 = d->C2; p.H = d->H; p.W = d->W;
    p.h2 = d->H / 2; p.w2 = d->W / 2;
    p.zins = zins ? 1 : 0;
    if (zins) { p.h2 = (d->H - 3) / 2; p.w2 = (d->W - 3) / 2; }
    p.in_ct8 = (x1_ctot ? x1_ctot : d->C1) / 8 * (x3f ? 2 : 1);
    p.Cout = d->Cout; p.OH = d->OH; p.OW = d->OW; p.oy0 = d->oy0; p.ox0 = d->ox0; p.pad = d->pad;
    p.AH = d->AH; p.AW = d->AW; p.ay0 = d->ay0; p.ax0 = d->ax0;
    p.x3 = (d->flags & IISEG_CONV_X3) ? 1 : 0;
    p.nkt = (d->C1 + d->C2) / 16 * (p.x3 ? 2 : 1);             // packed weight k-tiles
    const int bm = d->Cout > 32 ? 64 : 32;
    p.Mpad = (d->Cout + bm - 1) / bm * bm;
    // dense C8 output: the Cout channels padded to whole 16-channel groups
    p.out_ctot = (out_kind == 3 || d->out_ctot) ? octot : (d->Cout + 15) / 16 * 16;
    p.out_c0 = d->out_ctot ? d->out_c0 : 0;
    p.out_H = d->out_H ? d->out_H : d->OH;
    p.out_W = d->out_H ? d->out_W : d->OW;
    p.out_y0 = d->out_H ? d->out_y0 : 0;
    p.out_x0 = d->out_H ? d->out_x0 : 0;
    p.pool_H = fullH / 2; p.pool_W = fullW / 2;
    p.relu = (d->flags & IISEG_CONV_RELU) ? 1 : 0;
    static const int dbg = getenv("IISEG_BF16_DEBUG") ? atoi(getenv("IISEG_BF16_DEBUG")) : 0;
    p.debug = dbg;
    hipStream_t s = (hipStream_t)stream;
    if (out_kind == 3) return launch_c8<32, true>(s, p, plan, unpool);
    if (bm == 64) return launch_c8<64, false>(s, p, plan, unpool);
    return launch_c8<32, false>(s, p, plan, unpool);
}

extern "C" int iiseg_conv_c8_split_weights(void* stream, const float* w, int64_t stride_o,
                                           int64_t stride_c, int Cout, int Cin, float* out) {
    if (!w || !out) return IISEG_ERR_NULL;
    if (Cout <= 0 || Cin <= 0) return IISEG_ERR_SHAPE;
    const int Cp = (Cin + 15) / 16 * 16;
    const int64_t total = (int64_t)Cout * Cp * 9;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    IISEG_LAUNCH(split_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, stride_o,
                       stride_c, out, Cin, Cp, total);
    return iiseg_check_launch();
}

extern "C" int iiseg_nchw_to_c8(void* stream, const float* x, void* out, int B, int C, int H, int W,
                                int C8n) {
    if (!x || !out) return IISEG_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || C8n * 8 < C) return IISEG_ERR_SHAPE;
    const int64_t total = (int64_t)B * C8n * H * W;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    IISEG_LAUNCH(nchw_to_c8_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x,
                       (uint4*)out, C, H * W, C8n, total, 0, 0);
    return iiseg_check_launch();
}

extern "C" int iiseg_nchw_to_c8_slice(void* stream, const float* x, void* out, int B, int C, int H, int W,
                                      int C8tot, int c8_0) {
    if (!x || !out) return IISEG_ERR_NULL;
    const int C8n = (C + 7) / 8;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || c8_0 < 0 || c8_0 + C8n > C8tot) return IISEG_ERR_SHAPE;
    const int64_t total = (int64_t)B * C8n * H * W;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    IISEG_LAUNCH(nchw_to_c8_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x,
                       (uint4*)out, C, H * W, C8n, total, C8tot, c8_0);
    return iiseg_check_launch();
}

extern "C" int iiseg_c8_slice_to_nchw(void* stream, const void* x, float* out, int B, int C, int H, int W,
                                      int C8tot, int c8_0) {
    if (!x || !out) return IISEG_ERR_NULL;
    const int C8n = (C + 7) / 8;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || c8_0 < 0 || c8_0 + C8n > C8tot) return IISEG_ERR_SHAPE;
    const int64_t total = (int64_t)B * C8n * H * W;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    IISEG_LAUNCH(c8_to_nchw_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)x, out, C, H * W, C8n, total, C8tot, c8_0);
    return iiseg_check_launch();
}

extern "C" int iiseg_nchw_to_c8x3(void* stream, const float* x, void* out, int B, int C, int H, int W,
                                  int C8n) {
    if (!x || !out) return IISEG_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || C8n * 8 < C) return IISEG_ERR_SHAPE;
    const int64_t total = (int64_t)B * C8n * H * W;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    IISEG_LAUNCH(nchw_to_c8_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x,
                       (uint4*)out, C, H * W, C8n, total, 0, 0);
    return iiseg_check_launch();
}

extern "C" int iiseg_c8_to_nchw(void* stream, const void* x, float* out, int B, int C, int H, int W,
                                int C8n) {
    if (!x || !out) return IISEG_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || C8n * 8 < C) return IISEG_ERR_SHAPE;
    const int64_t total = (int64_t)B * C8n * H * W;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    IISEG_LAUNCH(c8_to_nchw_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)x, out, C, H * W, C8n, total, 0, 0);
    return iiseg_check_launch();
}

extern "C" int iiseg_c8x3_to_nchw(void* stream, const void* x, float* out, int B, int C, int H, int W,
                                  int C8n) {
    if (!x || !out) return IISEG_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || C8n * 8 < C) return IISEG_ERR_SHAPE;
    const int64_t total = (int64_t)B * C8n * H * W;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    IISEG_LAUNCH(c8_to_nchw_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)x, out, C, H * W, C8n, total, 0, 0);
    return iiseg_check_launch();
}

static int pool_mask_c8_impl(void* stream, const void* pre, int pre_f32, void* pooled, uint8_t* mask,
                             int BC8, int PH, int PW, int py0, int px0, int H, int W, int y0, int x0,
                             int wh, int ww, int x3_c8n) {
    if (!pre || !pooled) return IISEG_ERR_NULL;
    if (x3_c8n && (!pre_f32 || x3_c8n < 0 || BC8 % x3_c8n)) return IISEG_ERR_UNSUPPORTED;
    if (BC8 <= 0 || wh <= 0 || ww <= 0) return IISEG_ERR_SHAPE;
    // every 2x2 window of the pooled region must lie inside the stored piece of the pre-pool map
    if (y0 < 0 || x0 < 0 || y0 + wh > H / 2 || x0 + ww > W / 2 || 2 * y0 < py0 || 2 * x0 < px0 ||
        2 * (y0 + wh) > py0 + PH || 2 * (x0 + ww) > px0 + PW)
        return IISEG_ERR_SHAPE;
    const int64_t total = (int64_t)BC8 * wh * ww;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    if (pre_f32)
        IISEG_LAUNCH(pool_mask_c8_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const uint4*)pre, (uint4*)pooled, (uint2*)mask, PH, PW, py0, px0, H / 2,
                           W / 2, y0, x0, wh, ww, x3_c8n, total);
    else
        IISEG_LAUNCH(pool_mask_c8_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const uint4*)pre, (uint4*)pooled, (uint2*)mask, PH, PW, py0, px0, H / 2,
                           W / 2, y0, x0, wh, ww, 0, total);
    return iiseg_check_launch();
}

extern "C" int iiseg_pool_mask_c8(void* stream, const void* pre, int pre_f32, void* pooled,
                                  uint8_t* mask, int BC8, int PH, int PW, int py0, int px0, int H,
                                  int W, int y0, int x0, int wh, int ww) {
    return pool_mask_c8_impl(stream, pre, pre_f32, pooled, mask, BC8, PH, PW, py0, px0, H, W, y0, x0,
                             wh, ww, 0);
}

extern "C" int iiseg_pool_mask_c8x3(void* stream, const void* pre, void* pooled, uint8_t* mask,
                                    int B, int C8n, int PH, int PW, int py0, int px0, int H, int W,
                                    int y0, int x0, int wh, int ww) {
    if (B <= 0 || C8n <= 0) return IISEG_ERR_SHAPE;
    return pool_mask_c8_impl(stream, pre, 1, pooled, mask, B * C8n, PH, PW, py0, px0, H, W, y0, x0, wh,
                             ww, C8n);
}

extern "C" int iiseg_unpool_c8(void* stream, const void* up, const uint8_t* mask, void* out, int BC8, int H, int W,
                               int y0, int x0, int wh, int ww) {
    if (!up || !mask || !out) return IISEG_ERR_NULL;
    const int h2 = H / 2, w2 = W / 2;
    if (BC8 <= 0 || H < 2 || W < 2 || wh <= 0 || ww <= 0 || y0 < 0 || x0 < 0 || y0 + wh > h2 || x0 + ww > w2)
        return IISEG_ERR_SHAPE;
    const int64_t total = (int64_t)BC8 * wh * ww;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    IISEG_LAUNCH(unpool_c8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)up,
                       (const uint2*)mask, (uint4*)out, H, W, h2, w2, y0, x0, wh, ww, total);
    return iiseg_check_launch();
}
