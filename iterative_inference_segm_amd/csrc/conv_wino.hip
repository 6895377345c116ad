// Winograd F(2x2, 3x3) convolution for gfx950 (CDNA4), fp32.
//
// The fp32 matrix pipe (v_mfma_f32_32x32x2_f32, 157 TFLOP/s) is what bounds the wide 3x3 layers of
// the hot path, so the way to go faster at full fp32 precision is to issue fewer multiplies:
// Y = A^T [ (G g G^T) . (B^T d B) ] A computes a 2x2 output tile from a 4x4 input tile with 16
// instead of 36 multiplies per (cin, cout) pair (2.25x fewer MFMAs).  Three kernels:
//   1. wino_input_kernel   V[xi][c][t]  = (B^T d B)[xi]          HBM-bound, lanes along tiles
//   2. wino_gemm_kernel    M[xi][co][t] = sum_c U[xi][c][co] * V[xi][c][t]
//                          16 independent GEMMs, MFMA-bound, both operands stream global -> LDS
//                          with 16-byte LDS-DMA (no gather, no VALU address work)
//   3. wino_output_kernel  out = relu(A^T M A + bias + add), cropped to the window
// U = G g G^T is packed once per layer (wino_weight_kernel, computed in double).
//
// Tiles are anchored at ABSOLUTE output coordinates of a fixed parity (desc.tile_y0 / tile_x0, a
// per-layer constant chosen by the caller), so a pixel is produced by the same tile, the same 4x4
// patch and the same fixed-order sums whatever window of the layer is being computed: windowed /
// placed launches stay bit-identical to full-map launches (the property the decoder dead-code
// elimination and the loop-invariant encoder borders rely on).  The caller picks the parity of
// the window it launches most: a 10x10 window at an odd origin is 5x5 tiles instead of 6x6.
//
// Replaces the same Lasagne Conv2DLayer(3x3, stride 1) call sites as conv_taps.hip for layers
// with Cin % 16 == 0 (models/fcn8.py:41-71, models/fcn_down.py:102-104, models/fcn_up.py:83-86);
// fusions kept: two-source channel concat (h first), bias / skip-add / ReLU / window / placement.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "iiseg.h"
#include "common.h"
#include "conv_common.h"

using namespace iiseg;

namespace {

struct WinoParams {
    const float* x1;
    const float* x2;
    const float* pre;     // unpool mode: x1 = up, pre, pooled (DePool2D operands)
    const float* pooled;
    int h2, w2;
    const float* U;
    const float* bias;
    const float* add;
    float* V;
    float* M;
    float* out;
    int B, C1, C2, H, W;
    int Cout, pad;
    int oy0, ox0, OH, OW;    // output window in conv-output coordinates
    int ty0, tx0, nty, ntx;  // first tile's output row / column (absolute), tile counts
    int T, Tpad;             // B*nty*ntx, padded to the GEMM pixel tile
    int Kc, Mpad;            // channels (multiple of 16), output channels padded to the GEMM tile
    int AH, AW, ay0, ax0;
    int relu;
    int out_ctot, out_c0, out_H, out_W, out_y0, out_x0;
    int n_ttiles, n_mtiles;
};

constexpr int RSRC_W3 = 0x00027000;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const float* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, RSRC_W3);
}

// ---- 0. weights: U[xi][c][co] = (G g G^T)[xi], g = w[co][c] (cross-correlation, P1) -----------
__global__ void wino_weight_kernel(const float* __restrict__ w, int64_t so, int64_t sc,
                                   float* __restrict__ U, int Cin, int Cout, int Kc, int Mpad) {
    const int64_t n = (int64_t)Kc * Mpad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / Mpad), co = (int)(i % Mpad);
        double g[3][3], t[4][3];
        const bool real = c < Cin && co < Cout;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = real ? (double)w[co * so + c * sc + a * 3 + b] : 0.0;
#pragma unroll
        for (int b = 0; b < 3; ++b) {  // G g
            t[0][b] = g[0][b];
            t[1][b] = 0.5 * (g[0][b] + g[1][b] + g[2][b]);
            t[2][b] = 0.5 * (g[0][b] - g[1][b] + g[2][b]);
            t[3][b] = g[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {  // (G g) G^T
            const double u0 = t[a][0], u1 = 0.5 * (t[a][0] + t[a][1] + t[a][2]),
                         u2 = 0.5 * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
            U[(int64_t)(a * 4 + 0) * n + i] = (float)u0;
            U[(int64_t)(a * 4 + 1) * n + i] = (float)u1;
            U[(int64_t)(a * 4 + 2) * n + i] = (float)u2;
            U[(int64_t)(a * 4 + 3) * n + i] = (float)u3;
        }
    }
}

// ---- 1. input transform ------------------------------------------------------------------------
// One thread = one tile x ICH channels; lanes run along tiles (coalesced V stores).
// UNPOOL: the logical input is DePool2D(up = x1, pre, pooled) (layers/mylayers.py:88-115), formed
// while loading the patch: element (iy, ix) = pre == pooled[iy/2, ix/2] ? up[iy/2, ix/2] : 0 inside
// the 2h x 2w region, 0 outside.  PY / PX = parity of the patch origin (uniform over a launch), so
// the 3x3 block of pooled/up values a 4x4 patch touches is indexed at compile time.
constexpr int ICH = 4;
template <bool UNPOOL, int PY, int PX>
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoParams p) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= p.T) return;
    const int ntt = p.nty * p.ntx;
    const int b = t / ntt;
    const int r = t - b * ntt;
    const int tyl = r / p.ntx, txl = r - tyl * p.ntx;
    const int iy0 = p.ty0 + 2 * tyl - p.pad, ix0 = p.tx0 + 2 * txl - p.pad;
    int rowoff[4];
    bool rok[4], cok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rok[i] = (unsigned)(iy0 + i) < (unsigned)p.H;
        cok[i] = (unsigned)(ix0 + i) < (unsigned)p.W;
        rowoff[i] = (iy0 + i) * p.W + ix0;
    }
    // unpool: pooled-plane block rows/cols q = 0..2 cover patch rows (PY + i) >> 1
    constexpr int NQY = UNPOOL ? (PY ? 3 : 2) : 1, NQX = UNPOOL ? (PX ? 3 : 2) : 1;
    const int qy0 = (iy0 - PY) >> 1, qx0 = (ix0 - PX) >> 1;   // floor(iy0 / 2) for either parity
    bool qrok[NQY], qcok[NQX];
    int qoff[NQY];
    if constexpr (UNPOOL) {
#pragma unroll
        for (int i = 0; i < NQY; ++i) {
            qrok[i] = (unsigned)(qy0 + i) < (unsigned)p.h2;
            qoff[i] = (qy0 + i) * p.w2 + qx0;
        }
#pragma unroll
        for (int j = 0; j < NQX; ++j) qcok[j] = (unsigned)(qx0 + j) < (unsigned)p.w2;
    }
    const size_t HW = (size_t)p.H * p.W, hw2 = (size_t)p.h2 * p.w2;
    const size_t xis = (size_t)p.Kc * p.Tpad;
    // Phase 1: every load of the thread's channels, unconditional (offsets of elements outside the
    // image are clamped to element 0 of the plane and masked afterwards) so that all are in
    // flight together; phase 2: transform + store.  (Loads placed after a store could not be
    // moved up: V and the inputs may alias as far as the compiler knows.)
    constexpr int NC_ = UNPOOL ? ICH / 2 : ICH;
    const int c0 = blockIdx.y * NC_;
    float pv[NC_][4][4];
    float pq[UNPOOL ? NC_ : 1][NQY][NQX], uq[UNPOOL ? NC_ : 1][NQY][NQX];
#pragma unroll
    for (int cc = 0; cc < NC_; ++cc) {
        const int c = min(c0 + cc, p.Kc - 1);
        if constexpr (UNPOOL) {
            const float* prep = p.pre + ((size_t)b * p.C1 + c) * HW;
            const float* poolp = p.pooled + ((size_t)b * p.C1 + c) * hw2;
            const float* upp = p.x1 + ((size_t)b * p.C1 + c) * hw2;
#pragma unroll
            for (int i = 0; i < NQY; ++i)
#pragma unroll
                for (int j = 0; j < NQX; ++j) {
                    const int o = (qrok[i] && qcok[j]) ? qoff[i] + j : 0;
                    pq[cc][i][j] = poolp[o];
                    uq[cc][i][j] = upp[o];
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) pv[cc][i][j] = prep[(rok[i] && cok[j]) ? rowoff[i] + j : 0];
        } else {
            const float* src = c < p.C1 ? p.x1 + ((size_t)b * p.C1 + c) * HW
                                        : p.x2 + ((size_t)b * p.C2 + (c - p.C1)) * HW;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) pv[cc][i][j] = src[(rok[i] && cok[j]) ? rowoff[i] + j : 0];
        }
    }
#pragma unroll
    for (int cc = 0; cc < NC_; ++cc) {
        const int c = c0 + cc;
        if (c >= p.Kc) break;
        float d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (UNPOOL) {
                    const int qi = (PY + i) >> 1, qj = (PX + j) >> 1;
                    const bool ok = rok[i] && cok[j] && qrok[qi] && qcok[qj];
                    d[i][j] = (ok && pv[cc][i][j] == pq[cc][qi][qj]) ? uq[cc][qi][qj] : 0.f;
                } else {
                    d[i][j] = (rok[i] && cok[j]) ? pv[cc][i][j] : 0.f;
                }
            }
        float e[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // B^T d
            e[0][j] = d[0][j] - d[2][j];
            e[1][j] = d[1][j] + d[2][j];
            e[2][j] = d[2][j] - d[1][j];
            e[3][j] = d[1][j] - d[3][j];
        }
        float* v = p.V + (size_t)c * p.Tpad + t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // (B^T d) B
            v[(size_t)(i * 4 + 0) * xis] = e[i][0] - e[i][2];
            v[(size_t)(i * 4 + 1) * xis] = e[i][1] + e[i][2];
            v[(size_t)(i * 4 + 2) * xis] = e[i][2] - e[i][1];
            v[(size_t)(i * 4 + 3) * xis] = e[i][1] - e[i][3];
        }
    }
}


// LDS-staged input transform for maps with >= 128 tiles per image: a workgroup takes 256
// consecutive tiles of ONE image, stages the input rows they touch (whole rows of the tile grid:
// coalesced loads, ~1.3 loads per tile and channel instead of 16; the DePool2D mask is applied per
// staged element, 3 loads instead of 34 per tile) and every thread reads its 4x4 patch from LDS.
constexpr int ILDS_E = 12, ILDS_CH = 8;   // staged elements per thread, channels per workgroup
template <bool UNPOOL, int NT>
__global__ __launch_bounds__(NT) void wino_input_lds_kernel(const WinoParams p, const int chunks) {
    constexpr int ILDS_CAP = ILDS_E * NT;
    __shared__ __attribute__((aligned(16))) float Ls[2][ILDS_CAP];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
    const int ntt = p.nty * p.ntx;
    const int tl0 = chunk * NT, tl = tl0 + tid;
    const bool tvalid = tl < ntt;
    const int row_first = tl0 / p.ntx;
    const int row_last = min(ntt - 1, tl0 + NT - 1) / p.ntx;
    const int NR = 2 * (row_last - row_first + 1) + 2, NC = 2 * p.ntx + 2, NE = NR * NC;
    const int iyb = p.ty0 + 2 * row_first - p.pad, ixb = p.tx0 - p.pad;
    int goff[ILDS_E], qoff[UNPOOL ? ILDS_E : 1];
#pragma unroll
    for (int i = 0; i < ILDS_E; ++i) {
        const int e = i * NT + tid;
        const int r = e / NC, c = e - r * NC;
        const int iy = iyb + r, ix = ixb + c;
        bool ok = e < NE && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        goff[i] = ok ? iy * p.W + ix : -1;
        if constexpr (UNPOOL) {
            ok = ok && iy < 2 * p.h2 && ix < 2 * p.w2;
            qoff[i] = ok ? (iy >> 1) * p.w2 + (ix >> 1) : -1;
        }
    }
    const int tyl = tl / p.ntx, txl = tl - tyl * p.ntx;
    const int lbase = 2 * (tyl - row_first) * NC + 2 * txl;
    const size_t HW = (size_t)p.H * p.W, hw2 = (size_t)p.h2 * p.w2;
    const size_t xis = (size_t)p.Kc * p.Tpad;
    const int t = b * ntt + tl;
    const int c0 = blockIdx.y * ILDS_CH;
    float v[ILDS_E];

    auto fetch = [&](int c) __attribute__((always_inline)) {
        if constexpr (UNPOOL) {
            const float* prep = p.pre + ((size_t)b * p.C1 + c) * HW;
            const float* poolp = p.pooled + ((size_t)b * p.C1 + c) * hw2;
            const float* upp = p.x1 + ((size_t)b * p.C1 + c) * hw2;
            // all three loads of every staged element unconditionally (offsets of elements outside
            // the 2h x 2w region clamped to 0 and masked afterwards): one batch in flight instead
            // of a dependent branch per element
            float pv[ILDS_E], pq[ILDS_E], uq[ILDS_E];
#pragma unroll
            for (int i = 0; i < ILDS_E; ++i) {
                const int go = qoff[i] >= 0 ? goff[i] : 0, qo = qoff[i] >= 0 ? qoff[i] : 0;
                pv[i] = prep[go];
                pq[i] = poolp[qo];
                uq[i] = upp[qo];
            }
#pragma unroll
            for (int i = 0; i < ILDS_E; ++i) v[i] = (qoff[i] >= 0 && pv[i] == pq[i]) ? uq[i] : 0.f;
        } else {
            const float* src = c < p.C1 ? p.x1 + ((size_t)b * p.C1 + c) * HW
                                        : p.x2 + ((size_t)b * p.C2 + (c - p.C1)) * HW;
            // (unconditional loads at clamped offsets, masked afterwards -- as above: one batch in flight)
            float pv[ILDS_E];
#pragma unroll
            for (int i = 0; i < ILDS_E; ++i) pv[i] = src[goff[i] >= 0 ? goff[i] : 0];
#pragma unroll
            for (int i = 0; i < ILDS_E; ++i) v[i] = goff[i] >= 0 ? pv[i] : 0.f;
        }
    };

    fetch(c0);
    for (int cc = 0; cc < ILDS_CH; ++cc) {
        const int c = c0 + cc;   // Kc is a multiple of 16: always a real (possibly zero-weight) channel
        float* L = Ls[cc & 1];
#pragma unroll
        for (int i = 0; i < ILDS_E; ++i)
            if (i * NT + tid < NE) L[i * NT + tid] = v[i];
        if (cc + 1 < ILDS_CH) fetch(c + 1);
        __syncthreads();
        if (tvalid) {
            float d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float2 lo = *reinterpret_cast<const float2*>(L + lbase + i * NC);
                const float2 hi = *reinterpret_cast<const float2*>(L + lbase + i * NC + 2);
                d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
            }
            float e[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // B^T d
                e[0][j] = d[0][j] - d[2][j];
                e[1][j] = d[1][j] + d[2][j];
                e[2][j] = d[2][j] - d[1][j];
                e[3][j] = d[1][j] - d[3][j];
            }
            float* vo = p.V + (size_t)c * p.Tpad + t;
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // (B^T d) B
                vo[(size_t)(i * 4 + 0) * xis] = e[i][0] - e[i][2];
                vo[(size_t)(i * 4 + 1) * xis] = e[i][1] + e[i][2];
                vo[(size_t)(i * 4 + 2) * xis] = e[i][2] - e[i][1];
                vo[(size_t)(i * 4 + 3) * xis] = e[i][1] - e[i][3];
            }
        }
    }
}

// ---- 2. the 16 GEMMs ---------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int NBUF, int BK = 16>
__global__ __launch_bounds__(WM * WN * 64, 2) void wino_gemm_kernel(const WinoParams p) {
    constexpr int NCH = BK / 2;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int AV = BK * BM / 4, BV = BK * BN / 4;  // float4 per operand tile
    constexpr int NA = NW == 8 ? 256 : NT;             // threads staging A / B
    constexpr int APT = AV / NA, BPT = BV / 256;
    static_assert((NW == 4 || NW == 8) && AV % NA == 0 && BV % 256 == 0, "tile config");

    __shared__ __attribute__((aligned(16))) float As[NBUF][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[NBUF][BK][BN];

    // XCD-aware order: each XCD (bid % 8) walks a contiguous run of (xi, tile) work, inside a
    // run groups of 8 pixel-tiles x all channel-tiles share U and V rows in that XCD's L2
    const int per_xi = p.n_ttiles * p.n_mtiles;
    int xi, tt, mt;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb / 8, r = nb % 8, xcd = bid % 8, l = bid / 8;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + l;
        xi = v / per_xi;
        const int w = v - xi * per_xi;
        constexpr int GP = 8;
        const int gsize = GP * p.n_mtiles;
        const int g = w / gsize, rr = w % gsize;
        const int gp = min(GP, p.n_ttiles - g * GP);
        tt = g * GP + rr % gp;
        mt = rr / gp;
    }
    const int m0 = mt * BM, t0 = tt * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    const __amdgpu_buffer_rsrc_t arsrc = mk_rsrc(p.U + (size_t)xi * p.Kc * p.Mpad, p.Kc * p.Mpad * 4);
    const __amdgpu_buffer_rsrc_t brsrc = mk_rsrc(p.V + (size_t)xi * p.Kc * p.Tpad, p.Kc * p.Tpad * 4);
    const bool bstager = NW == 4 || wave < 4;   // wave-uniform roles
    const bool astager = NW == 4 || wave >= 4;
    const int atid = NW == 8 ? tid - 256 : tid;
    const int awave = NW == 8 ? wave - 4 : wave;

    // global -> LDS directly, 16 bytes per lane: thread-linear == LDS-linear
#define WINO_STAGE(KT, BUF)                                                                        \
    {                                                                                              \
        if (astager)                                                                               \
            static_for<0, APT>([&](auto J) __attribute__((always_inline)) {                        \
                constexpr int j = decltype(J)::value;                                              \
                const int f = j * NA + atid;                                                       \
                const int row = f / (BM / 4), c4 = f % (BM / 4);                                   \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                          \
                    arsrc,                                                                         \
                    (__attribute__((address_space(3))) void*)(&As[BUF][0][0] + (j * NA + awave * 64) * 4), \
                    16, (int)(4u * (unsigned)(((KT) * BK + row) * p.Mpad + m0 + c4 * 4)), 0, 0, 0); \
            });                                                                                    \
        if (bstager)                                                                               \
            static_for<0, BPT>([&](auto J) __attribute__((always_inline)) {                        \
                constexpr int j = decltype(J)::value;                                              \
                const int f = j * 256 + (tid & 255);                                               \
                const int row = f / (BN / 4), c4 = f % (BN / 4);                                   \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                          \
                    brsrc,                                                                         \
                    (__attribute__((address_space(3))) void*)(&Bs[BUF][0][0] + (j * 256 + (wave & 3) * 64) * 4), \
                    16, (int)(4u * (unsigned)(((KT) * BK + row) * p.Tpad + t0 + c4 * 4)), 0, 0, 0); \
            });                                                                                    \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = p.Kc / BK;
    // NBUF-deep ring of LDS tiles: tile kt+NBUF-1 is requested while tile kt is multiplied, so a
    // global -> LDS transfer has NBUF-1 tiles of MFMA work to land.  vmcnt counts this wave's
    // own outstanding DMA ops, in order: leaving one stage's worth in flight means every older
    // stage has landed.
    constexpr int OPS_A = APT, OPS_B = BPT;
#define WINO_WAIT_ONE_STAGE_IN_FLIGHT()                                                            \
    {                                                                                              \
        if constexpr (NW == 8) {                                                                   \
            if (astager) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS_A) : "memory");              \
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS_B) : "memory");                      \
        } else {                                                                                   \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS_A + OPS_B) : "memory");                   \
        }                                                                                          \
    }
    WINO_STAGE(0, 0)
    if constexpr (NBUF == 3) {
        if (nkt > 1) {
            WINO_STAGE(1, 1)
            WINO_WAIT_ONE_STAGE_IN_FLIGHT()
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    int buf = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + NBUF - 1 < nkt;
        int sbuf = buf + NBUF - 1;               // ring slot of the tile requested now
        if (sbuf >= NBUF) sbuf -= NBUF;
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = As[buf][lh][wm * WTM + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = Bs[buf][lh][wn * WTN + j * 32 + l31];
        static_for<0, NCH>([&](auto CH) __attribute__((always_inline)) {
            constexpr int ch = decltype(CH)::value;
            if constexpr (ch + 1 < NCH) {
                const int kk = (ch + 1) * 2 + lh;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(ch + 1) & 1][i] = As[buf][kk][wm * WTM + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[(ch + 1) & 1][j] = Bs[buf][kk][wn * WTN + j * 32 + l31];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ch & 1][i], b[ch & 1][j],
                                                                     acc[i][j], 0, 0, 0);
            if constexpr (ch == 0) {
                if (more) WINO_STAGE(kt + NBUF - 1, sbuf)
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (NBUF == 3 && more) {
            WINO_WAIT_ONE_STAGE_IN_FLIGHT()
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (++buf == NBUF) buf = 0;
    }
#undef WINO_STAGE
#undef WINO_WAIT_ONE_STAGE_IN_FLIGHT

    // M[xi][co][t]: rows of the C/D layout are channels, columns (lane & 31) are tiles
    float* Mx = p.M + (size_t)xi * p.Mpad * p.Tpad;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int t = t0 + wn * WTN + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                Mx[(size_t)co * p.Tpad + t] = acc[i][j][r];
            }
    }
}

// ---- 2+3 fused: one workgroup walks all 16 xi of its (channel, tile) block ------------------------
// M_xi lives in registers only: after the K loop of each xi it is folded into the four output
// accumulators Y_ab += AT[a][i] * AT[b][j] * M_xi (coefficients 0 / +-1, xi = 4i + j), so the
// products never travel to HBM and the output transform costs 4 FMAs per element per xi.
template <int BM, int BN, int WM, int WN, int BK, int MINW>
__global__ __launch_bounds__(WM * WN * 64, MINW) void wino_fused_kernel(const WinoParams p) {
    constexpr int NCH = BK / 2;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NT = WM * WN * 64;
    constexpr int AV = BK * BM / 4, BV = BK * BN / 4;
    constexpr int APT = AV / NT, BPT = BV / NT;
    static_assert(AV % NT == 0 && BV % NT == 0, "tile config");

    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    int tt, mt;
    tile_of_block(blockIdx.x, gridDim.x, p.n_ttiles, p.n_mtiles, tt, mt);
    const int m0 = mt * BM, t0 = tt * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nkt = p.Kc / BK;
    const size_t ustride = (size_t)p.Kc * p.Mpad, vstride = (size_t)p.Kc * p.Tpad;
    const int ubytes = p.Kc * p.Mpad * 4, vbytes = p.Kc * p.Tpad * 4;

    // stage k-tile KT of transform point XI into LDS buffer BUF (global -> LDS, 16 B per lane)
#define WINO_STAGE(XI, KT, BUF)                                                                    \
    {                                                                                              \
        const __amdgpu_buffer_rsrc_t ar = mk_rsrc(p.U + (size_t)(XI) * ustride, ubytes);           \
        const __amdgpu_buffer_rsrc_t br = mk_rsrc(p.V + (size_t)(XI) * vstride, vbytes);           \
        static_for<0, APT>([&](auto J) __attribute__((always_inline)) {                            \
            constexpr int j = decltype(J)::value;                                                  \
            const int f = j * NT + tid;                                                            \
            const int row = f / (BM / 4), c4 = f % (BM / 4);                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                              \
                ar, (__attribute__((address_space(3))) void*)(&As[BUF][0][0] + (j * NT + wave * 64) * 4), \
                16, (int)(4u * (unsigned)(((KT) * BK + row) * p.Mpad + m0 + c4 * 4)), 0, 0, 0);    \
        });                                                                                        \
        static_for<0, BPT>([&](auto J) __attribute__((always_inline)) {                            \
            constexpr int j = decltype(J)::value;                                                  \
            const int f = j * NT + tid;                                                            \
            const int row = f / (BN / 4), c4 = f % (BN / 4);                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                              \
                br, (__attribute__((address_space(3))) void*)(&Bs[BUF][0][0] + (j * NT + wave * 64) * 4), \
                16, (int)(4u * (unsigned)(((KT) * BK + row) * p.Tpad + t0 + c4 * 4)), 0, 0, 0);    \
        });                                                                                        \
    }

    f32x16 acc[TM][TN], Y[4][TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                Y[0][i][j][r] = Y[1][i][j][r] = Y[2][i][j][r] = Y[3][i][j][r] = 0.f;
            }

    WINO_STAGE(0, 0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int xi = 0, kt = 0;              // the k-tile being multiplied
    const int total = 16 * nkt;
    for (int s = 0; s < total; ++s) {
        const int buf = s & 1;
        int nxi = xi, nkt1 = kt + 1;  // the k-tile being staged
        if (nkt1 == nkt) { nkt1 = 0; ++nxi; }
        const bool more = s + 1 < total;
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = As[buf][lh][wm * WTM + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = Bs[buf][lh][wn * WTN + j * 32 + l31];
        static_for<0, NCH>([&](auto CH) __attribute__((always_inline)) {
            constexpr int ch = decltype(CH)::value;
            if constexpr (ch + 1 < NCH) {
                const int kk = (ch + 1) * 2 + lh;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(ch + 1) & 1][i] = As[buf][kk][wm * WTM + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[(ch + 1) & 1][j] = Bs[buf][kk][wn * WTN + j * 32 + l31];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ch & 1][i], b[ch & 1][j],
                                                                     acc[i][j], 0, 0, 0);
            if constexpr (ch == 0) {
                if (more) WINO_STAGE(nxi, nkt1, buf ^ 1)
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (nkt1 == 0) {
            // M_xi complete: Y_ab += AT[a][xi/4] * AT[b][xi%4] * M_xi,  AT = [1 1 1 0; 0 1 -1 -1]
            const int wi = xi >> 2, wj = xi & 3;
            const float r0 = wi < 3 ? 1.f : 0.f, r1 = wi == 0 ? 0.f : (wi == 1 ? 1.f : -1.f);
            const float c0 = wj < 3 ? 1.f : 0.f, c1 = wj == 0 ? 0.f : (wj == 1 ? 1.f : -1.f);
            const float k00 = r0 * c0, k01 = r0 * c1, k10 = r1 * c0, k11 = r1 * c1;
            // coefficients are 0 / +-1 and wave-uniform: skip the accumulators xi does not reach
#define WINO_FOLD(Q, KQ)                                                                            \
            if ((KQ) != 0.f) {                                                                      \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                      \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                      \
                _Pragma("unroll") for (int r = 0; r < 16; ++r)                                      \
                    Y[Q][i][j][r] = fmaf((KQ), acc[i][j][r], Y[Q][i][j][r]);                        \
            }
            WINO_FOLD(0, k00)
            WINO_FOLD(1, k01)
            WINO_FOLD(2, k10)
            WINO_FOLD(3, k11)
#undef WINO_FOLD
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        xi = nxi;
        kt = nkt1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef WINO_STAGE

    // epilogue: rows of the C/D layout are channels, columns (lane & 31) are tiles.  A tile's two
    // pixels of a row go out as ONE 8-byte store (lanes are consecutive tiles: 256 contiguous bytes
    // per 32 lanes) wherever both columns are inside the window; the channel offsets are formed once
    // per (i, r) from 32-bit products.
    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
    const int ntt = p.nty * p.ntx;
    const size_t OPL = (size_t)p.out_H * p.out_W, APL = (size_t)p.AH * p.AW;
    // The tile's BM bias values go through LDS (free after the last k-tile's barrier): read from global per
    // (i, r) each of them is a vector load that cannot move above the stores in front of it -- 16 TM TN
    // dependent memory round trips at the end of every workgroup; LDS reads are not held back by global stores.
    float* Lbias = &As[0][0][0];
    for (int f = tid; f < BM; f += NT) Lbias[f] = (p.bias && m0 + f < p.Cout) ? p.bias[m0 + f] : 0.f;
    __syncthreads();
    float bias_r[TM][16];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) bias_r[i][r] = Lbias[wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int t = t0 + wn * WTN + j * 32 + l31;
        if (t >= p.T) continue;
        const int b = t / ntt;
        const int rr = t - b * ntt;
        const int tyl = rr / p.ntx, txl = rr - tyl * p.ntx;
        const int wy = p.ty0 + 2 * tyl - p.oy0, wx = p.tx0 + 2 * txl - p.ox0;
        const bool rok[2] = {(unsigned)wy < (unsigned)p.OH, (unsigned)(wy + 1) < (unsigned)p.OH};
        const bool cok[2] = {(unsigned)wx < (unsigned)p.OW, (unsigned)(wx + 1) < (unsigned)p.OW};
        const bool pair = cok[0] && cok[1];
        float* ob = p.out + ((size_t)b * p.out_ctot + p.out_c0) * OPL +
                    (ptrdiff_t)(p.out_y0 + wy) * p.out_W + p.out_x0 + wx;
        const float* ab = p.add ? p.add + (size_t)b * p.Cout * APL +
                                      (ptrdiff_t)(p.ay0 + wy) * p.AW + p.ax0 + wx
                                : nullptr;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co >= p.Cout) continue;
                const float bias = bias_r[i][r];
                float* oc = ob + (size_t)co * OPL;
                const float* ac = ab ? ab + (size_t)co * APL : nullptr;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    if (!rok[a]) continue;
                    float v0 = Y[2 * a][i][j][r] + bias, v1 = Y[2 * a + 1][i][j][r] + bias;
                    if (pair) {
                        if (ac) {
                            const f32x2u a2 = *reinterpret_cast<const f32x2u*>(ac + a * p.AW);
                            v0 += a2[0];
                            v1 += a2[1];
                        }
                        if (p.relu) {
                            v0 = fmaxf(v0, 0.f);
                            v1 = fmaxf(v1, 0.f);
                        }
                        *reinterpret_cast<f32x2u*>(oc + (size_t)a * p.out_W) = f32x2u{v0, v1};
                    } else {
                        if (cok[0]) {
                            if (ac) v0 += ac[a * p.AW];
                            if (p.relu) v0 = fmaxf(v0, 0.f);
                            oc[(size_t)a * p.out_W] = v0;
                        }
                        if (cok[1]) {
                            if (ac) v1 += ac[a * p.AW + 1];
                            if (p.relu) v1 = fmaxf(v1, 0.f);
                            oc[(size_t)a * p.out_W + 1] = v1;
                        }
                    }
                }
            }
    }
}

// ---- 3. output transform + epilogue ------------------------------------------------------------
constexpr int OCH = 4;
__global__ __launch_bounds__(256) void wino_output_kernel(const WinoParams p) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= p.T) return;
    const int ntt = p.nty * p.ntx;
    const int b = t / ntt;
    const int r = t - b * ntt;
    const int tyl = r / p.ntx, txl = r - tyl * p.ntx;
    const int wy = p.ty0 + 2 * tyl - p.oy0, wx = p.tx0 + 2 * txl - p.ox0;  // window coords
    const bool okr[2] = {(unsigned)wy < (unsigned)p.OH, (unsigned)(wy + 1) < (unsigned)p.OH};
    const bool okc[2] = {(unsigned)wx < (unsigned)p.OW, (unsigned)(wx + 1) < (unsigned)p.OW};
    const size_t xis = (size_t)p.Mpad * p.Tpad;
    const size_t OPL = (size_t)p.out_H * p.out_W, APL = (size_t)p.AH * p.AW;
    const int co0 = blockIdx.y * OCH;
    // All loads of the thread's OCH channels first (M, skip-add; channel index clamped instead of
    // branching), then the stores: `out` may alias `M` / `add` as far as the compiler knows, so
    // loads placed after a store could not be moved up and every channel would be a round trip.
    float mv[OCH][16], av[OCH][4];
#pragma unroll
    for (int cc = 0; cc < OCH; ++cc) {
        const int co = min(co0 + cc, p.Cout - 1);
        const float* m = p.M + (size_t)co * p.Tpad + t;
#pragma unroll
        for (int x = 0; x < 16; ++x) mv[cc][x] = m[(size_t)x * xis];
        if (p.add) {
            const float* ad = p.add + ((size_t)b * p.Cout + co) * APL;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool ok = okr[i] && okc[j];
                    const ptrdiff_t idx = (ptrdiff_t)(p.ay0 + wy + i) * p.AW + p.ax0 + wx + j;
                    const float a = ad[ok ? idx : 0];
                    av[cc][i * 2 + j] = ok ? a : 0.f;
                }
        }
    }
#pragma unroll
    for (int cc = 0; cc < OCH; ++cc) {
        const int co = co0 + cc;
        if (co >= p.Cout) break;
        float s2[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // A^T m
            const float m0 = mv[cc][j], m1 = mv[cc][4 + j], m2 = mv[cc][8 + j], m3 = mv[cc][12 + j];
            s2[0][j] = m0 + m1 + m2;
            s2[1][j] = m1 - m2 - m3;
        }
        const float bias = p.bias ? p.bias[co] : 0.f;
        float* o = p.out + ((size_t)b * p.out_ctot + p.out_c0 + co) * OPL +
                   (ptrdiff_t)(p.out_y0 + wy) * p.out_W + p.out_x0 + wx;
#pragma unroll
        for (int i = 0; i < 2; ++i) {  // (A^T m) A
            float y[2] = {s2[i][0] + s2[i][1] + s2[i][2], s2[i][1] - s2[i][2] - s2[i][3]};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (okr[i] && okc[j]) {
                    float v = y[j] + bias;
                    if (p.add) v += av[cc][i * 2 + j];
                    if (p.relu) v = fmaxf(v, 0.f);
                    o[(ptrdiff_t)i * p.out_W + j] = v;
                }
            }
        }
    }
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// geometry shared by the sizing queries and the launcher
struct WinoGeom {
    int Kc, Mpad, bm, ty0, tx0, nty, ntx, T, Tpad;
};

int wino_geom(const iiseg_conv_desc* d, WinoGeom& g) {
    if (!d) return IISEG_ERR_NULL;
    if (d->KH != 3 || d->KW != 3 || d->dil != 1 || (d->flags & IISEG_CONV_TRANSPOSED2))
        return IISEG_ERR_UNSUPPORTED;
    if ((d->flags & IISEG_CONV_UNPOOL) && d->C2 != 0) return IISEG_ERR_UNSUPPORTED;
    if (d->B <= 0 || d->C1 <= 0 || d->C2 < 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 ||
        d->pad < 0 || d->OH <= 0 || d->OW <= 0 || d->oy0 < 0 || d->ox0 < 0)
        return IISEG_ERR_SHAPE;
    const int fullH = d->H + 2 * d->pad - 2, fullW = d->W + 2 * d->pad - 2;
    if (d->oy0 + d->OH > fullH || d->ox0 + d->OW > fullW) return IISEG_ERR_SHAPE;
    if ((d->C1 + d->C2) % 16) return IISEG_ERR_UNSUPPORTED;
    g.Kc = d->C1 + d->C2;
    g.bm = d->Cout > 128 ? 256 : (d->Cout > 64 ? 128 : 64);
    g.Mpad = round_up(d->Cout, g.bm);
    // tiles cover output rows r, r+1 with r = tile_y0 (mod 2): the first one is the last such row
    // at or before the window origin
    if ((d->tile_y0 | d->tile_x0) & ~1) return IISEG_ERR_SHAPE;
    g.ty0 = d->oy0 - ((d->oy0 - d->tile_y0) & 1);
    g.tx0 = d->ox0 - ((d->ox0 - d->tile_x0) & 1);
    g.nty = (d->oy0 + d->OH - g.ty0 + 1) >> 1;
    g.ntx = (d->ox0 + d->OW - g.tx0 + 1) >> 1;
    const int64_t T = (int64_t)d->B * g.nty * g.ntx;
    const int64_t Tpad = (T + 127) / 128 * 128;
    // buffer descriptors address one xi-slice of U / V with 32-bit byte offsets
    if (Tpad * g.Kc * 4 >= (int64_t)1 << 31 || (int64_t)g.Kc * g.Mpad * 4 >= (int64_t)1 << 31)
        return IISEG_ERR_UNSUPPORTED;
    g.T = (int)T;
    g.Tpad = (int)Tpad;
    return IISEG_OK;
}

void launch_wino_input(hipStream_t s, const WinoParams& p, bool unpool) {
    static const int lds = getenv("IISEG_WINO_INPUT_LDS") ? atoi(getenv("IISEG_WINO_INPUT_LDS")) : 1;
    const int ntt = p.nty * p.ntx;
    // tiles per workgroup (one image per workgroup): the size that leaves the fewest idle lanes
    int nt = 0;
    double best = 0.0;
    for (int cand = 256; cand >= 128; cand >>= 1) {
        // rows of the tile grid a chunk can span, and the staged region they need
        const int span = (cand - 1 + p.ntx - 1) / p.ntx + 1;
        const int rows = span < p.nty ? span : p.nty;
        if ((2 * rows + 2) * (2 * p.ntx + 2) > ILDS_E * cand) continue;
        const double util = (double)ntt / (((ntt + cand - 1) / cand) * cand);
        if (util > best + 0.02) { best = util; nt = cand; }
    }
    if (lds && nt && best >= 0.8 && p.Kc % ILDS_CH == 0) {
        const int chunks = (ntt + nt - 1) / nt;
        const dim3 g2(p.B * chunks, p.Kc / ILDS_CH);
#define WINO_ILDS(U, N) IISEG_LAUNCH((wino_input_lds_kernel<U, N>), g2, dim3(N), 0, s, p, chunks)
        if (unpool) { if (nt == 256) WINO_ILDS(true, 256); else if (nt == 128) WINO_ILDS(true, 128); else WINO_ILDS(true, 64); }
        else { if (nt == 256) WINO_ILDS(false, 256); else if (nt == 128) WINO_ILDS(false, 128); else WINO_ILDS(false, 64); }
#undef WINO_ILDS
        return;
    }
    const int nc = unpool ? ICH / 2 : ICH;   // channels per thread
    const dim3 grid((p.T + 255) / 256, (p.Kc + nc - 1) / nc), block(256);
    if (!unpool) {
        IISEG_LAUNCH((wino_input_kernel<false, 0, 0>), grid, block, 0, s, p);
        return;
    }
    const int py = (p.ty0 - p.pad) & 1, px = (p.tx0 - p.pad) & 1;  // patch-origin parity
    if (py && px) IISEG_LAUNCH((wino_input_kernel<true, 1, 1>), grid, block, 0, s, p);
    else if (py) IISEG_LAUNCH((wino_input_kernel<true, 1, 0>), grid, block, 0, s, p);
    else if (px) IISEG_LAUNCH((wino_input_kernel<true, 0, 1>), grid, block, 0, s, p);
    else IISEG_LAUNCH((wino_input_kernel<true, 0, 0>), grid, block, 0, s, p);
}

}  // namespace

// Output transform + epilogue for products M[16][Mpad][Tpad] that another GEMM kernel wrote (the
// bf16-operand path of conv_wino_bf16.hip): same kernel, tile geometry recomputed from `d`.
int iiseg_wino_output_launch(hipStream_t s, const iiseg_conv_desc* d, const float* M, int Mpad,
                             int Tpad, const float* bias, const float* add, float* out) {
    WinoParams p = {};
    p.M = const_cast<float*>(M);
    p.bias = bias; p.add = add; p.out = out;
    p.B = d->B; p.Cout = d->Cout; p.pad = d->pad;
    p.oy0 = d->oy0; p.ox0 = d->ox0; p.OH = d->OH; p.OW = d->OW;
    p.ty0 = d->oy0 - ((d->oy0 - d->tile_y0) & 1);
    p.tx0 = d->ox0 - ((d->ox0 - d->tile_x0) & 1);
    p.nty = (d->oy0 + d->OH - p.ty0 + 1) >> 1;
    p.ntx = (d->ox0 + d->OW - p.tx0 + 1) >> 1;
    p.T = d->B * p.nty * p.ntx;
    p.Tpad = Tpad; p.Mpad = Mpad;
    p.AH = d->AH; p.AW = d->AW; p.ay0 = d->ay0; p.ax0 = d->ax0;
    p.relu = (d->flags & IISEG_CONV_RELU) ? 1 : 0;
    p.out_ctot = d->out_ctot > 0 ? d->out_ctot : d->Cout;
    p.out_c0 = d->out_ctot > 0 ? d->out_c0 : 0;
    p.out_H = d->out_H > 0 ? d->out_H : d->OH;
    p.out_W = d->out_H > 0 ? d->out_W : d->OW;
    p.out_y0 = d->out_H > 0 ? d->out_y0 : 0;
    p.out_x0 = d->out_H > 0 ? d->out_x0 : 0;
    IISEG_LAUNCH(wino_output_kernel, dim3((p.T + 255) / 256, (d->Cout + OCH - 1) / OCH),
                       dim3(256), 0, s, p);
    return iiseg_check_launch();
}

extern "C" int iiseg_conv_wino_supported(const iiseg_conv_desc* d) {
    WinoGeom g;
    return wino_geom(d, g) == IISEG_OK ? 1 : 0;
}

extern "C" int64_t iiseg_conv_wino_weight_elems(const iiseg_conv_desc* d) {
    WinoGeom g;
    if (wino_geom(d, g) != IISEG_OK) return 0;
    return (int64_t)16 * g.Kc * g.Mpad;
}

extern "C" int64_t iiseg_conv_wino_workspace_elems(const iiseg_conv_desc* d) {
    WinoGeom g;
    if (wino_geom(d, g) != IISEG_OK) return 0;
    return (int64_t)16 * g.Tpad * ((int64_t)g.Kc + g.Mpad);
}

extern "C" int iiseg_conv_wino_pack_f32(void* stream, const iiseg_conv_desc* d, const float* w,
                                        int64_t stride_o, int64_t stride_c, float* U) {
    WinoGeom g;
    const int st = wino_geom(d, g);
    if (st) return st;
    if (!w || !U) return IISEG_ERR_NULL;
    const int64_t n = (int64_t)g.Kc * g.Mpad;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    IISEG_LAUNCH(wino_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w,
                       stride_o, stride_c, U, d->C1 + d->C2, d->Cout, g.Kc, g.Mpad);
    return iiseg_check_launch();
}

extern "C" int iiseg_conv_wino_f32(void* stream, const iiseg_conv_desc* d, const float* x1,
                                   const float* x2, const float* pre, const float* pooled,
                                   const float* U, const float* bias,
                                   const float* add, float* workspace, float* out,
                                   uint32_t stages) {
    WinoGeom g;
    const int st = wino_geom(d, g);
    if (st) return st;
    if (!x1 || !U || !workspace || !out) return IISEG_ERR_NULL;
    if (d->C2 > 0 && !x2) return IISEG_ERR_NULL;
    const bool unpool = (d->flags & IISEG_CONV_UNPOOL) != 0;
    if (unpool && (!pre || !pooled)) return IISEG_ERR_NULL;
    if (((uintptr_t)U & 15) || ((uintptr_t)workspace & 15)) return IISEG_ERR_ALIGN;
    WinoParams p;
    p.x1 = x1;
    p.x2 = x2;
    p.pre = pre;
    p.pooled = pooled;
    p.h2 = d->H / 2;
    p.w2 = d->W / 2;
    p.U = U;
    p.bias = bias;
    p.add = add;
    p.V = workspace;
    p.M = workspace + (size_t)16 * g.Kc * g.Tpad;
    p.out = out;
    p.B = d->B; p.C1 = d->C1; p.C2 = d->C2; p.H = d->H; p.W = d->W;
    p.Cout = d->Cout; p.pad = d->pad;
    p.oy0 = d->oy0; p.ox0 = d->ox0; p.OH = d->OH; p.OW = d->OW;
    p.ty0 = g.ty0; p.tx0 = g.tx0; p.nty = g.nty; p.ntx = g.ntx;
    p.T = g.T; p.Tpad = g.Tpad; p.Kc = g.Kc; p.Mpad = g.Mpad;
    p.AH = d->AH; p.AW = d->AW; p.ay0 = d->ay0; p.ax0 = d->ax0;
    if (add && (d->ay0 < 0 || d->ax0 < 0 || d->ay0 + d->OH > d->AH || d->ax0 + d->OW > d->AW))
        return IISEG_ERR_SHAPE;
    p.relu = (d->flags & IISEG_CONV_RELU) ? 1 : 0;
    p.out_ctot = d->out_ctot > 0 ? d->out_ctot : d->Cout;
    p.out_c0 = d->out_ctot > 0 ? d->out_c0 : 0;
    if (p.out_c0 < 0 || p.out_c0 + d->Cout > p.out_ctot) return IISEG_ERR_SHAPE;
    p.out_H = d->out_H > 0 ? d->out_H : d->OH;
    p.out_W = d->out_H > 0 ? d->out_W : d->OW;
    p.out_y0 = d->out_H > 0 ? d->out_y0 : 0;
    p.out_x0 = d->out_H > 0 ? d->out_x0 : 0;
    if (p.out_y0 < 0 || p.out_x0 < 0 || p.out_y0 + d->OH > p.out_H || p.out_x0 + d->OW > p.out_W)
        return IISEG_ERR_SHAPE;
    p.n_ttiles = g.Tpad / 128;
    p.n_mtiles = g.Mpad / g.bm;
    hipStream_t s = (hipStream_t)stream;
    const int tb = (g.T + 255) / 256;
    if (stages & IISEG_WINO_FUSED) {
        if (g.Kc % 32) return IISEG_ERR_UNSUPPORTED;  // the fused kernel's k-tile is 32 channels
        // input transform, then GEMMs + output transform in one kernel (M stays in registers)
        if (stages & IISEG_WINO_INPUT)
            launch_wino_input(s, p, unpool);
        if (stages & IISEG_WINO_GEMM) {
            if (g.bm == 64) {   // narrow layers (Cout <= 64): 64-channel tiles, 4 waves of 64 x 32
                p.n_mtiles = g.Mpad / 64;
                IISEG_LAUNCH((wino_fused_kernel<64, 128, 1, 4, 32, 2>),
                                   dim3(p.n_ttiles * p.n_mtiles), dim3(256), 0, s, p);
                return iiseg_check_launch();
            }
            p.n_mtiles = g.Mpad / 128;
            // short K loops (<= 128 channels): two independent 4-wave workgroups per CU hide the
            // per-xi fold better than one 8-wave workgroup (scripts/bench_wino.py)
            // ... and, for any K, when the 8-wave kernel's few long workgroups would leave CUs idle
            // in the last round (work per CU is quantised in whole workgroups)
            const int w0 = p.n_ttiles * p.n_mtiles, w2 = (g.Tpad / 64) * p.n_mtiles;
            const double cost0 = (w0 + 255) / 256, cost2 = 0.5 * 1.03 * ((w2 + 255) / 256);
            // IISEG_WINO_FUSED_BK=64: 64-channel stages (same bits); measured slower (conv2_2 1.38 ->
            // 1.71 ms: one workgroup per CU instead of two), 32 stays the default
            static const int bk64 = getenv("IISEG_WINO_FUSED_BK") ? atoi(getenv("IISEG_WINO_FUSED_BK")) == 64 : 0;
            const bool big = bk64 && g.Kc % 64 == 0;
            if (g.Kc <= 128 || cost2 < cost0) {
                p.n_ttiles = g.Tpad / 64;
                if (big)
                    IISEG_LAUNCH((wino_fused_kernel<128, 64, 2, 2, 64, 2>),
                                       dim3(p.n_ttiles * p.n_mtiles), dim3(256), 0, s, p);
                else
                IISEG_LAUNCH((wino_fused_kernel<128, 64, 2, 2, 32, 2>),
                                   dim3(p.n_ttiles * p.n_mtiles), dim3(256), 0, s, p);
            } else if (big)
                IISEG_LAUNCH((wino_fused_kernel<128, 128, 2, 4, 64, 2>),
                                   dim3(p.n_ttiles * p.n_mtiles), dim3(512), 0, s, p);
            else
            IISEG_LAUNCH((wino_fused_kernel<128, 128, 2, 4, 32, 2>),
                               dim3(p.n_ttiles * p.n_mtiles), dim3(512), 0, s, p);
        }
        return iiseg_check_launch();
    }
    if (stages & IISEG_WINO_INPUT)
        launch_wino_input(s, p, unpool);
    const int grid = 16 * p.n_ttiles * p.n_mtiles;
    if (stages & IISEG_WINO_GEMM) {
        static const int nbuf = getenv("IISEG_WINO_NBUF") ? atoi(getenv("IISEG_WINO_NBUF")) : 2;
        int bm = g.bm;
        if (bm == 64) {
            IISEG_LAUNCH((wino_gemm_kernel<64, 128, 1, 4, 2>), dim3(grid), dim3(256), 0, s, p);
            bm = 0;
        }
        if (bm == 256) {
            // few, long 256x128 workgroups quantise badly over 256 CUs: fall back to 128x128
            // tiles (about 6 % less MFMA-efficient) when that fills the last round better
            const int w256 = grid, w128 = 2 * grid;
            if (0.5 * 1.06 * ((w128 + 255) / 256) < (double)((w256 + 255) / 256)) bm = 128;
        }
        const int grid2 = bm ? 16 * p.n_ttiles * (g.Mpad / bm) : 0;
        if (bm) p.n_mtiles = g.Mpad / bm;
        // IISEG_WINO_GEMM_BK=32: 32-channel k-tiles (same k order, same bits).  Tried because the bf16
        // kernels are paced per LDS-DMA stage; the fp32 GEMM is not -- measured 2.5 %
        // SLOWER (conv6_1 1.133 -> 1.198 ms, A/B on one device), so 16 stays the default.
        static const int bk32 = getenv("IISEG_WINO_GEMM_BK") ? atoi(getenv("IISEG_WINO_GEMM_BK")) == 32 : 0;
        if (bm && bk32 && g.Kc % 32 == 0 && nbuf != 3) {
            if (bm == 256)
                IISEG_LAUNCH((wino_gemm_kernel<256, 128, 4, 2, 2, 32>), dim3(grid), dim3(512), 0, s, p);
            else
                IISEG_LAUNCH((wino_gemm_kernel<128, 128, 2, 2, 2, 32>), dim3(grid2), dim3(256), 0, s, p);
        } else
        if (bm == 0) {
        } else if (bm == 256 && nbuf == 3)
            IISEG_LAUNCH((wino_gemm_kernel<256, 128, 4, 2, 3>), dim3(grid), dim3(512), 0, s, p);
        else if (bm == 256)
            IISEG_LAUNCH((wino_gemm_kernel<256, 128, 4, 2, 2>), dim3(grid), dim3(512), 0, s, p);
        else if (nbuf == 3)
            IISEG_LAUNCH((wino_gemm_kernel<128, 128, 2, 2, 3>), dim3(grid2), dim3(256), 0, s, p);
        else
            IISEG_LAUNCH((wino_gemm_kernel<128, 128, 2, 2, 2>), dim3(grid2), dim3(256), 0, s, p);
    }
    if (stages & IISEG_WINO_OUTPUT)
        IISEG_LAUNCH(wino_output_kernel, dim3(tb, (d->Cout + OCH - 1) / OCH), dim3(256), 0, s,
                           p);
    return iiseg_check_launch();
}

// ================================================================================================
// im2col + split-K GEMM for 'valid' KxK convolutions with a small output map (FCN-8's fc6: 7x7 over
// 13x13 -> 7x7, K = 25088): the table-driven gather of conv_igemm.hip tops out at 78 TFLOP/s there,
// the gather-free GEMM kernel above runs at 125-134.  The S split-K slices ride on the kernel's
// batch index (slice s = rows [s*Kc, (s+1)*Kc) of Wp and of the im2col matrix, contiguous in both),
// so that S * tiles fills the 256 CUs evenly; the output kernel sums the S partial products in a
// fixed order, adds bias, applies ReLU and writes NCHW.
// ================================================================================================
namespace {

struct GemmConvGeom {
    int K, Kpad, Mpad, bm, T, Tpad, S, Kc;
};

__global__ __launch_bounds__(256) void gemm_im2col_kernel(const float* __restrict__ x,
                                                          float* __restrict__ V, int B, int C, int H,
                                                          int W, int KH, int KW, int OH, int OW, int K,
                                                          int Kpad, int T, int Tpad) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int OHW = OH * OW;
    const int b = t / OHW, r = t - b * OHW;
    const int oy = r / OW, ox = r - oy * OW;
    const int KK = KH * KW;
    const float* xb = x + (size_t)b * C * H * W + (size_t)oy * W + ox;
    for (int k = blockIdx.y; k < Kpad; k += gridDim.y) {
        float v = 0.f;
        if (k < K) {
            const int c = k / KK, tap = k - c * KK;
            const int ky = tap / KW, kx = tap - ky * KW;
            v = xb[(size_t)c * H * W + ky * W + kx];
        }
        V[(size_t)k * Tpad + t] = v;
    }
}

__global__ __launch_bounds__(256) void gemm_output_kernel(const float* __restrict__ M,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ out, int Cout, int OHW,
                                                          int T, int Tpad, int Mpad, int S, int relu) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int b = t / OHW, px = t - b * OHW;
    const size_t ss = (size_t)Mpad * Tpad;
    for (int co = blockIdx.y; co < Cout; co += gridDim.y) {
        const float* m = M + (size_t)co * Tpad + t;
        float v = m[0];
        for (int s = 1; s < S; ++s) v += m[(size_t)s * ss];   // fixed order: deterministic
        if (bias) v += bias[co];
        if (relu) v = fmaxf(v, 0.f);
        out[((size_t)b * Cout + co) * OHW + px] = v;
    }
}

int gemm_conv_geom(const iiseg_conv_desc* d, GemmConvGeom& g) {
    if (!d) return IISEG_ERR_NULL;
    if (d->pad != 0 || d->dil != 1 || d->C2 != 0 ||
        (d->flags & (IISEG_CONV_UNPOOL | IISEG_CONV_TRANSPOSED2)))
        return IISEG_ERR_UNSUPPORTED;
    const int fullH = d->H - d->KH + 1, fullW = d->W - d->KW + 1;
    if (fullH <= 0 || fullW <= 0) return IISEG_ERR_SHAPE;
    // full, dense output only
    if (d->oy0 != 0 || d->ox0 != 0 || d->OH != fullH || d->OW != fullW || d->out_ctot != 0 ||
        d->out_H != 0)
        return IISEG_ERR_UNSUPPORTED;
    g.K = d->C1 * d->KH * d->KW;
    g.Kpad = d->Kpad;
    g.Mpad = d->Mpad;
    if (g.Kpad < g.K || g.Kpad % 16 || g.Mpad % 128 || g.Mpad < d->Cout) return IISEG_ERR_UNSUPPORTED;
    g.bm = g.Mpad % 256 == 0 ? 256 : 128;
    const int64_t T = (int64_t)d->B * fullH * fullW;
    const int64_t Tpad = (T + 127) / 128 * 128;
    if (Tpad * g.Kpad * 4 >= (int64_t)1 << 31 || (int64_t)g.Kpad * g.Mpad * 4 >= (int64_t)1 << 31)
        return IISEG_ERR_UNSUPPORTED;
    g.T = (int)T;
    g.Tpad = (int)Tpad;
    // split-K: the divisor S of Kpad/16 (<= 16) minimising  GEMM time (rounds of 256 CUs at the
    // measured 131 TFLOP/s) + the traffic of writing and re-reading the S partial products (5 TB/s).
    // S fixes the association of the K sum, so it must NOT depend on the batch: it is chosen for a
    // nominal 3200 output pixels (64 images x 7x7, the fc6 / fc7 case) whatever the real T is,
    // which keeps an image's result independent of the images it is batched with.
    const int tiles = 25 * (g.Mpad / g.bm), units = g.Kpad / 16;
    const double tile_s = 2.0 * g.Kpad * g.bm * 128 / (131e12 / 256);
    const double red_s = 8.0 * g.Mpad * 3200 / 5e12;
    double best = 1e30;
    g.S = 1;
    for (int S = 1; S <= 16; ++S) {
        if (units % S || units / S < 8) continue;
        const double cost = (double)((S * tiles + 255) / 256) / S * tile_s + S * red_s;
        if (cost < best) { best = cost; g.S = S; }
    }
    g.Kc = g.Kpad / g.S;
    return IISEG_OK;
}

}  // namespace

// bias + ReLU + NCHW store of a GEMM result M[Mpad][Tpad] (S = 1) for the bf16 GEMM path
int iiseg_gemm_output_launch(hipStream_t s, const float* M, const float* bias, float* out, int Cout,
                             int OHW, int T, int Tpad, int Mpad, int relu) {
    const int cy = Cout < 1024 ? Cout : 1024;
    IISEG_LAUNCH(gemm_output_kernel, dim3((T + 255) / 256, cy), dim3(256), 0, s, M, bias, out,
                       Cout, OHW, T, Tpad, Mpad, 1, relu);
    return iiseg_check_launch();
}

extern "C" int iiseg_conv_gemm_supported(const iiseg_conv_desc* d) {
    GemmConvGeom g;
    return gemm_conv_geom(d, g) == IISEG_OK ? 1 : 0;
}

extern "C" int64_t iiseg_conv_gemm_workspace_elems(const iiseg_conv_desc* d) {
    GemmConvGeom g;
    if (gemm_conv_geom(d, g) != IISEG_OK) return 0;
    return (int64_t)g.Tpad * ((int64_t)g.Kpad + (int64_t)g.S * g.Mpad);
}

extern "C" int iiseg_conv_gemm_f32(void* stream, const iiseg_conv_desc* d, const float* x,
                                   const float* wp, const float* bias, float* workspace,
                                   float* out, uint32_t stages) {
    GemmConvGeom g;
    const int st = gemm_conv_geom(d, g);
    if (st) return st;
    if (!x || !wp || !workspace || !out) return IISEG_ERR_NULL;
    if (((uintptr_t)wp & 15) || ((uintptr_t)workspace & 15)) return IISEG_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    float* V = workspace;
    float* M = workspace + (size_t)g.Kpad * g.Tpad;
    const int OH = d->OH, OW = d->OW, tb = (g.T + 255) / 256;
    if (stages & IISEG_WINO_INPUT)
        IISEG_LAUNCH(gemm_im2col_kernel, dim3(tb, g.Kpad < 1024 ? g.Kpad : 1024), dim3(256), 0,
                           s, x, V, d->B, d->C1, d->H, d->W, d->KH, d->KW, OH, OW, g.K, g.Kpad, g.T,
                           g.Tpad);
    WinoParams p = {};
    p.U = wp;
    p.V = V;
    p.M = M;
    p.Kc = g.Kc;
    p.Mpad = g.Mpad;
    p.Tpad = g.Tpad;
    p.T = g.T;
    p.n_ttiles = g.Tpad / 128;
    p.n_mtiles = g.Mpad / g.bm;
    const int grid = g.S * p.n_ttiles * p.n_mtiles;
    if (stages & IISEG_WINO_GEMM) {
        if (g.bm == 256)
            IISEG_LAUNCH((wino_gemm_kernel<256, 128, 4, 2, 2>), dim3(grid), dim3(512), 0, s, p);
        else
            IISEG_LAUNCH((wino_gemm_kernel<128, 128, 2, 2, 2>), dim3(grid), dim3(256), 0, s, p);
    }
    const int cy = d->Cout < 1024 ? d->Cout : 1024;
    if (stages & IISEG_WINO_OUTPUT)
        IISEG_LAUNCH(gemm_output_kernel, dim3(tb, cy), dim3(256), 0, s, M, bias, out, d->Cout,
                           OH * OW, g.T, g.Tpad, g.Mpad, g.S, (d->flags & IISEG_CONV_RELU) ? 1 : 0);
    return iiseg_check_launch();
}
