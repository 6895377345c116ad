// Winograd F(2x2, 3x3) convolution on the bf16 matrix pipe of gfx950 (CDNA4):
// v_mfma_f32_32x32x16_bf16, bf16 operands, fp32 accumulation -- the 16-bit MFMA path of the hot
// path's wide 3x3 layers (BASELINE north_star: ">= 1000 images/s at >= 40 % of fp16 MFMA peak";
// configs[2]: "bf16 with fp32 accumulate").  Activations stay fp32 NCHW in HBM and at the API;
// only the two GEMM operands are rounded to bf16 (round to nearest even):
//   U16[xi][c/8][co][8]  = bf16((G g G^T)[xi])      packed once per layer (computed in double)
//   V16[xi][c/8][t][8]   = bf16((B^T d B)[xi])      input transform, d in fp32 (DePool2D mask and
//                                                   channel concat applied while loading)
// Both are "k8-chunk" images: the 8 channels a lane feeds to one MFMA (k = 8*(lane>>5) + j) are 16
// contiguous bytes, so operand tiles stream global -> LDS with 16-byte LDS-DMA and a fragment is one
// conflict-free ds_read_b128 (lanes 0..31 read 512 contiguous bytes).
// One kernel multiplies AND output-transforms: a workgroup walks the 16 transform points of its
// (channel, tile) block, folds each finished product tile into four fp32 output accumulators
// (Y_ab += AT[a][i] AT[b][j] M_ij, coefficients 0 / +-1) and applies the epilogue (bias, skip-add
// with crop, ReLU, window, placement, channel slice): the products M never reach HBM.
//
// Numerics: statistical parity only (8 significant bits per operand); the fp32 and float64 paths
// are the ones with tolerance claims.  Results are deterministic, and equal
// patches still give bit-equal outputs (same fixed-order sums, same tile anchoring as the fp32
// Winograd path), so the pad-100 border ties of the equality masks stay ties.
// Same Lasagne Conv2DLayer(3x3, stride 1) call sites as conv_wino.hip (models/fcn8.py:41-71,
// models/fcn_down.py:102-104, models/fcn_up.py:83-86).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "iiseg.h"
#include "common.h"
#include "conv_common.h"

using namespace iiseg;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));   // 8-byte store, 4-byte aligned

struct WinoBfParams {
    const float* x1;
    const float* x2;
    const float* pre;     // unpool mode: x1 = up, pre, pooled (DePool2D operands)
    const float* pooled;
    int h2, w2;
    const uint4* U;       // U16 chunks
    const float* bias;
    const float* add;
    uint4* V;             // V16 chunks
    float* out;
    int B, C1, C2, H, W;
    int Cout, pad;
    int oy0, ox0, OH, OW;    // output window in conv-output coordinates
    int ty0, tx0, nty, ntx;  // first tile's output row / column (absolute), tile counts
    int T, Tpad;             // B*nty*ntx, padded to the GEMM pixel tile
    int Kc, Mpad;            // channels padded to the k-tile, output channels padded to the GEMM tile
    int AH, AW, ay0, ax0;
    int relu;
    int out_ctot, out_c0, out_H, out_W, out_y0, out_x0;
    int n_ttiles, n_mtiles;
    int debug;               // timing experiments only (IISEG_BF16_DEBUG): 1 no A DMA, 2 no B DMA, 4 no MFMA
};

constexpr int RSRC_W3 = 0x00027000;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, RSRC_W3);
}

// two floats -> two bf16 (round to nearest even; a NaN stays a NaN: v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// ---- 0. weights: U16[xi][c/8][co][c%8] = bf16((G g G^T)[xi]), g = w[co][c] (cross-correlation) ----
__global__ void wino_weight_bf16_kernel(const float* __restrict__ w, int64_t so, int64_t sc,
                                        __bf16* __restrict__ U, int Cin, int Cout, int Kc, int Mpad) {
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
        const int64_t chunk = ((int64_t)(c >> 3) * Mpad + co) * 8 + (c & 7);
#pragma unroll
        for (int a = 0; a < 4; ++a) {  // (G g) G^T
            const double u[4] = {t[a][0], 0.5 * (t[a][0] + t[a][1] + t[a][2]),
                                 0.5 * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
#pragma unroll
            for (int b = 0; b < 4; ++b) U[(int64_t)(a * 4 + b) * n + chunk] = (__bf16)(float)u[b];
        }
    }
}

// ---- 1. input transform -> V16 -----------------------------------------------------------------
// One thread = one tile x 8 channels (one k8 chunk per transform point); lanes run along tiles, so
// every store instruction writes 64 consecutive 16-byte chunks.  Channels are handled in pairs (two
// fp32 results pack into one dword of the chunk); all loads of a half-group (4 channels) are issued
// before any store.  UNPOOL / PY / PX as in conv_wino.hip: the DePool2D equality mask
// (layers/mylayers.py:88-115) is applied to the fp32 values while the patch is loaded.
template <bool UNPOOL, int PY, int PX>
__global__ __launch_bounds__(256) void wino_input_bf16_kernel(const WinoBfParams p) {
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
    const int Ctot = p.C1 + p.C2;
    const int kc = blockIdx.y;                 // channel group: channels 8*kc .. 8*kc + 7
    const size_t xis = (size_t)(p.Kc >> 3) * p.Tpad;       // chunks per transform point
    uint32_t res[16][4];                       // [xi][channel pair] = the 16 chunks of this thread
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        constexpr int NC_ = 4;
        float pv[NC_][4][4];
        float pq[UNPOOL ? NC_ : 1][NQY][NQX], uq[UNPOOL ? NC_ : 1][NQY][NQX];
        bool creal[NC_];
#pragma unroll
        for (int cc = 0; cc < NC_; ++cc) {
            const int c = kc * 8 + half * NC_ + cc;
            creal[cc] = c < Ctot;              // channels beyond the layer's: zero (padded k-tile)
            const int cs = creal[cc] ? c : 0;
            if constexpr (UNPOOL) {
                const float* prep = p.pre + ((size_t)b * p.C1 + cs) * HW;
                const float* poolp = p.pooled + ((size_t)b * p.C1 + cs) * hw2;
                const float* upp = p.x1 + ((size_t)b * p.C1 + cs) * hw2;
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
                    for (int j = 0; j < 4; ++j)
                        pv[cc][i][j] = prep[(rok[i] && cok[j]) ? rowoff[i] + j : 0];
            } else {
                const float* src = cs < p.C1 ? p.x1 + ((size_t)b * p.C1 + cs) * HW
                                             : p.x2 + ((size_t)b * p.C2 + (cs - p.C1)) * HW;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        pv[cc][i][j] = src[(rok[i] && cok[j]) ? rowoff[i] + j : 0];
            }
        }
#pragma unroll
        for (int pr = 0; pr < NC_ / 2; ++pr) {
            float vv[2][16];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cc = pr * 2 + u;
                float d[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (UNPOOL) {
                            const int qi = (PY + i) >> 1, qj = (PX + j) >> 1;
                            const bool ok = creal[cc] && rok[i] && cok[j] && qrok[qi] && qcok[qj];
                            d[i][j] = (ok && pv[cc][i][j] == pq[cc][qi][qj]) ? uq[cc][qi][qj] : 0.f;
                        } else {
                            d[i][j] = (creal[cc] && rok[i] && cok[j]) ? pv[cc][i][j] : 0.f;
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
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // (B^T d) B
                    vv[u][i * 4 + 0] = e[i][0] - e[i][2];
                    vv[u][i * 4 + 1] = e[i][1] + e[i][2];
                    vv[u][i * 4 + 2] = e[i][2] - e[i][1];
                    vv[u][i * 4 + 3] = e[i][1] - e[i][3];
                }
            }
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) res[xi][half * 2 + pr] = pack_bf16(vv[0][xi], vv[1][xi]);
        }
    }
    uint4* v = p.V + (size_t)kc * p.Tpad + t;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
        v[(size_t)xi * xis] = make_uint4(res[xi][0], res[xi][1], res[xi][2], res[xi][3]);
}


// LDS-staged input transform for maps with >= 128 tiles per image (as conv_wino.hip's
// wino_input_lds_kernel): a workgroup takes NT consecutive tiles of ONE image and the 8 channels of
// one k8 chunk; per channel it stages the input rows those tiles touch (whole rows of the tile grid:
// coalesced loads, ~1.3 loads per tile and channel instead of 16; the DePool2D mask is applied per
// staged element) and every thread reads its 4x4 patch from LDS, transforms it and keeps the 16
// results; after the 8th channel the 16 chunks are stored.
constexpr int IBF_E = 12;   // staged elements per thread and channel
template <bool UNPOOL, int NT>
__global__ __launch_bounds__(NT) void wino_input_lds_bf16_kernel(const WinoBfParams p, const int chunks) {
    constexpr int CAP = IBF_E * NT;
    __shared__ __attribute__((aligned(16))) float Ls[2][CAP];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
    const int ntt = p.nty * p.ntx;
    const int tl0 = chunk * NT, tl = tl0 + tid;
    const bool tvalid = tl < ntt;
    const int row_first = tl0 / p.ntx;
    const int row_last = min(ntt - 1, tl0 + NT - 1) / p.ntx;
    const int NR = 2 * (row_last - row_first + 1) + 2, NC = 2 * p.ntx + 2, NE = NR * NC;
    const int iyb = p.ty0 + 2 * row_first - p.pad, ixb = p.tx0 - p.pad;
    int goff[IBF_E], qoff[UNPOOL ? IBF_E : 1];
#pragma unroll
    for (int i = 0; i < IBF_E; ++i) {
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
    const int Ctot = p.C1 + p.C2;
    const int kc = blockIdx.y;
    float v[IBF_E];

    auto fetch = [&](int c) __attribute__((always_inline)) {
        if (c >= Ctot) {                       // padded k-tile: zero channel
#pragma unroll
            for (int i = 0; i < IBF_E; ++i) v[i] = 0.f;
            return;
        }
        if constexpr (UNPOOL) {
            const float* prep = p.pre + ((size_t)b * p.C1 + c) * HW;
            const float* poolp = p.pooled + ((size_t)b * p.C1 + c) * hw2;
            const float* upp = p.x1 + ((size_t)b * p.C1 + c) * hw2;
            // all three loads of every staged element unconditionally (offsets of elements outside
            // the 2h x 2w region clamped to 0 and masked afterwards): one batch in flight instead
            // of a dependent branch per element
            float pv[IBF_E], pq[IBF_E], uq[IBF_E];
#pragma unroll
            for (int i = 0; i < IBF_E; ++i) {
                const int go = qoff[i] >= 0 ? goff[i] : 0, qo = qoff[i] >= 0 ? qoff[i] : 0;
                pv[i] = prep[go];
                pq[i] = poolp[qo];
                uq[i] = upp[qo];
            }
#pragma unroll
            for (int i = 0; i < IBF_E; ++i) v[i] = (qoff[i] >= 0 && pv[i] == pq[i]) ? uq[i] : 0.f;
        } else {
            const float* src = c < p.C1 ? p.x1 + ((size_t)b * p.C1 + c) * HW
                                        : p.x2 + ((size_t)b * p.C2 + (c - p.C1)) * HW;
            float pv[IBF_E];                   // (unconditional loads at clamped offsets, masked afterwards)
#pragma unroll
            for (int i = 0; i < IBF_E; ++i) pv[i] = src[goff[i] >= 0 ? goff[i] : 0];
#pragma unroll
            for (int i = 0; i < IBF_E; ++i) v[i] = goff[i] >= 0 ? pv[i] : 0.f;
        }
    };

    uint32_t res[16][4];
    float even[16];
    fetch(kc * 8);
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        float* L = Ls[cc & 1];
#pragma unroll
        for (int i = 0; i < IBF_E; ++i)
            if (i * NT + tid < NE) L[i * NT + tid] = v[i];
        if (cc + 1 < 8) fetch(kc * 8 + cc + 1);
        __syncthreads();
        float vv[16];
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
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // (B^T d) B
                vv[i * 4 + 0] = e[i][0] - e[i][2];
                vv[i * 4 + 1] = e[i][1] + e[i][2];
                vv[i * 4 + 2] = e[i][2] - e[i][1];
                vv[i * 4 + 3] = e[i][1] - e[i][3];
            }
        } else {
#pragma unroll
            for (int x = 0; x < 16; ++x) vv[x] = 0.f;
        }
        if (cc & 1) {
#pragma unroll
            for (int x = 0; x < 16; ++x) res[x][cc >> 1] = pack_bf16(even[x], vv[x]);
        } else {
#pragma unroll
            for (int x = 0; x < 16; ++x) even[x] = vv[x];
        }
    }
    if (tvalid) {
        const size_t xis = (size_t)(p.Kc >> 3) * p.Tpad;
        uint4* vo = p.V + (size_t)kc * p.Tpad + (size_t)b * ntt + tl;
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
            vo[(size_t)xi * xis] = make_uint4(res[xi][0], res[xi][1], res[xi][2], res[xi][3]);
    }
}

void launch_wino_input_bf16(hipStream_t s, const WinoBfParams& p, bool unpool) {
    static const int lds = getenv("IISEG_WINO_INPUT_LDS") ? atoi(getenv("IISEG_WINO_INPUT_LDS")) : 1;
    const int ntt = p.nty * p.ntx;
    // tiles per workgroup (one image per workgroup): the size that leaves the fewest idle lanes
    int nt = 0;
    double best = 0.0;
    for (int cand = 256; cand >= 128; cand >>= 1) {
        const int span = (cand - 1 + p.ntx - 1) / p.ntx + 1;
        const int rows = span < p.nty ? span : p.nty;
        if ((2 * rows + 2) * (2 * p.ntx + 2) > IBF_E * cand) continue;
        const double util = (double)ntt / (((ntt + cand - 1) / cand) * cand);
        if (util > best + 0.02) { best = util; nt = cand; }
    }
    // (A/B on one device, DePool2D variant with its three loads per staged element issued as one
    // unconditional batch: up_conv3 0.24 ms staged vs 0.34 ms un-staged)
    static const int lds_unpool = getenv("IISEG_WINO_INPUT_LDS_UNPOOL") ? atoi(getenv("IISEG_WINO_INPUT_LDS_UNPOOL")) : 1;
    if (lds && nt && best >= 0.8 && (!unpool || lds_unpool)) {
        const int chunks = (ntt + nt - 1) / nt;
        const dim3 g2(p.B * chunks, p.Kc / 8);
        if (unpool) {
            if (nt == 256) IISEG_LAUNCH((wino_input_lds_bf16_kernel<true, 256>), g2, dim3(256), 0, s, p, chunks);
            else IISEG_LAUNCH((wino_input_lds_bf16_kernel<true, 128>), g2, dim3(128), 0, s, p, chunks);
        } else {
            if (nt == 256) IISEG_LAUNCH((wino_input_lds_bf16_kernel<false, 256>), g2, dim3(256), 0, s, p, chunks);
            else IISEG_LAUNCH((wino_input_lds_bf16_kernel<false, 128>), g2, dim3(128), 0, s, p, chunks);
        }
        return;
    }
    const dim3 grid((p.T + 255) / 256, p.Kc / 8), block(256);
    if (!unpool) {
        IISEG_LAUNCH((wino_input_bf16_kernel<false, 0, 0>), grid, block, 0, s, p);
        return;
    }
    const int py = (p.ty0 - p.pad) & 1, px = (p.tx0 - p.pad) & 1;  // patch-origin parity
    if (py && px) IISEG_LAUNCH((wino_input_bf16_kernel<true, 1, 1>), grid, block, 0, s, p);
    else if (py) IISEG_LAUNCH((wino_input_bf16_kernel<true, 1, 0>), grid, block, 0, s, p);
    else if (px) IISEG_LAUNCH((wino_input_bf16_kernel<true, 0, 1>), grid, block, 0, s, p);
    else IISEG_LAUNCH((wino_input_bf16_kernel<true, 0, 0>), grid, block, 0, s, p);
}

// ---- 2+3. the 16 GEMMs + output transform + epilogue in one kernel --------------------------------
// As / Bs hold k-tiles of BK channels as BK/8 rows of 16-byte chunks, in an NBUF-deep ring: the
// bf16 matrix pipe needs only ~256 cycles per k-tile and wave, far less than a global -> LDS
// transfer takes to land, so NBUF-1 k-tiles are kept in flight (counted vmcnt, raw s_barrier: a
// __syncthreads() would drain the LDS-DMA queue).  A wave owns TM x TN blocks of 32 channels x 32
// tiles; k-step s of a k-tile reads chunk rows 2s + (lane >> 5).
template <int BM, int BN, int WM, int WN, int BK, int NBUF, int MINW>
__global__ __launch_bounds__(WM * WN * 64, MINW) void wino_fused_bf16_kernel(const WinoBfParams p) {
    constexpr int KR = BK / 8, NS = BK / 16;   // chunk rows per k-tile, MFMA k-steps per k-tile
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NT = WM * WN * 64;
    constexpr int AV = KR * BM, BV = KR * BN;  // chunks per operand tile
    constexpr int APT = AV / NT, BPT = BV / NT;
    constexpr int OPS = APT + BPT;             // LDS-DMA instructions per thread and stage
    static_assert(AV % NT == 0 && BV % NT == 0 && BK % 16 == 0 && NBUF >= 2, "tile config");

    __shared__ __attribute__((aligned(16))) uint4 As[NBUF][KR][BM];
    __shared__ __attribute__((aligned(16))) uint4 Bs[NBUF][KR][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nkt = p.Kc / BK;
    const int kcr = p.Kc >> 3;                             // chunk rows per transform point
    // one descriptor per operand over all 16 points (sizes checked < 4 GB by the host)
    const __amdgpu_buffer_rsrc_t ar = mk_rsrc(p.U, (int)(16u * 16u * (unsigned)kcr * (unsigned)p.Mpad));
    const __amdgpu_buffer_rsrc_t br = mk_rsrc(p.V, (int)(16u * 16u * (unsigned)kcr * (unsigned)p.Tpad));

    // PERSISTENT workgroups: the grid is one residency round (a multiple of 8 blocks, or every
    // tile), and each workgroup walks tiles vb = blockIdx.x, + gridDim.x, ...  All workgroups start
    // together and every tile costs the same, so the workgroups of an XCD stay in step through the
    // (xi, k-tile) sequence: a U16 / V16 slab one of them pulls into that XCD's L2 is a hit for the
    // others.  (Non-persistent, the second and later rounds start out of phase, all 16 points'
    // slabs are live at once, nothing stays in the 4 MB L2 and every operand byte comes from beyond
    // it: measured 73 % of wave time parked on vmcnt at ~5 TB/s.)
    const int ntiles = p.n_ttiles * p.n_mtiles;
  for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
    int tt, mt;
    tile_of_block(vb, ntiles, p.n_ttiles, p.n_mtiles, tt, mt);
    const int m0 = mt * BM, t0 = tt * BN;

    // Staging addresses.  The U16 / V16 images are row-major in (xi, chunk row), so k-tile number S
    // (= xi * nkt + kt) starts at chunk row S * KR of both: the per-thread part of a DMA address
    // (row inside the k-tile, column inside the block) is loop-invariant and lives in VGPRs, the
    // k-tile part is ONE scalar byte offset per operand that advances by a constant per stage --
    // no per-stage vector arithmetic, no division, one buffer descriptor per operand.
    unsigned avo[APT], bvo[BPT];
#pragma unroll
    for (int j = 0; j < APT; ++j) {
        const int f = j * NT + tid;
        avo[j] = 16u * (unsigned)((f / BM) * p.Mpad + m0 + f % BM);
    }
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
        const int f = j * NT + tid;
        bvo[j] = 16u * (unsigned)((f / BN) * p.Tpad + t0 + f % BN);
    }
    unsigned sa = 0, sb = 0;                   // byte offsets of the next k-tile to stage
    const unsigned sa_step = 16u * KR * (unsigned)p.Mpad, sb_step = 16u * KR * (unsigned)p.Tpad;
#define WBF_STAGE(BUF)                                                                             \
    {                                                                                              \
        if (!(p.debug & 1))                                                                        \
        static_for<0, APT>([&](auto J) __attribute__((always_inline)) {                            \
            constexpr int j = decltype(J)::value;                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                              \
                ar, (__attribute__((address_space(3))) void*)(&As[BUF][0][0] + j * NT + wave * 64), \
                16, (int)avo[j], (int)sa, 0, 0);                                                   \
        });                                                                                        \
        if (!(p.debug & 2))                                                                        \
        static_for<0, BPT>([&](auto J) __attribute__((always_inline)) {                            \
            constexpr int j = decltype(J)::value;                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                              \
                br, (__attribute__((address_space(3))) void*)(&Bs[BUF][0][0] + j * NT + wave * 64), \
                16, (int)bvo[j], (int)sb, 0, 0);                                                   \
        });                                                                                        \
        sa += sa_step;                                                                             \
        sb += sb_step;                                                                             \
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

    const int total = 16 * nkt;
    // prologue: NBUF-1 stages in flight, the first one landed
#pragma unroll
    for (int q = 0; q < NBUF - 1; ++q)
        if (q < total) WBF_STAGE(q)
    if (total >= NBUF - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * OPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int kt = 0, xi = 0, buf = 0;
    for (int s = 0; s < total; ++s) {
        const bool more = s + NBUF - 1 < total;   // a stage is issued in this iteration
        int sbuf = buf + NBUF - 1;                // ring slot of the tile requested now
        if (sbuf >= NBUF) sbuf -= NBUF;
        uint4 a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = As[buf][lh][wm * WTM + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = Bs[buf][lh][wn * WTN + j * 32 + l31];
        static_for<0, NS>([&](auto ST) __attribute__((always_inline)) {
            constexpr int st = decltype(ST)::value;
            if constexpr (st + 1 < NS) {
                const int kr = (st + 1) * 2 + lh;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(st + 1) & 1][i] = As[buf][kr][wm * WTM + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[(st + 1) & 1][j] = Bs[buf][kr][wn * WTN + j * 32 + l31];
            }
            if (!(p.debug & 4)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, a[st & 1][i]), __builtin_bit_cast(bf16x8, b[st & 1][j]),
                        acc[i][j], 0, 0, 0);
            }
            if constexpr (st == 0) {
                if (more) WBF_STAGE(sbuf)
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (++kt == nkt) {
            kt = 0;
            // M_xi complete: Y_ab += AT[a][xi/4] * AT[b][xi%4] * M_xi,  AT = [1 1 1 0; 0 1 -1 -1]
            const int wi = xi >> 2, wj = xi & 3;
            const float r0 = wi < 3 ? 1.f : 0.f, r1 = wi == 0 ? 0.f : (wi == 1 ? 1.f : -1.f);
            const float c0 = wj < 3 ? 1.f : 0.f, c1 = wj == 0 ? 0.f : (wj == 1 ? 1.f : -1.f);
            const float k00 = r0 * c0, k01 = r0 * c1, k10 = r1 * c0, k11 = r1 * c1;
#define WBF_FOLD(Q, KQ)                                                                             \
            if ((KQ) != 0.f) {                                                                      \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                      \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                      \
                _Pragma("unroll") for (int r = 0; r < 16; ++r)                                      \
                    Y[Q][i][j][r] = fmaf((KQ), acc[i][j][r], Y[Q][i][j][r]);                        \
            }
            WBF_FOLD(0, k00)
            WBF_FOLD(1, k01)
            WBF_FOLD(2, k10)
            WBF_FOLD(3, k11)
#undef WBF_FOLD
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            ++xi;
        }
        // the next tile (issued NBUF-1 iterations ago) must have landed: all but the NBUF-2 most
        // recent stages of this thread; then every wave's reads of `buf` are done (barrier)
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * OPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (++buf == NBUF) buf = 0;
    }
#undef WBF_STAGE

    // epilogue: rows of the C/D layout are channels, columns (lane & 31) are tiles
    const int ntt = p.nty * p.ntx;
    const size_t OPL = (size_t)p.out_H * p.out_W, APL = (size_t)p.AH * p.AW;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int t = t0 + wn * WTN + j * 32 + l31;
        if (t >= p.T) continue;
        const int b = t / ntt;
        const int rr = t - b * ntt;
        const int tyl = rr / p.ntx, txl = rr - tyl * p.ntx;
        const int wy = p.ty0 + 2 * tyl - p.oy0, wx = p.tx0 + 2 * txl - p.ox0;
        const bool ok[4] = {(unsigned)wy < (unsigned)p.OH && (unsigned)wx < (unsigned)p.OW,
                            (unsigned)wy < (unsigned)p.OH && (unsigned)(wx + 1) < (unsigned)p.OW,
                            (unsigned)(wy + 1) < (unsigned)p.OH && (unsigned)wx < (unsigned)p.OW,
                            (unsigned)(wy + 1) < (unsigned)p.OH && (unsigned)(wx + 1) < (unsigned)p.OW};
        float* ob = p.out + ((size_t)b * p.out_ctot + p.out_c0) * OPL +
                    (ptrdiff_t)(p.out_y0 + wy) * p.out_W + p.out_x0 + wx;
        const float* ab = p.add ? p.add + (size_t)b * p.Cout * APL +
                                      (ptrdiff_t)(p.ay0 + wy) * p.AW + p.ax0 + wx
                                : nullptr;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // skip-add values first, all loads in flight together (indices clamped instead of
            // branches), then the stores: `out` and `add` may alias as far as the compiler knows
            float av[16][4];
            if (ab) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = min(m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, p.Cout - 1);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        av[r][q] = ok[q] ? ab[(size_t)co * APL + (q >> 1) * p.AW + (q & 1)] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co >= p.Cout) continue;
                const float bias = p.bias ? p.bias[co] : 0.f;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = Y[q][i][j][r] + bias;
                    if (ab) v[q] += av[r][q];
                    if (p.relu) v[q] = fmaxf(v[q], 0.f);
                }
                // the two pixels of a tile row as ONE 8-byte store (lanes = consecutive tiles: a
                // half-wave writes 256 contiguous bytes); 4-byte alignment is all gfx950 needs
                float* o = ob + (size_t)co * OPL;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    if (ok[2 * a] && ok[2 * a + 1])
                        *reinterpret_cast<f32x2u*>(o + (size_t)a * p.out_W) = f32x2u{v[2 * a], v[2 * a + 1]};
                    else if (ok[2 * a]) o[(size_t)a * p.out_W] = v[2 * a];
                    else if (ok[2 * a + 1]) o[(size_t)a * p.out_W + 1] = v[2 * a + 1];
                }
            }
        }
    }
    __syncthreads();   // next tile: the ring is refilled from slot 0
  }
}


// ---- 2'. the GEMMs alone: M[xi][co][t] = sum_c U16[xi][c][co] * V16[xi][c][t], fp32 M ----------------
// For the DEEP layers (>= 1024 channels on 10^2..19^2 windows: few tiles, long K) the fused kernel's
// 128 x 64 / 128 x 128 output blocks re-read U16 / V16 too often for what a CU can pull through
// LDS-DMA (~25-30 GB/s per CU beyond L2): there the 16 points become 16x more independent
// workgroups with 256 x 128 blocks (85 flop/B), M makes one round trip through HBM (small: few
// tiles) and conv_wino.hip's output-transform kernel finishes the layer.  Also the plain GEMM of
// the 'valid' K x K layers (fc6 / fc7 / score_fr: n_xi = 1).
template <int BM, int BN, int WM, int WN, int BK, int NBUF>
__global__ __launch_bounds__(WM * WN * 64, 2) void wino_gemm_bf16_kernel(const WinoBfParams p, float* M) {
    constexpr int KR = BK / 8, NS = BK / 16;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NT = WM * WN * 64;
    constexpr int AV = KR * BM, BV = KR * BN;
    constexpr int APT = AV / NT, BPT = BV / NT;
    constexpr int OPS = APT + BPT;
    static_assert(AV % NT == 0 && BV % NT == 0 && BK % 16 == 0 && NBUF >= 2, "tile config");

    __shared__ __attribute__((aligned(16))) uint4 As[NBUF][KR][BM];
    __shared__ __attribute__((aligned(16))) uint4 Bs[NBUF][KR][BN];

    // XCD-aware order: each XCD (bid % 8) walks a contiguous run of (xi, tile) work, inside a run
    // groups of 8 pixel-tiles x all channel-tiles share U and V rows in that XCD's L2
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
    const int nkt = p.Kc / BK;
    const int kcr = p.Kc >> 3;
    const __amdgpu_buffer_rsrc_t ar = mk_rsrc(p.U + (size_t)xi * kcr * p.Mpad, kcr * p.Mpad * 16);
    const __amdgpu_buffer_rsrc_t br = mk_rsrc(p.V + (size_t)xi * kcr * p.Tpad, kcr * p.Tpad * 16);

    // loop-invariant per-thread chunk offsets; the k-tile is one scalar byte offset per operand
    unsigned avo[APT], bvo[BPT];
#pragma unroll
    for (int j = 0; j < APT; ++j) {
        const int f = j * NT + tid;
        avo[j] = 16u * (unsigned)((f / BM) * p.Mpad + m0 + f % BM);
    }
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
        const int f = j * NT + tid;
        bvo[j] = 16u * (unsigned)((f / BN) * p.Tpad + t0 + f % BN);
    }
    unsigned sa = 0, sb = 0;
    const unsigned sa_step = 16u * KR * (unsigned)p.Mpad, sb_step = 16u * KR * (unsigned)p.Tpad;
#define WBG_STAGE(BUF)                                                                             \
    {                                                                                              \
        static_for<0, APT>([&](auto J) __attribute__((always_inline)) {                            \
            constexpr int j = decltype(J)::value;                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                              \
                ar, (__attribute__((address_space(3))) void*)(&As[BUF][0][0] + j * NT + wave * 64), \
                16, (int)avo[j], (int)sa, 0, 0);                                                   \
        });                                                                                        \
        static_for<0, BPT>([&](auto J) __attribute__((always_inline)) {                            \
            constexpr int j = decltype(J)::value;                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                              \
                br, (__attribute__((address_space(3))) void*)(&Bs[BUF][0][0] + j * NT + wave * 64), \
                16, (int)bvo[j], (int)sb, 0, 0);                                                   \
        });                                                                                        \
        sa += sa_step;                                                                             \
        sb += sb_step;                                                                             \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int q = 0; q < NBUF - 1; ++q)
        if (q < nkt) WBG_STAGE(q)
    if (nkt >= NBUF - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * OPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int buf = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + NBUF - 1 < nkt;
        int sbuf = buf + NBUF - 1;
        if (sbuf >= NBUF) sbuf -= NBUF;
        uint4 a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = As[buf][lh][wm * WTM + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = Bs[buf][lh][wn * WTN + j * 32 + l31];
        static_for<0, NS>([&](auto ST) __attribute__((always_inline)) {
            constexpr int st = decltype(ST)::value;
            if constexpr (st + 1 < NS) {
                const int kr = (st + 1) * 2 + lh;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(st + 1) & 1][i] = As[buf][kr][wm * WTM + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[(st + 1) & 1][j] = Bs[buf][kr][wn * WTN + j * 32 + l31];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, a[st & 1][i]), __builtin_bit_cast(bf16x8, b[st & 1][j]),
                        acc[i][j], 0, 0, 0);
            if constexpr (st == 0) {
                if (more) WBG_STAGE(sbuf)
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * OPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (++buf == NBUF) buf = 0;
    }
#undef WBG_STAGE

    // M[xi][co][t]: rows of the C/D layout are channels, columns (lane & 31) are tiles
    float* Mx = M + (size_t)xi * p.Mpad * p.Tpad;
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

// launches the GEMM kernel for n_xi points: 256 x 128 blocks when Mpad allows, else 128 x 128
void launch_gemm_bf16(hipStream_t s, WinoBfParams p, float* M, int n_xi) {
    static const int big = getenv("IISEG_BF16_GEMM_256") ? atoi(getenv("IISEG_BF16_GEMM_256")) : 1;
    if (big && p.Mpad % 256 == 0 && p.Tpad % 256 == 0 &&
        n_xi * (p.Mpad / 256) * (p.Tpad / 256) >= 200) {
        // 256 x 256 blocks, 8 waves of 128 x 64: a k-tile stage (1 us + 35 ns/KB)
        // carries twice the flops of the 256 x 128 form for 1.2x its time
        p.n_ttiles = p.Tpad / 256;
        p.n_mtiles = p.Mpad / 256;
        // Plain GEMMs (fc6 / fc7: one long k-range per block, nothing else on the CU): 32-channel k-tiles in a
        // four-deep ring -- the same LDS, three tiles in flight instead of one; same k order, same bits.  A/B on
        // one box (scripts/fc_gemm_time.py, whole call): fc6 0.886 -> 0.822 ms, fc7 0.151 -> 0.152.  The Winograd
        // GEMMs (16 points, short k-ranges) keep the two-deep 64-channel ring they were tuned on.
        // IISEG_BF16_GEMM_VAR = 0 / 1 forces one form.
        static const int env = getenv("IISEG_BF16_GEMM_VAR") ? atoi(getenv("IISEG_BF16_GEMM_VAR")) : -1;
        const int var = env >= 0 ? env : (n_xi == 1 ? 1 : 0);
        if (var == 1)
            IISEG_LAUNCH((wino_gemm_bf16_kernel<256, 256, 2, 4, 32, 4>),
                               dim3(n_xi * p.n_ttiles * p.n_mtiles), dim3(512), 0, s, p, M);
        else
            IISEG_LAUNCH((wino_gemm_bf16_kernel<256, 256, 2, 4, 64, 2>),
                               dim3(n_xi * p.n_ttiles * p.n_mtiles), dim3(512), 0, s, p, M);
        return;
    }
    p.n_ttiles = p.Tpad / 128;
    if (p.Mpad % 256 == 0) {
        p.n_mtiles = p.Mpad / 256;
        IISEG_LAUNCH((wino_gemm_bf16_kernel<256, 128, 4, 2, 64, 3>),
                           dim3(n_xi * p.n_ttiles * p.n_mtiles), dim3(512), 0, s, p, M);
    } else {
        p.n_mtiles = p.Mpad / 128;
        IISEG_LAUNCH((wino_gemm_bf16_kernel<128, 128, 2, 2, 64, 3>),
                           dim3(n_xi * p.n_ttiles * p.n_mtiles), dim3(256), 0, s, p, M);
    }
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

constexpr int WBF_BK = 64;   // k-tile of the GEMM kernel: channels are padded to it

// grid of a persistent launch: every tile if they all fit one residency round, else one round
// (256 CUs x workgroups per CU; a multiple of 8 so that a workgroup's tiles stay on "its" XCD run)
int persistent_grid(int ntiles, int per_cu) {
    static const int on = getenv("IISEG_BF16_PERSISTENT") ? atoi(getenv("IISEG_BF16_PERSISTENT")) : 1;
    const int round = 256 * per_cu;
    return (!on || ntiles <= round) ? ntiles : round;
}

struct WinoBfGeom {
    int Kc, Mpad, bm, ty0, tx0, nty, ntx, T, Tpad;
    bool fused;   // GEMMs + output transform in one kernel (else: GEMM kernel -> M -> output kernel)
};

// channels from which the separate GEMM (256 x 128 blocks, M through HBM) beats the fused kernel
int split_min_kc() {
    static const int v = getenv("IISEG_BF16_SPLIT_MIN_KC") ? atoi(getenv("IISEG_BF16_SPLIT_MIN_KC")) : 1024;
    return v;
}

int wino_bf16_geom(const iiseg_conv_desc* d, WinoBfGeom& g) {
    if (!d) return IISEG_ERR_NULL;
    if (d->KH != 3 || d->KW != 3 || d->dil != 1 || (d->flags & IISEG_CONV_TRANSPOSED2))
        return IISEG_ERR_UNSUPPORTED;
    if ((d->flags & IISEG_CONV_UNPOOL) && d->C2 != 0) return IISEG_ERR_UNSUPPORTED;
    if (d->B <= 0 || d->C1 <= 0 || d->C2 < 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 ||
        d->pad < 0 || d->OH <= 0 || d->OW <= 0 || d->oy0 < 0 || d->ox0 < 0)
        return IISEG_ERR_SHAPE;
    const int fullH = d->H + 2 * d->pad - 2, fullW = d->W + 2 * d->pad - 2;
    if (d->oy0 + d->OH > fullH || d->ox0 + d->OW > fullW) return IISEG_ERR_SHAPE;
    g.Kc = round_up(d->C1 + d->C2, WBF_BK);
    g.bm = d->Cout > 64 ? 128 : 64;
    g.Mpad = round_up(d->Cout, g.bm);
    if ((d->tile_y0 | d->tile_x0) & ~1) return IISEG_ERR_SHAPE;
    g.ty0 = d->oy0 - ((d->oy0 - d->tile_y0) & 1);
    g.tx0 = d->ox0 - ((d->ox0 - d->tile_x0) & 1);
    g.nty = (d->oy0 + d->OH - g.ty0 + 1) >> 1;
    g.ntx = (d->ox0 + d->OW - g.tx0 + 1) >> 1;
    const int64_t T = (int64_t)d->B * g.nty * g.ntx;
    const bool split = g.Kc >= split_min_kc() && g.Mpad % 128 == 0;
    // (the separate GEMM kernel may use 256-tile blocks)
    const int64_t Tpad = split ? (T + 255) / 256 * 256 : (T + 127) / 128 * 128;
    // one buffer descriptor addresses all 16 points of U16 / V16 with 32-bit byte offsets
    if (16 * Tpad * g.Kc * 2 >= ((int64_t)1 << 32) - (1 << 20) ||
        (int64_t)16 * g.Kc * g.Mpad * 2 >= ((int64_t)1 << 32) - (1 << 20))
        return IISEG_ERR_UNSUPPORTED;
    g.T = (int)T;
    g.Tpad = (int)Tpad;
    g.fused = !split;
    return IISEG_OK;
}

}  // namespace

int iiseg_wino_output_launch(hipStream_t s, const iiseg_conv_desc* d, const float* M, int Mpad,
                             int Tpad, const float* bias, const float* add, float* out);
int iiseg_gemm_output_launch(hipStream_t s, const float* M, const float* bias, float* out, int Cout,
                             int OHW, int T, int Tpad, int Mpad, int relu);

extern "C" int iiseg_conv_wino_bf16_supported(const iiseg_conv_desc* d) {
    WinoBfGeom g;
    return wino_bf16_geom(d, g) == IISEG_OK ? 1 : 0;
}

extern "C" int64_t iiseg_conv_wino_bf16_weight_bytes(const iiseg_conv_desc* d) {
    WinoBfGeom g;
    if (wino_bf16_geom(d, g) != IISEG_OK) return 0;
    return (int64_t)16 * g.Kc * g.Mpad * 2;
}

extern "C" int64_t iiseg_conv_wino_bf16_workspace_bytes(const iiseg_conv_desc* d) {
    WinoBfGeom g;
    if (wino_bf16_geom(d, g) != IISEG_OK) return 0;
    return (int64_t)16 * g.Kc * g.Tpad * 2 + (g.fused ? 0 : (int64_t)16 * g.Mpad * g.Tpad * 4);
}

extern "C" int iiseg_conv_wino_bf16_pack(void* stream, const iiseg_conv_desc* d, const float* w,
                                         int64_t stride_o, int64_t stride_c, void* U16) {
    WinoBfGeom g;
    const int st = wino_bf16_geom(d, g);
    if (st) return st;
    if (!w || !U16) return IISEG_ERR_NULL;
    if ((uintptr_t)U16 & 15) return IISEG_ERR_ALIGN;
    const int64_t n = (int64_t)g.Kc * g.Mpad;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    IISEG_LAUNCH(wino_weight_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w,
                       stride_o, stride_c, (__bf16*)U16, d->C1 + d->C2, d->Cout, g.Kc, g.Mpad);
    return iiseg_check_launch();
}

extern "C" int iiseg_conv_wino_bf16(void* stream, const iiseg_conv_desc* d, const float* x1,
                                    const float* x2, const float* pre, const float* pooled,
                                    const void* U16, const float* bias, const float* add,
                                    void* workspace, float* out, uint32_t stages) {
    WinoBfGeom g;
    const int st = wino_bf16_geom(d, g);
    if (st) return st;
    if (!x1 || !U16 || !workspace || !out) return IISEG_ERR_NULL;
    if (d->C2 > 0 && !x2) return IISEG_ERR_NULL;
    const bool unpool = (d->flags & IISEG_CONV_UNPOOL) != 0;
    if (unpool && (!pre || !pooled)) return IISEG_ERR_NULL;
    if (((uintptr_t)U16 & 15) || ((uintptr_t)workspace & 15)) return IISEG_ERR_ALIGN;
    WinoBfParams p;
    p.x1 = x1; p.x2 = x2; p.pre = pre; p.pooled = pooled;
    p.h2 = d->H / 2; p.w2 = d->W / 2;
    p.U = (const uint4*)U16;
    p.bias = bias; p.add = add;
    p.V = (uint4*)workspace;
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
    static const int dbg = getenv("IISEG_BF16_DEBUG") ? atoi(getenv("IISEG_BF16_DEBUG")) : 0;
    p.debug = dbg;
    p.out_ctot = d->out_ctot > 0 ? d->out_ctot : d->Cout;
    p.out_c0 = d->out_ctot > 0 ? d->out_c0 : 0;
    if (p.out_c0 < 0 || p.out_c0 + d->Cout > p.out_ctot) return IISEG_ERR_SHAPE;
    p.out_H = d->out_H > 0 ? d->out_H : d->OH;
    p.out_W = d->out_H > 0 ? d->out_W : d->OW;
    p.out_y0 = d->out_H > 0 ? d->out_y0 : 0;
    p.out_x0 = d->out_H > 0 ? d->out_x0 : 0;
    if (p.out_y0 < 0 || p.out_x0 < 0 || p.out_y0 + d->OH > p.out_H || p.out_x0 + d->OW > p.out_W)
        return IISEG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (stages & IISEG_WINO_INPUT) launch_wino_input_bf16(s, p, unpool);
    if ((stages & IISEG_WINO_GEMM) && !g.fused) {
        float* M = (float*)((char*)workspace + (size_t)16 * g.Kc * g.Tpad * 2);
        launch_gemm_bf16(s, p, M, 16);
        if (iiseg_check_launch()) return IISEG_ERR_LAUNCH;
        return iiseg_wino_output_launch(s, d, M, g.Mpad, g.Tpad, bias, add, out);
    }
    // The kernel keeps four fp32 output accumulator sets next to the product tile: ~345 VGPRs.  With
    // MINW = 2 (two 4-wave workgroups per CU) it is held to 256 and spills 151-172 of them to scratch;
    // with MINW = 1 (IISEG_WBF_MINW=1) the 4-wave variants are spill-free at one workgroup per CU.
    // Measured A/B on one MI355X (round 3, batch 64, profiles/r03_wino_fused_bf16_minw.txt): the
    // spill-free form is SLOWER -- 512 -> 512 at 39^2 0.984 vs 0.844 ms, 256 -> 512 at 35^2 0.467 vs
    // 0.367, 512 -> 1024 at 22^2 0.476 vs 0.367 (4-wave tiles) -- the second workgroup per CU hides
    // more latency than the scratch traffic costs, so MINW = 2 stays the default.  (The 8-wave
    // 128 x 128 variants cannot have more than 256 registers per lane and are the same in both forms.)
    static const int minw2 = getenv("IISEG_WBF_MINW") ? atoi(getenv("IISEG_WBF_MINW")) != 1 : 1;
#define WBF_UNPAREN(...) __VA_ARGS__
#define WBF_FUSED_LAUNCH(ARGS, GRID, BLOCK)                                                        \
    do {                                                                                           \
        if (minw2) IISEG_LAUNCH((wino_fused_bf16_kernel<WBF_UNPAREN ARGS, 2>), GRID, BLOCK, 0, s, p); \
        else IISEG_LAUNCH((wino_fused_bf16_kernel<WBF_UNPAREN ARGS, 1>), GRID, BLOCK, 0, s, p);      \
    } while (0)
    if (stages & IISEG_WINO_GEMM) {
        if (g.bm == 64) {
            p.n_ttiles = g.Tpad / 128;
            p.n_mtiles = g.Mpad / 64;
            WBF_FUSED_LAUNCH((64, 128, 1, 4, WBF_BK, 3), dim3(persistent_grid(p.n_ttiles * p.n_mtiles, 2)), dim3(256));
            return iiseg_check_launch();
        }
        p.n_mtiles = g.Mpad / 128;
        // few tiles: 4-wave workgroups of 128 x 64 fill the CUs better than 8-wave 128 x 128 ones
        const int w128 = (g.Tpad / 128) * p.n_mtiles;
        static const int force = getenv("IISEG_BF16_FUSED_TILE") ? atoi(getenv("IISEG_BF16_FUSED_TILE")) : 0;
        static const int var = getenv("IISEG_BF16_FUSED_VAR") ? atoi(getenv("IISEG_BF16_FUSED_VAR")) : 0;
        static const int min128 = getenv("IISEG_BF16_FUSED_128_MIN") ? atoi(getenv("IISEG_BF16_FUSED_128_MIN")) : 512;
        if (force ? force == 64 : w128 < min128) {
            p.n_ttiles = g.Tpad / 64;
            const dim3 grid(persistent_grid(p.n_ttiles * p.n_mtiles, 2));
            if (var == 1)        // 2-deep ring
                WBF_FUSED_LAUNCH((128, 64, 2, 2, 64, 2), grid, dim3(256));
            else if (var == 2)   // 32-channel k-tiles, 6-deep ring (same LDS, more tiles in flight)
                WBF_FUSED_LAUNCH((128, 64, 2, 2, 32, 6), grid, dim3(256));
            else
                WBF_FUSED_LAUNCH((128, 64, 2, 2, 64, 3), grid, dim3(256));
        } else {
            p.n_ttiles = g.Tpad / 128;
            const dim3 grid(persistent_grid(p.n_ttiles * p.n_mtiles, 1));
            if (var == 1)
                WBF_FUSED_LAUNCH((128, 128, 2, 4, 64, 2), grid, dim3(512));
            else if (var == 2)
                WBF_FUSED_LAUNCH((128, 128, 2, 4, 32, 6), grid, dim3(512));
            else if (var == 3 || g.Kc % 128)
                WBF_FUSED_LAUNCH((128, 128, 2, 4, 64, 3), grid, dim3(512));
            else
                // default: 128-channel k-tiles, 2-deep ring.  Measured on one device (A/B by
                // IISEG_BF16_FUSED_VAR): a stage costs ~1 us + 35 ns/KB whatever the ring depth, so
                // fewer, larger stages win (512 ch, 39^2 window: 0.65 vs 0.76 ms at 64-channel
                // stages, 1.16 ms at 32-channel ones)
                WBF_FUSED_LAUNCH((128, 128, 2, 4, 128, 2), grid, dim3(512));
        }
    }
    return iiseg_check_launch();
}

// ================================================================================================
// 'valid' K x K layers computed in full (FCN-8's fc6 7x7, fc7 / score_fr 1x1; models/fcn8.py:75-85)
// as im2col -> bf16 GEMM (the kernel above, one "point", no split-K: K = 25088 is 392 k-tiles of
// cheap bf16 MFMAs) -> bias / ReLU / NCHW store.  k = (c * KH + ky) * KW + kx.
// ================================================================================================
namespace {

struct GemmBfGeom {
    int K, Kc, Mpad, T, Tpad;
};

int gemm_bf16_geom(const iiseg_conv_desc* d, GemmBfGeom& g) {
    if (!d) return IISEG_ERR_NULL;
    if (d->pad != 0 || d->dil != 1 || d->C2 != 0 ||
        (d->flags & (IISEG_CONV_UNPOOL | IISEG_CONV_TRANSPOSED2)))
        return IISEG_ERR_UNSUPPORTED;
    const int fullH = d->H - d->KH + 1, fullW = d->W - d->KW + 1;
    if (d->B <= 0 || d->C1 <= 0 || d->Cout <= 0 || fullH <= 0 || fullW <= 0) return IISEG_ERR_SHAPE;
    if (d->oy0 != 0 || d->ox0 != 0 || d->OH != fullH || d->OW != fullW || d->out_ctot != 0 ||
        d->out_H != 0)
        return IISEG_ERR_UNSUPPORTED;
    g.K = d->C1 * d->KH * d->KW;
    g.Kc = round_up(g.K, WBF_BK);
    g.Mpad = round_up(d->Cout, 128);
    const int64_t T = (int64_t)d->B * fullH * fullW;
    const int64_t Tpad = (T + 255) / 256 * 256;
    if (Tpad * g.Kc * 2 >= (int64_t)1 << 31 || (int64_t)g.Kc * g.Mpad * 2 >= (int64_t)1 << 31)
        return IISEG_ERR_UNSUPPORTED;
    g.T = (int)T;
    g.Tpad = (int)Tpad;
    return IISEG_OK;
}

// U16[k/8][co][k%8] = bf16(w[co][k]), zero beyond K / Cout
__global__ void gemm_weight_bf16_kernel(const float* __restrict__ w, int64_t so, int64_t sc,
                                        __bf16* __restrict__ U, int KK, int K, int Cout, int Kc, int Mpad) {
    const int64_t n = (int64_t)Kc * Mpad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        const int64_t r = i >> 3;
        const int co = (int)(r % Mpad), k = (int)(r / Mpad) * 8 + j;
        float v = 0.f;
        if (k < K && co < Cout) {
            const int c = k / KK, tap = k - c * KK;
            v = w[co * so + c * sc + tap];
        }
        U[i] = (__bf16)v;
    }
}

// V16[k/8][t][k%8] = bf16(x[b][c][oy + ky][ox + kx]); one thread = one pixel x one 8-k chunk
__global__ __launch_bounds__(256) void gemm_im2col_bf16_kernel(const float* __restrict__ x,
                                                               uint4* __restrict__ V, int C, int H,
                                                               int W, int KH, int KW, int OH, int OW,
                                                               int K, int Kc, int T, int Tpad) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int OHW = OH * OW;
    const int b = t / OHW, r = t - b * OHW;
    const int oy = r / OW, ox = r - oy * OW;
    const int KK = KH * KW;
    const float* xb = x + (size_t)b * C * H * W + (size_t)oy * W + ox;
    for (int kc = blockIdx.y; kc < (Kc >> 3); kc += gridDim.y) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kc * 8 + j;
            const int c = k / KK, tap = k - c * KK;
            const int ky = tap / KW, kx = tap - ky * KW;
            const float t8 = xb[k < K ? (size_t)c * H * W + ky * W + kx : 0];     // (no conditional loads)
            v[j] = k < K ? t8 : 0.f;
        }
        V[(size_t)kc * Tpad + t] = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]),
                                              pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
    }
}

}  // namespace

extern "C" int iiseg_conv_gemm_bf16_supported(const iiseg_conv_desc* d) {
    GemmBfGeom g;
    return gemm_bf16_geom(d, g) == IISEG_OK ? 1 : 0;
}

extern "C" int64_t iiseg_conv_gemm_bf16_weight_bytes(const iiseg_conv_desc* d) {
    GemmBfGeom g;
    if (gemm_bf16_geom(d, g) != IISEG_OK) return 0;
    return (int64_t)g.Kc * g.Mpad * 2;
}

extern "C" int64_t iiseg_conv_gemm_bf16_workspace_bytes(const iiseg_conv_desc* d) {
    GemmBfGeom g;
    if (gemm_bf16_geom(d, g) != IISEG_OK) return 0;
    return (int64_t)g.Kc * g.Tpad * 2 + (int64_t)g.Mpad * g.Tpad * 4;
}

extern "C" int iiseg_conv_gemm_bf16_pack(void* stream, const iiseg_conv_desc* d, const float* w,
                                         int64_t stride_o, int64_t stride_c, void* U16) {
    GemmBfGeom g;
    const int st = gemm_bf16_geom(d, g);
    if (st) return st;
    if (!w || !U16) return IISEG_ERR_NULL;
    if ((uintptr_t)U16 & 15) return IISEG_ERR_ALIGN;
    const int64_t n = (int64_t)g.Kc * g.Mpad;
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    IISEG_LAUNCH(gemm_weight_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w,
                       stride_o, stride_c, (__bf16*)U16, d->KH * d->KW, g.K, d->Cout, g.Kc, g.Mpad);
    return iiseg_check_launch();
}

extern "C" int iiseg_conv_gemm_bf16(void* stream, const iiseg_conv_desc* d, const float* x,
                                    const void* U16, const float* bias, void* workspace, float* out) {
    GemmBfGeom g;
    const int st = gemm_bf16_geom(d, g);
    if (st) return st;
    if (!x || !U16 || !workspace || !out) return IISEG_ERR_NULL;
    if (((uintptr_t)U16 & 15) || ((uintptr_t)workspace & 15)) return IISEG_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    uint4* V = (uint4*)workspace;
    float* M = (float*)((char*)workspace + (size_t)g.Kc * g.Tpad * 2);
    const int kcr = g.Kc >> 3;
    IISEG_LAUNCH(gemm_im2col_bf16_kernel, dim3((g.T + 255) / 256, kcr < 512 ? kcr : 512),
                       dim3(256), 0, s, x, V, d->C1, d->H, d->W, d->KH, d->KW, d->OH, d->OW, g.K,
                       g.Kc, g.T, g.Tpad);
    WinoBfParams p = {};
    p.U = (const uint4*)U16;
    p.V = V;
    p.Kc = g.Kc; p.Mpad = g.Mpad; p.Tpad = g.Tpad; p.T = g.T;
    launch_gemm_bf16(s, p, M, 1);
    if (iiseg_check_launch()) return IISEG_ERR_LAUNCH;
    return iiseg_gemm_output_launch(s, M, bias, out, d->Cout, d->OH * d->OW, g.T, g.Tpad, g.Mpad,
                                    (d->flags & IISEG_CONV_RELU) ? 1 : 0);
}
