// 1x1 convolution on bf16 C8 activations (gfx950, v_mfma_f32_32x32x16_bf16, fp32 accumulate): FC-DenseNet's
// TransitionDown (BatchNorm -> ReLU -> 1x1 conv -> 2x2 max-pool: FC_DenseNet.layers.TransitionDown, called
// at models/FCDenseNet.py:95) and the SoftmaxLayer's 1x1 class-score convolution (models/FCDenseNet.py:134)
// as ONE kernel each on the dense block's C8 stack -- no fp32 NCHW copy of the stack, no separate
// normalisation, pooling or layout-conversion pass.
//
// The C8 format IS the MFMA operand layout of a 1x1 layer: the 8 channels of chunk c8 at a pixel are the
// 16 bytes lane (column = pixel, k-group = c8 & 1) feeds to one 32x32x16 step, so the B operand goes
// global -> registers with one 16-byte load per lane and k-step (no LDS, no gather); BatchNorm + ReLU
// (max(a x + b, 0) with the folded (a, b) of bn_fold, rounded to bf16 once, as in conv_c8_m16.hip) is
// applied to those 8 values in registers.  A workgroup = 4 waves x 128 pixels and 64 output channels.
//   POOL : a wave's 32 MFMA columns are 32 pooling windows and its four column tiles the four pixels of
//          a window, so the 2x2 max-pool is a max over four accumulators of the same lane; the pooled
//          values (+ bias) go as bf16 C8 into chunk planes [out_c8_0, ...) of the NEXT dense block's stack.
//   !POOL: columns are consecutive pixels; fp32 NCHW output (class scores).
// Bound by HBM (a TransitionDown reads its stack once per 64 output channels, mostly from L2) and by the
// BatchNorm arithmetic, not by the matrix pipe: 104 GFLOP per batch of 32 in all five TransitionDowns.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "iiseg.h"
#include "common.h"
#include "conv_common.h"

using namespace iiseg;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

constexpr int TN = 4;
constexpr int BMP = 64;            // rows the packed weights are padded to
constexpr int BN_CAP = 2048;          // channels whose (a, b) fit the LDS table

struct P1x1 {
    const u32x4* x;        // (B, in_c8tot, H, W, 8) bf16
    const float* bn_a;     // folded BatchNorm of the first Cin channels, or NULL
    const float* bn_b;
    const u32x4* wp;       // bf16 [Mpad][Cin]
    const float* bias;
    void* out;
    int in_c8tot, Cin;
    int out_c8tot, out_c8_0;
    int B, H, W, Cout, Mpad;
    int QH, QW;            // pooled map
    int N;                 // POOL: windows B * QH * QW; else pixels B * H * W
    int n_ptiles, n_mtiles;
};

// TM: 32-channel row blocks per wave (workgroup = 32 TM output channels)
template <bool POOL, bool BNRELU, int TM>
__global__ __launch_bounds__(256, TM == 1 ? 3 : 2) void conv1x1_c8_kernel(const P1x1 p) {
    constexpr int BM = 32 * TM;
    __shared__ __attribute__((aligned(16))) float sa[BNRELU ? BN_CAP : 4];
    __shared__ __attribute__((aligned(16))) float sb[BNRELU ? BN_CAP : 4];
    int pt, mt;
    tile_of_block(blockIdx.x, gridDim.x, p.n_ptiles, p.n_mtiles, pt, mt);
    const int m0 = mt * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, g = lane >> 5;
    const int HW = p.H * p.W;
    if constexpr (BNRELU) {
        for (int c = tid; c < p.Cin; c += 256) {
            sa[c] = p.bn_a[c];
            sb[c] = p.bn_b[c];
        }
        __syncthreads();
    }

    // this lane's four pixels (one per column tile): offsets in 16-byte chunks inside chunk plane 0
    size_t boff[TN];
    bool valid[TN];
    int ob = 0, oy = 0, ox = 0;               // POOL: the window; else unused
    if constexpr (POOL) {
        const int u = pt * 128 + wave * 32 + col;
        const bool ok = u < p.N;
        const int uu = ok ? u : 0;
        const int qpi = p.QH * p.QW;
        ob = uu / qpi;
        const int r = uu - ob * qpi;
        oy = r / p.QW;
        ox = r - oy * p.QW;
#pragma unroll
        for (int q = 0; q < TN; ++q) {
            valid[q] = ok;
            boff[q] = (size_t)ob * p.in_c8tot * HW + (size_t)(2 * oy + (q >> 1)) * p.W + 2 * ox + (q & 1);
        }
    } else {
#pragma unroll
        for (int q = 0; q < TN; ++q) {
            const int px = pt * 512 + wave * 128 + q * 32 + col;
            valid[q] = px < p.N;
            const int pp = valid[q] ? px : 0;
            const int b = pp / HW;
            boff[q] = (size_t)b * p.in_c8tot * HW + (size_t)(pp - b * HW);
        }
    }
    // weight rows of this lane: m0 + i * 32 + col, chunks of 8 input channels
    const int cin8 = p.Cin >> 3;
    const u32x4* wrow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) wrow[i] = p.wp + (size_t)(m0 + i * 32 + col) * cin8;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < TN; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;

    const int nks = p.Cin >> 4;
    u32x4 braw[2][TN], araw[2][TM];
    auto load = [&](int ks, int s) __attribute__((always_inline)) {
        const int c8 = 2 * ks + g;
#pragma unroll
        for (int q = 0; q < TN; ++q) braw[s][q] = p.x[boff[q] + (size_t)c8 * HW];
#pragma unroll
        for (int i = 0; i < TM; ++i) araw[s][i] = wrow[i][c8];
    };
    load(0, 0);
    for (int ks = 0; ks < nks; ++ks) {
        const int s = ks & 1;
        if (ks + 1 < nks) load(ks + 1, s ^ 1);
        u32x4 bq[TN];
        if constexpr (BNRELU) {
            const int c0 = (2 * ks + g) * 8;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(&sa[c0]), a1 = *reinterpret_cast<const f32x4*>(&sa[c0 + 4]);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(&sb[c0]), b1 = *reinterpret_cast<const f32x4*>(&sb[c0 + 4]);
            const float av[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const float bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
            for (int q = 0; q < TN; ++q)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t u = braw[s][q][w];
                    const float lo = fmaxf(__builtin_fmaf(bf_lo(u), av[2 * w], bv[2 * w]), 0.f);
                    const float hi = fmaxf(__builtin_fmaf(bf_hi(u), av[2 * w + 1], bv[2 * w + 1]), 0.f);
                    bq[q][w] = pack_bf16(lo, hi);
                }
        } else {
#pragma unroll
            for (int q = 0; q < TN; ++q) bq[q] = braw[s][q];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < TN; ++q)
                acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, araw[s][i]),
                                                                   __builtin_bit_cast(bf16x8, bq[q]), acc[i][q],
                                                                   0, 0, 0);
    }

    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    // The lane's bias values, all loaded before the first store (index clamped, not branched): a load placed
    // between two stores stays there -- one memory round trip per output element otherwise.
    float bias_r[TM][16];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            bias_r[i][r] = p.bias ? p.bias[co < p.Cout ? co : p.Cout - 1] : 0.f;
        }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (POOL) {
        if (!valid[0]) return;
        const size_t qpl = (size_t)p.QH * p.QW;
        char* ob8 = (char*)p.out + (((size_t)ob * p.out_c8tot + p.out_c8_0) * qpl + (size_t)oy * p.QW + ox) * 16 +
                    8 * g;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int cbase = m0 + i * 32 + rg * 8;           // chunk of 8 output channels
                if (cbase >= p.Cout) continue;
                float m[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = rg * 4 + k;
                    const int co = cbase + 4 * g + k;
                    const float v = fmaxf(fmaxf(acc[i][0][r], acc[i][1][r]), fmaxf(acc[i][2][r], acc[i][3][r]));
                    // channels the layer does not have stay exact zeros (padding of the chunk)
                    m[k] = co < p.Cout ? v + bias_r[i][r] : 0.f;
                }
                u32x2 w2;
                w2[0] = pack_bf16(m[0], m[1]);
                w2[1] = pack_bf16(m[2], m[3]);
                *reinterpret_cast<u32x2*>(ob8 + (size_t)(cbase >> 3) * qpl * 16) = w2;
            }
    } else {
        // (stores through a buffer descriptor, the out-of-range offset instead of a branch per element: behind
        // per-lane branches hipcc waits for the previous store before every next one)
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
            p.out, 0, (int)((unsigned)p.N * (unsigned)p.Cout * 4u), 0x00027000);
#pragma unroll
        for (int q = 0; q < TN; ++q) {
            const int px = pt * 512 + wave * 128 + q * 32 + col;
            const int pp = valid[q] ? px : 0;
            const int b = pp / HW, pix = pp - b * HW;
            const unsigned o0 = (unsigned)(b * p.Cout * HW + pix) * 4u;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                    __builtin_amdgcn_raw_buffer_store_b32(
                        __builtin_bit_cast(unsigned, acc[i][q][r] + bias_r[i][r]), ro,
                        (int)((valid[q] && co < p.Cout) ? o0 + (unsigned)(co * HW) * 4u : 0x80000000u), 0, 0);
                }
        }
    }
}

__global__ void conv1x1_c8_pack_kernel(const float* __restrict__ w, int64_t so, int64_t sc, __bf16* wp, int Cout,
                                       int Cin_w, int Cin, int Mpad) {
    const int64_t n = (int64_t)Mpad * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / Cin), c = (int)(i % Cin);
        wp[i] = (__bf16)((m < Cout && c < Cin_w) ? w[m * so + c * sc] : 0.f);
    }
}

int check1x1(int B, int Cin, int in_ctot, int H, int W, int Cout) {
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return IISEG_ERR_SHAPE;
    if (Cin % 16 || in_ctot % 8 || in_ctot < Cin) return IISEG_ERR_UNSUPPORTED;
    if ((int64_t)B * H * W >= (1ll << 31) - 1024) return IISEG_ERR_UNSUPPORTED;
    return IISEG_OK;
}

}  // namespace

extern "C" int64_t iiseg_conv1x1_c8_weight_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cin % 16) return 0;
    return (int64_t)((Cout + BMP - 1) / BMP * BMP) * Cin * 2;
}

// w: fp32 [Cout][Cin_w] with element strides (stride_o, stride_c); Cin >= Cin_w is the padded channel
// count the layer will be run with (multiple of 16; the extra columns are zero)
extern "C" int iiseg_conv1x1_c8_pack(void* stream, const float* w, int64_t stride_o, int64_t stride_c, int Cout,
                                     int Cin_w, int Cin, void* wp) {
    if (!w || !wp) return IISEG_ERR_NULL;
    if (Cout <= 0 || Cin_w <= 0 || Cin < Cin_w || Cin % 16) return IISEG_ERR_SHAPE;
    if ((uintptr_t)wp & 15) return IISEG_ERR_ALIGN;
    const int Mpad = (Cout + BMP - 1) / BMP * BMP;
    const int64_t n = (int64_t)Mpad * Cin;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    IISEG_LAUNCH(conv1x1_c8_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, stride_o, stride_c,
                 (__bf16*)wp, Cout, Cin_w, Cin, Mpad);
    return iiseg_check_launch();
}

extern "C" int iiseg_conv1x1_c8(void* stream, const void* x, int B, int Cin, int in_ctot, int H, int W,
                                const float* bn_a, const float* bn_b, const void* wp, const float* bias, int Cout,
                                int pool, void* out, int out_ctot, int out_c0) {
    const int st = check1x1(B, Cin, in_ctot, H, W, Cout);
    if (st) return st;
    if (!x || !wp || !out) return IISEG_ERR_NULL;
    if ((bn_a == nullptr) != (bn_b == nullptr)) return IISEG_ERR_NULL;
    if (bn_a && Cin > BN_CAP) return IISEG_ERR_UNSUPPORTED;
    if (((uintptr_t)x & 15) || ((uintptr_t)wp & 15) || ((uintptr_t)out & 15)) return IISEG_ERR_ALIGN;
    P1x1 p = {};
    p.x = (const u32x4*)x; p.bn_a = bn_a; p.bn_b = bn_b; p.wp = (const u32x4*)wp; p.bias = bias; p.out = out;
    p.in_c8tot = in_ctot / 8; p.Cin = Cin;
    p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.Mpad = (Cout + BMP - 1) / BMP * BMP;
    // 32-channel workgroups for layers of at most 32 output channels (the class-score layer: half the
    // accumulators, three waves per SIMD), 64-channel ones otherwise (the input is re-read per channel block)
    static const int tm_env = getenv("IISEG_C8_1X1_TM") ? atoi(getenv("IISEG_C8_1X1_TM")) : 0;
    const int tm = tm_env ? tm_env : (Cout <= 32 ? 1 : 2);
    p.n_mtiles = (Cout + 32 * tm - 1) / (32 * tm);
    hipStream_t s = (hipStream_t)stream;
#define C1X1_LAUNCH(PL, BNR)                                                                          \
    do {                                                                                              \
        if (tm == 1) IISEG_LAUNCH((conv1x1_c8_kernel<PL, BNR, 1>), grid, dim3(256), 0, s, p);          \
        else IISEG_LAUNCH((conv1x1_c8_kernel<PL, BNR, 2>), grid, dim3(256), 0, s, p);                  \
    } while (0)
    if (pool) {
        // bf16 C8 output: channels [out_c0, out_c0 + Cout) of a (B, out_ctot / 8, H / 2, W / 2, 8) tensor
        if (out_ctot % 8 || out_c0 % 8 || out_c0 < 0 || out_c0 + (Cout + 7) / 8 * 8 > out_ctot) return IISEG_ERR_SHAPE;
        p.QH = H / 2; p.QW = W / 2;
        if (p.QH <= 0 || p.QW <= 0) return IISEG_ERR_SHAPE;
        p.out_c8tot = out_ctot / 8; p.out_c8_0 = out_c0 / 8;
        p.N = B * p.QH * p.QW;
        p.n_ptiles = (p.N + 127) / 128;
        const dim3 grid(p.n_ptiles * p.n_mtiles);
        if (bn_a) C1X1_LAUNCH(true, true); else C1X1_LAUNCH(true, false);
    } else {
        // fp32 NCHW output (B, Cout, H, W), addressed with 32-bit byte offsets
        if ((int64_t)B * H * W * Cout * 4 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
        p.N = B * H * W;
        p.n_ptiles = (p.N + 511) / 512;
        const dim3 grid(p.n_ptiles * p.n_mtiles);
        if (bn_a) C1X1_LAUNCH(false, true); else C1X1_LAUNCH(false, false);
    }
#undef C1X1_LAUNCH
    return iiseg_check_launch();
}
