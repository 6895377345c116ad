// Shared pieces of the implicit-GEMM convolution kernels (conv_igemm.hip: table-driven gather,
// any filter; conv_taps.hip: static-tap buffer-load gather, 1x1 / 3x3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iiseg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// compile-time unrolled loop: f(ic<I>) for I in [B, E)
template <int V> struct ic { static constexpr int value = V; };
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(f);
    }
}

struct ConvParams {
    const float* x1;
    const float* x2;
    const float* pre;
    const float* pooled;
    const float* wp;
    const int4* ktab;
    const float* bias;
    const float* add;
    float* out;
    int B, C1, C2, H, W;
    int h2, w2;  // pooled dims (unpool mode)
    int Cout, OH, OW, oy0, ox0;
    int AH, AW, ay0, ax0;
    int Kpad, Mpad;
    int pad, dil;
    int P;  // B*OH*OW
    int n_ptiles, n_mtiles;
    int relu;
    int debug_nogather;
    int out_ctot, out_c0;  // destination channel slice (out_ctot == Cout, out_c0 == 0: dense)
    int transposed;        // 3x3 stride-2 transposed convolution (conv_taps only)
    int out_H, out_W, out_y0, out_x0;  // destination planes / placement (dense: OH, OW, 0, 0)
    float* pool;           // fused 2x2 max-pool of the output (conv_halo only), full (fullH/2, fullW/2) planes
    int pool_H, pool_W;
    // DePool2D equality masks as BYTES instead of the pre-pool map (conv_halo kernels only):
    // mask[b][c][y/2][x/2] bit (y&1)*2 + (x&1) = (pre[y][x] == pooled[y/2][x/2]).  mask_out: written
    // by the fused-pool epilogue next to `pool` (then `out` may be NULL: the pre-pool map is not
    // stored at all); mask_in: the unpool input gather reads it instead of pre / pooled
    // (x1 = up as before).
    const unsigned char* mask_in;
    unsigned char* mask_out;
    // fused input BatchNorm + ReLU (conv_halo16 only): x <- max((x - mean[c]) * (gamma[c] * inv_std[c])
    // + beta[c], 0) while the patch is staged; in_bstride = elements between images of x1 (0: dense)
    const float *bn_beta, *bn_gamma, *bn_mean, *bn_inv_std;
    long long in_bstride;
};

// Tile order: the XCD that gets block b is b % 8 (round-robin dispatch, speed only), so give
// each XCD a contiguous run of tiles and, inside a run, walk groups of 8 pixel-tiles x all
// channel-tiles so that co-resident blocks of one XCD share both X rows and W rows in its L2.
__device__ inline void tile_of_block(int bid, int nblocks, int n_p, int n_m, int& pt, int& mt) {
    const int q = nblocks / 8, r = nblocks % 8;
    const int xcd = bid % 8, l = bid / 8;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + l;
    constexpr int GP = 8;
    const int gsize = GP * n_m;
    const int g = v / gsize, rr = v % gsize;
    const int gp = min(GP, n_p - g * GP);
    pt = g * GP + rr % gp;
    mt = rr / gp;
}

// Epilogue shared by both kernels: bias, skip add (center-cropped), ReLU, NCHW store.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p,
                                              f32x16 (&acc)[BM / WM / 32][BN / WN / 32], int p0,
                                              int m0, int wm, int wn, int lane) {
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    const int l31 = lane & 31, lh = lane >> 5;
    const int OHW = p.OH * p.OW;
    // C/D layout of the 32x32 MFMA: column = lane & 31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pe = p0 + wn * WTN + j * 32 + l31;
        if (pe >= p.P) continue;
        const int eb = pe / OHW;
        const int rem = pe - eb * OHW;
        const int eoy = rem / p.OW, eox = rem - eoy * p.OW;
        const size_t OPL = (size_t)p.out_H * p.out_W;  // destination plane (dense: OH*OW)
        float* outp = p.out + ((size_t)eb * p.out_ctot + p.out_c0) * OPL +
                      (size_t)(p.out_y0 + eoy) * p.out_W + p.out_x0 + eox;
        const float* addp = nullptr;
        size_t AHW = 0;
        if (p.add) {
            AHW = (size_t)p.AH * p.AW;
            addp = p.add + (size_t)eb * p.Cout * AHW + (size_t)(p.ay0 + eoy) * p.AW + p.ax0 + eox;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < p.Cout) {
                    float v = acc[i][j][r];
                    if (p.bias) v += p.bias[co];
                    if (addp) v += addp[(size_t)co * AHW];
                    if (p.relu) v = fmaxf(v, 0.f);
                    outp[(size_t)co * OPL] = v;
                }
            }
        }
    }
}

typedef int i32x4s __attribute__((ext_vector_type(4)));

// Buffer descriptor as four scalar words (for the inline-asm LDS-DMA below): base, stride 0,
// num_records = bytes, raw 32-bit data format.  Every word is made wave-uniform explicitly.
__device__ __forceinline__ i32x4s mk_srsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    i32x4s r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00027000;
    return r;
}
// One LDS-DMA piece: every lane fetches the 16 bytes at (descriptor base + voff + soff) and the wave's
// 64 pieces land contiguously at LDS byte address `lds` (wave-uniform).  Written as inline asm ON
// PURPOSE: beside a `__builtin_amdgcn_raw_ptr_buffer_load_lds` hipcc puts `s_waitcnt vmcnt(0)` in
// front of every later LDS read it cannot prove disjoint (all of them, with a runtime ring index),
// which serialises the prefetch with the MFMAs it should run under (checked in the .s).  An asm DMA
// is outside hipcc's bookkeeping: it is retired by the explicit `s_waitcnt vmcnt(0)` + barrier at the
// top of the k-loop and nothing else.  M0 (the DMA's LDS base) is saved and restored in the same
// statement (the compiler does not expect it to change).
__device__ __forceinline__ void dma16(i32x4s rsrc, unsigned lds, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}


}  // namespace iiseg
