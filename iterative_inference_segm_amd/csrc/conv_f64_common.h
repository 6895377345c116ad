// float64 direct-convolution kernels (conv_f64.hip: static-tap gather, any 1x1 / 3x3 request;
// conv_halo_f64.hip: halo-tile form of the plain 3x3 layers): the launch parameters they share.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iiseg {

struct ConvParams64 {
    const double* x1;
    const double* x2;
    const double* pre;
    const double* pooled;
    const double* wp;
    const double* bias;
    const double* add;
    double* out;
    int B, C1, C2, H, W;
    int h2, w2;
    int Cout, OH, OW, oy0, ox0;
    int AH, AW, ay0, ax0;
    int Kpad, Mpad;
    int pad, dil;
    int P;
    int n_ptiles, n_mtiles;
    int relu;
    int out_ctot, out_c0;
    int transposed;
    int out_H, out_W, out_y0, out_x0;
    // fused 2x2 max-pool of the output (conv_halo_f64 only, Cout > 16): full (fullH / 2, fullW / 2) planes
    double* pool;
    int pool_H, pool_W;
    // DePool2D masks as BYTES (as ConvParams::mask_in / mask_out of the fp32 kernels): mask[b][c][y/2][x/2] bit
    // (y&1)*2 + (x&1) = (pre[y][x] == pooled[y/2][x/2]).  mask_out: written next to `pool` (then `out` may be NULL);
    // mask_in: the DePool2D staging reads it instead of pre / pooled (x1 = up as before)
    const unsigned char* mask_in;
    unsigned char* mask_out;
};

// conv_halo_f64.hip: true when the halo-tile kernel can run this (already validated) request
bool iiseg_conv_halo_f64_ok(const ConvParams64& p, int KH, int KW);
// BM = 16 for layers with at most 16 output channels, else 64 (p.Mpad: multiple of 64 either way)
int iiseg_launch_conv_halo_f64(hipStream_t s, const ConvParams64& p, bool unpool);

}  // namespace iiseg
