/*
 * oracle/conv_ref.c -- plain-C float64 restatement of the stride-1 cross-correlation that
 * Lasagne Conv2DLayer(flip_filters=False) / DilatedConv2DLayer compute (SURVEY P1, P11;
 * reference models/fcn8.py:34-85, models/fcn_down.py:102-104, models/fcn_up.py:83-86).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Why C and not BLAS: the DePool2D
 * equality masks (layers/mylayers.py:111-114) compare conv outputs for EXACT equality, and in
 * the pad-100 border the compared values are mathematically identical.  A BLAS GEMM rounds
 * identical columns differently depending on their position (measured: 1.7e-16 spreads at
 * conv4_1), which breaks those ties at random.  Here every output element is the same
 * fixed-order sum  ((...((w0*x0) + w1*x1) + ...) + bias)  over k = (c, ky, kx) with separate
 * IEEE multiply and add (compiled with -ffp-contract=off), so equal patches give bit-equal
 * outputs wherever they sit.  Threads split outputs, never a sum.
 */
#include <stdlib.h>
#include <string.h>

#define OB 4 /* output channels per register block */
#define XB 8 /* output pixels (along x) per register block */

/* x: (B,C,H,W)  w: (O,C,KH,KW)  b: (O) or NULL  out: (B,O,OH,OW)
 * OH = H + 2*pad - dil*(KH-1), OW likewise.  Returns 0, or -1 on allocation failure. */
int iio_conv2d_f64(const double* x, const double* w, const double* b, double* out, int B, int C,
                   int H, int W, int O, int KH, int KW, int pad, int dil, int relu) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int OH = Hp - dil * (KH - 1), OW = Wp - dil * (KW - 1);
    const long K = (long)C * KH * KW;
    /* zero-padded copy with XB slack at the row ends so blocked loads never leave the buffer */
    const long rowp = Wp + XB;
    double* xp = (double*)calloc((size_t)B * C * Hp * rowp + XB, sizeof(double));
    if (!xp) return -1;
    for (long nc = 0; nc < (long)B * C; ++nc)
        for (int y = 0; y < H; ++y)
            memcpy(xp + (nc * Hp + y + pad) * rowp + pad, x + (nc * H + y) * W,
                   sizeof(double) * W);
    const int nob = (O + OB - 1) / OB;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int n = 0; n < B; ++n) {
        for (int ob = 0; ob < nob; ++ob) {
            const int o0 = ob * OB;
            const int on = O - o0 < OB ? O - o0 : OB;
            for (int oy = 0; oy < OH; ++oy) {
                for (int ox0 = 0; ox0 < OW; ox0 += XB) {
                    double acc[OB][XB];
                    for (int i = 0; i < OB; ++i)
                        for (int j = 0; j < XB; ++j) acc[i][j] = 0.0;
                    for (int c = 0; c < C; ++c) {
                        for (int ky = 0; ky < KH; ++ky) {
                            const double* xr =
                                xp + (((long)n * C + c) * Hp + oy + ky * dil) * rowp + ox0;
                            for (int kx = 0; kx < KW; ++kx) {
                                const double* xv = xr + kx * dil;
                                const long kidx = ((long)c * KH + ky) * KW + kx;
                                for (int i = 0; i < OB; ++i) {
                                    /* rows beyond O reuse the last valid filter; never stored */
                                    const int oi = i < on ? o0 + i : o0 + on - 1;
                                    const double wv = w[oi * K + kidx];
                                    for (int j = 0; j < XB; ++j) acc[i][j] += wv * xv[j];
                                }
                            }
                        }
                    }
                    const int xn = OW - ox0 < XB ? OW - ox0 : XB;
                    for (int i = 0; i < on; ++i) {
                        double* op = out + (((long)n * O + o0 + i) * OH + oy) * OW + ox0;
                        const double bv = b ? b[o0 + i] : 0.0;
                        for (int j = 0; j < xn; ++j) {
                            double v = acc[i][j] + bv;
                            if (relu && v < 0.0) v = 0.0;
                            op[j] = v;
                        }
                    }
                }
            }
        }
    }
    free(xp);
    return 0;
}
