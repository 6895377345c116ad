/*
 * iiseg.h -- C ABI of libiiseg_hip.so, the MI355X (gfx950) kernel library behind the
 * iterative-inference hot path of adri-romsor/iterative_inference_segm.
 *
 * The reference has no FFI: its device boundary is the four Theano functions compiled at
 * iterative_inference.py:187-210 (pred_fcn_fn, pred_dae_fn, de_fn, val_fn), whose graphs are
 * made of the Lasagne layers cited per entry point below.  Each entry point replaces the
 * Theano op(s) that those layers lower to.
 *
 * Conventions:
 *   - plain C, no torch types; all tensor pointers are DEVICE pointers to C-contiguous NCHW
 *     float32 unless stated; the caller owns every byte (inputs, outputs, workspaces);
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); it never
 *     allocates, never synchronises, and is safe to capture into a hipGraph;
 *   - return value: IISEG_OK (0) or a negative iiseg_status; iiseg_strerror() names it;
 *   - calls on different streams are thread-safe; one stream = program order.
 */
#ifndef IISEG_H
#define IISEG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum iiseg_status {
    IISEG_OK = 0,
    IISEG_ERR_NULL = -1,       /* a required pointer is NULL                         */
    IISEG_ERR_SHAPE = -2,      /* inconsistent / unsupported shape                   */
    IISEG_ERR_ALIGN = -3,      /* pointer not aligned as the kernel requires         */
    IISEG_ERR_LAUNCH = -4,     /* hipLaunchKernel / hipGetLastError failed           */
    IISEG_ERR_UNSUPPORTED = -5 /* valid request, but no kernel variant implements it */
} iiseg_status;

const char* iiseg_strerror(int status);
/* the HIP runtime's name for the error behind the last IISEG_ERR_LAUNCH of this library */
const char* iiseg_last_hip_error(void);
/* ABI version, bumped on any signature change. */
int iiseg_abi_version(void);
/* Name of the GPU architecture the library was compiled for ("gfx950"). */
const char* iiseg_target_arch(void);

/* Launch profiling (measurement only; bench.py's roofline pass).  Between iiseg_profile_begin(capacity)
 * and iiseg_profile_end every kernel launch of this library carries its own start / stop HIP events ON ITS
 * DISPATCH (hipExtLaunchKernelGGL), in launch order; iiseg_profile_count() = launches so far (-1 when
 * off); iiseg_profile_end waits for them and writes the elapsed milliseconds of up to `capacity` launches
 * -- each kernel's execution time on the stream it was launched on, the duration rocprofv3 reports, without
 * the barrier packets separately recorded events put between two kernels (7 % of a 0.18 ms kernel) --
 * and returns their number.  Process-wide state: one profiling pass at a time. */
int iiseg_profile_begin(int capacity);
int iiseg_profile_count(void);
int iiseg_profile_end(float* ms, int capacity);

/* ---------------------------------------------------------------------------------------
 * Implicit-GEMM convolution (stride 1), fp32 MFMA.
 * Replaces: Lasagne Conv2DLayer / DilatedConv2DLayer (+ fused Elemwise bias/ReLU), i.e.
 * Theano CorrMM, at models/fcn8.py:34-85,92-93,102-103; models/fcn_down.py:102-104;
 * models/fcn_up.py:83-86; models/contextmod_dae.py:74-105.  Fusions: the h-concat of
 * models/model_helpers.py:93-94 (two sources, h first), DePool2D of layers/mylayers.py:88-115
 * as an input gather, ElemwiseSumLayer/CroppingLayer of models/fcn_up.py:96-113 as epilogue.
 * ------------------------------------------------------------------------------------- */

#define IISEG_CONV_RELU 1u   /* out = max(out, 0)                                   */
#define IISEG_CONV_UNPOOL 2u /* logical input = eq-mask unpool(up=x1, pre, pooled)  */
/* KxK (K = 3 or 4) stride-2 TRANSPOSED convolution, crop='valid': Deconv2DLayer of FC-DenseNet's
 * TransitionUp (K=3, models/FCDenseNet.py:119 via FC_DenseNet.layers) and of the DAE's
 * unpool_type='standard' (K=4, models/fcn_up.py:41-45; float32 only).  Logical output
 * ((H-1)*2+K, (W-1)*2+K); w is the reference layout W[in][out][K][K] (pass stride_o = K*K,
 * stride_c = Cout*K*K), the spatial flip of Lasagne's gradient form is applied
 * while packing; pad/dil are ignored. */
#define IISEG_CONV_TRANSPOSED2 4u
/* iiseg_conv_c8 only: the split-operand ("bf16x3") fp32-class mode of the 16-bit matrix pipe.  Every
 * bf16 C8 tensor of the call (x1, `up`, add of kind 1, out of kind 1, pool_out) is a hi / lo PAIR in
 * one allocation, (B, 2 C/8, H, W, 8): chunks [0, C/8) of an image = bf16(v), chunks [C/8, 2 C/8) =
 * bf16(v - hi); the packed weights hold two k-groups [W_hi | W_lo] (pack a (Cout, 2 C1) filter built
 * that way with iiseg_conv_halo_bf16_pack); the layer accumulates x_lo W_hi + x_hi W_lo + x_hi W_hi
 * in fp32.  d->C1 stays the logical channel count; C2 must be 0. */
#define IISEG_CONV_X3 8u
/* iiseg_conv_c8_slice only: the logical input is x1 with one zero inserted between neighbouring pixels and a
 * two-pixel zero frame, z[2 i + 2][2 j + 2] = x1[i][j] on (H, W) = (2 h + 3, 2 w + 3) -- a 'valid' 3x3
 * correlation of z with the in / out-swapped filter IS the 3x3 stride-2 'valid' transposed convolution of
 * FC-DenseNet's TransitionUp (models/FCDenseNet.py:118-121; Deconv2DLayer, P3) */
#define IISEG_CONV_ZINS 16u

typedef struct iiseg_conv_desc {
    /* logical input (after concat / unpool): (B, C1 + C2, H, W) */
    int32_t B, C1, C2, H, W;
    /* filter */
    int32_t Cout, KH, KW, pad, dil;
    /* logical output is (H + 2*pad - dil*(KH-1), W + 2*pad - dil*(KW-1)); only the window
     * [oy0, oy0+OH) x [ox0, ox0+OW) is computed and `out` is (B, Cout, OH, OW). */
    int32_t oy0, ox0, OH, OW;
    /* optional epilogue add: out += add[b, co, ay0 + oy, ax0 + ox]; add is (B, Cout, AH, AW) */
    int32_t AH, AW, ay0, ax0;
    uint32_t flags;
    /* packed-weight geometry produced by iiseg_conv_pack_f32 */
    int32_t Kpad, Mpad;
    /* destination as a channel slice of a wider tensor (B, out_ctot, OH, OW), first channel
     * out_c0 -- lets dense blocks grow a preallocated stack without concat copies
     * (ConcatLayer([stack, l]), models/FCDenseNet.py:92).  out_ctot == 0: dense (B, Cout, OH, OW). */
    int32_t out_ctot, out_c0;
    /* destination as a spatial window of a larger plane: the (OH, OW) result is written at
     * (out_y0, out_x0) of planes of size (out_H, out_W).  out_H == 0: dense (OH, OW) planes.
     * Used to compute only the part of a decoder level that reaches the final center crop
     * (models/fcn_up.py:104-113) while every tensor keeps its full-size addressing. */
    int32_t out_H, out_W, out_y0, out_x0;
    /* Winograd path only: parity (0 or 1) of the absolute output row / column at which the 2x2
     * tiles start.  Must be the same for every launch of a layer whose windows are expected to
     * agree bit for bit; pick the parity of the most frequent window origin. */
    int32_t tile_y0, tile_x0;
} iiseg_conv_desc;

/* Number of int32x4 entries of the gather table for `d` (== d->Kpad). */
int iiseg_conv_ktab_entries(const iiseg_conv_desc* d);
/* Fills d->Kpad / d->Mpad for the (C1+C2, KH, KW, Cout) of `d`. */
int iiseg_conv_plan(iiseg_conv_desc* d);

/* Packs reference-layout weights into the kernel layout Wp[Kpad][Mpad] (k = (c*KH+ky)*KW+kx)
 * and builds the gather table.  `w` element (o, c, ky, kx) is read at
 * w[o*stride_o + c*stride_c + ky*KW + kx], which covers Conv2DLayer W[out,in,kh,kw]
 * (stride_o = Cin*KH*KW, stride_c = KH*KW) and DilatedConv2DLayer W[in,out,kh,kw]
 * (stride_o = KH*KW, stride_c = Cout*KH*KW; P11).  wp: Kpad*Mpad floats; ktab: Kpad*4 int32. */
int iiseg_conv_pack_f32(void* stream, const iiseg_conv_desc* d, const float* w,
                        int64_t stride_o, int64_t stride_c, float* wp, int32_t* ktab);

/* out = epilogue(conv(input)).  x2 may be NULL when C2 == 0.  With IISEG_CONV_UNPOOL:
 * x1 = up (B, C1, H/2, W/2), pre (B, C1, H, W), pooled (B, C1, H/2, W/2), C2 must be 0.
 * bias (Cout) and add may be NULL. */
int iiseg_conv_f32(void* stream, const iiseg_conv_desc* d, const float* x1, const float* x2,
                   const float* pre, const float* pooled, const float* wp, const int32_t* ktab,
                   const float* bias, const float* add, float* out);

/* iiseg_conv_f32 that also writes the 2x2/2 max-pool (ignore_border) of its output: the
 * Pool2DLayer that follows the conv (models/fcn_down.py:122, models/fcn8.py:38-45) fused into the
 * epilogue of the halo-tile kernel, saving the pool kernel's read of the full-resolution map.
 * pool_out is the FULL pooled tensor (B, Cout, fullH/2, fullW/2); the pooled positions of the
 * computed window are written in place.  Supported (iiseg_conv_pool_supported) for the 3x3 layers
 * the halo kernel takes (16 < Cout < 256) with an even window origin and whole pooling windows. */
/* 1 when iiseg_conv_f32 runs this (planned, plain: no add / pool / masks) request on the vector-ALU kernel
 * for layers between at most 16 channels on either side (csrc/conv_small.hip: the context module's 1x1 /
 * dilated 3x3 layers, models/contextmod_dae.py:74-105) -- a scheduling fact, for profiles. */
int iiseg_conv_small_supported(const iiseg_conv_desc* d);
int iiseg_conv_pool_supported(const iiseg_conv_desc* d);
int iiseg_conv_pool_f32(void* stream, const iiseg_conv_desc* d, const float* x1, const float* x2,
                        const float* pre, const float* pooled, const float* wp, const int32_t* ktab,
                        const float* bias, const float* add, float* out, float* pool_out);

/* DePool2D equality masks as BYTES (halo-tile kernels only): the reference's DePool2D needs
 * pre == pooled per element (layers/mylayers.py:111-114); instead of keeping the pre-pool map for
 * that comparison, the conv whose epilogue does the pooling can write
 *     mask[b][c][y/2][x/2] bit (y & 1) * 2 + (x & 1) = (pre[b][c][y][x] == pooled[b][c][y/2][x/2])
 * (B, Cout, fullH/2, fullW/2 bytes, same placement rules as pool_out) and the decoder conv can take
 * its IISEG_CONV_UNPOOL input from x1 = up and mask_in = that tensor instead of pre / pooled: 5 bytes
 * read per unpooled element instead of 12, and the largest map of every level is never written or
 * re-read.  iiseg_conv_mask_f32 is iiseg_conv_pool_f32 with both options: `out` may be NULL when
 * pool_out and mask_out are given (the pre-pool map is then not stored at all); with mask_in, pre
 * and pooled are ignored (pass NULL).  Same decisions, bit for bit, as the pre / pooled form.
 * iiseg_conv_mask_supported: 1 if the request runs on a halo kernel (3x3, dil 1, Cout < 256; the
 * pool / mask_out part needs 16 < Cout and whole pooling windows, as iiseg_conv_pool_supported). */
int iiseg_conv_mask_supported(const iiseg_conv_desc* d);
int iiseg_conv_mask_f32(void* stream, const iiseg_conv_desc* d, const float* x1, const float* x2,
                        const float* pre, const float* pooled, const uint8_t* mask_in,
                        const float* wp, const int32_t* ktab, const float* bias, const float* add,
                        float* out, float* pool_out, uint8_t* mask_out);

/* 3x3 convolution with at most 16 output channels whose INPUT is normalised and rectified while it
 * is staged: x <- max((x - mean[c]) * (gamma[c] * inv_std[c]) + beta[c], 0) per input channel --
 * BN_ReLU_Conv of FC-DenseNet (models/FCDenseNet.py:12,90,109,123) as one kernel instead of
 * iiseg_bn_relu_f32 + iiseg_conv_f32 (the normalised copy of the growing feature stack is never
 * written).  x is (B, >= C1, H, W) with x_bstride elements between images (the first C1 channels of
 * a stack).  Same arithmetic as iiseg_bn_relu_f32; zero padding pads the normalised map. */
int iiseg_conv_bnrelu_supported(const iiseg_conv_desc* d);
int iiseg_conv_bnrelu_f32(void* stream, const iiseg_conv_desc* d, const float* x, int64_t x_bstride,
                          const float* beta, const float* gamma, const float* mean,
                          const float* inv_std, const float* wp, const int32_t* ktab,
                          const float* bias, float* out);

/* Winograd F(2x2,3x3) form of the same convolution (same call sites as iiseg_conv_f32 for
 * 3x3, dil 1, stride 1 layers with (C1+C2) % 16 == 0, no TRANSPOSED2; with IISEG_CONV_UNPOOL the
 * DePool2D mask is applied while the input transform loads its 4x4 patches, operands as in
 * iiseg_conv_f32):
 * 2.25x fewer fp32 multiplies on the matrix pipe.  Same descriptor (window, placement, channel
 * slice, add, ReLU; Kpad/Mpad are ignored), results equal to iiseg_conv_f32 up to fp32 rounding,
 * and bit-identical between any two windows of one layer launched with the same tile anchor
 * parity (tile_y0 / tile_x0 of the descriptor).
 *   iiseg_conv_wino_supported      1 if `d` can run on this path, else 0
 *   iiseg_conv_wino_weight_elems   floats of the transformed weights U = G g G^T
 *   iiseg_conv_wino_workspace_elems floats of the caller-owned workspace one call needs
 *                                  (transformed input V + products M), for the window in `d`
 *   iiseg_conv_wino_pack_f32       w (layout as in iiseg_conv_pack_f32) -> U, once per layer
 *   iiseg_conv_wino_f32            input transform -> 16 GEMMs -> output transform + epilogue;
 *                                  `stages` selects which of the three kernels are enqueued
 *                                  (IISEG_WINO_ALL normally; single stages let a caller time
 *                                  the MFMA kernel on its own) */
#define IISEG_WINO_INPUT 1u
#define IISEG_WINO_GEMM 2u
#define IISEG_WINO_OUTPUT 4u
#define IISEG_WINO_ALL 7u
/* with IISEG_WINO_FUSED the GEMM stage also applies the output transform + epilogue (the
 * products stay in registers; IISEG_WINO_OUTPUT is ignored): INPUT | GEMM | FUSED is a full call */
#define IISEG_WINO_FUSED 8u
int iiseg_conv_wino_supported(const iiseg_conv_desc* d);
int64_t iiseg_conv_wino_weight_elems(const iiseg_conv_desc* d);
int64_t iiseg_conv_wino_workspace_elems(const iiseg_conv_desc* d);
int iiseg_conv_wino_pack_f32(void* stream, const iiseg_conv_desc* d, const float* w,
                             int64_t stride_o, int64_t stride_c, float* U);
int iiseg_conv_wino_f32(void* stream, const iiseg_conv_desc* d, const float* x1, const float* x2,
                        const float* pre, const float* pooled, const float* U, const float* bias,
                        const float* add, float* workspace, float* out, uint32_t stages);

/* ---------------------------------------------------------------------------------------
 * 16-bit MFMA path (BASELINE north_star "fp16 MFMA peak", configs[2] "bf16 with fp32 accumulate"):
 * the same Winograd F(2x2,3x3) convolution with bf16 GEMM operands on v_mfma_f32_32x32x16_bf16
 * and fp32 accumulation.  Tensors at the ABI stay fp32 NCHW; weights are rounded to bf16 when they
 * are packed, activations when the input transform writes its workspace; the 16 GEMMs, the output
 * transform and the epilogue (bias, add, ReLU, window, placement, channel slice -- same descriptor
 * fields as iiseg_conv_wino_f32) run as one kernel.  Statistical parity only (8 significant bits
 * per operand).  Any (C1 + C2) is accepted (channels are zero-padded to the 64-channel k-tile).
 *   iiseg_conv_wino_bf16_supported        1 if `d` can run on this path, else 0
 *   iiseg_conv_wino_bf16_weight_bytes     bytes of the packed bf16 weights U16
 *   iiseg_conv_wino_bf16_workspace_bytes  bytes of caller-owned workspace (transformed input V16;
 *                                         for layers with >= 1024 channels, which run the GEMMs as
 *                                         their own kernel with 256 x 128 blocks, also the fp32
 *                                         products M)
 *   iiseg_conv_wino_bf16_pack             w (layout as in iiseg_conv_pack_f32) -> U16, once per layer
 *   iiseg_conv_wino_bf16                  stages: IISEG_WINO_INPUT | IISEG_WINO_GEMM (= all)
 * ------------------------------------------------------------------------------------- */
int iiseg_conv_wino_bf16_supported(const iiseg_conv_desc* d);
int64_t iiseg_conv_wino_bf16_weight_bytes(const iiseg_conv_desc* d);
int64_t iiseg_conv_wino_bf16_workspace_bytes(const iiseg_conv_desc* d);
int iiseg_conv_wino_bf16_pack(void* stream, const iiseg_conv_desc* d, const float* w,
                              int64_t stride_o, int64_t stride_c, void* U16);
int iiseg_conv_wino_bf16(void* stream, const iiseg_conv_desc* d, const float* x1, const float* x2,
                         const float* pre, const float* pooled, const void* U16, const float* bias,
                         const float* add, void* workspace, float* out, uint32_t stages);

/* bf16-operand GEMM form of the 'valid' K x K layers computed in full into a dense output (fc6 7x7,
 * fc7 / score_fr 1x1: models/fcn8.py:75-85), the counterpart of iiseg_conv_gemm_f32: im2col to a
 * bf16 workspace, one bf16 GEMM (no split-K), bias / ReLU / NCHW store.  U16: ..._weight_bytes bytes
 * from iiseg_conv_gemm_bf16_pack (w layout as in iiseg_conv_pack_f32); workspace: ..._workspace_bytes. */
int iiseg_conv_gemm_bf16_supported(const iiseg_conv_desc* d);
int64_t iiseg_conv_gemm_bf16_weight_bytes(const iiseg_conv_desc* d);
int64_t iiseg_conv_gemm_bf16_workspace_bytes(const iiseg_conv_desc* d);
int iiseg_conv_gemm_bf16_pack(void* stream, const iiseg_conv_desc* d, const float* w,
                              int64_t stride_o, int64_t stride_c, void* U16);
int iiseg_conv_gemm_bf16(void* stream, const iiseg_conv_desc* d, const float* x, const void* U16,
                         const float* bias, void* workspace, float* out);

/* Halo-tile DIRECT 3x3 convolution with bf16 MFMA operands (fp32 accumulate) for the shallow layers
 * the Winograd form does not pay for (few channels, large maps; HBM-bound): same descriptor
 * semantics and fusions as iiseg_conv_f32 / iiseg_conv_pool_f32 for 3x3, dil 1, stride 1 layers
 * (two sources need C1 % 16 == 0).  wp16: iiseg_conv_halo_bf16_weight_bytes bytes, filled by
 * iiseg_conv_halo_bf16_pack from w (layout as in iiseg_conv_pack_f32).  pool_out (may be NULL): the
 * FULL pooled tensor, as in iiseg_conv_pool_f32 (even window origin, whole pooling windows, no add).
 * mask_in / mask_out (may be NULL): DePool2D masks as bytes, see iiseg_conv_mask_f32. */
int iiseg_conv_halo_bf16_supported(const iiseg_conv_desc* d);
int64_t iiseg_conv_halo_bf16_weight_bytes(const iiseg_conv_desc* d);
int iiseg_conv_halo_bf16_pack(void* stream, const iiseg_conv_desc* d, const float* w,
                              int64_t stride_o, int64_t stride_c, void* wp16);
int iiseg_conv_halo_bf16(void* stream, const iiseg_conv_desc* d, const float* x1, const float* x2,
                         const float* pre, const float* pooled, const void* wp16, const float* bias,
                         const float* add, float* out, float* pool_out, const uint8_t* mask_in,
                         uint8_t* mask_out);

/* float64 Winograd F(2x2,3x3) form of iiseg_conv_f64 for the wide 3x3 layers of the strict-parity
 * path (stride 1, dil 1, (C1+C2) % 16 == 0, no TRANSPOSED2; with IISEG_CONV_UNPOOL x1 = up, pre,
 * pooled as in iiseg_conv_f64 and the DePool2D mask is applied by the input transform): input transform ->
 * 16 batched GEMMs on v_mfma_f64_16x16x4_f64 (operands by LDS-DMA) -> output transform + epilogue.
 * Same descriptor semantics as iiseg_conv_wino_f32 (window, placement, channel slice, add, ReLU, tile
 * anchor parity tile_y0 / tile_x0); results equal to iiseg_conv_f64 up to float64 rounding.
 * U: iiseg_conv_wino_f64_weight_elems doubles (iiseg_conv_wino_pack_f64, w as in iiseg_conv_pack_f64);
 * workspace: iiseg_conv_wino_f64_workspace_elems doubles, caller-owned. */
int iiseg_conv_wino_f64_supported(const iiseg_conv_desc* d);
int64_t iiseg_conv_wino_f64_weight_elems(const iiseg_conv_desc* d);
int64_t iiseg_conv_wino_f64_workspace_elems(const iiseg_conv_desc* d);
int iiseg_conv_wino_pack_f64(void* stream, const iiseg_conv_desc* d, const double* w,
                             int64_t stride_o, int64_t stride_c, double* U);
int iiseg_conv_wino_f64(void* stream, const iiseg_conv_desc* d, const double* x1, const double* x2,
                        const double* pre, const double* pooled, const double* U, const double* bias,
                        const double* add, double* workspace, double* out);

/* bf16 "C8" activations: the 16-bit MFMA path with the tensors BETWEEN layers already in the matrix
 * pipe's operand format.  A C8 tensor of C channels is (B, ceil(C/8), H, W, 8) bf16 -- the 8 channels
 * a lane feeds to one MFMA are 16 contiguous bytes -- with zero channels beyond C.  Same Lasagne
 * Conv2DLayer(3x3, stride 1) call sites as iiseg_conv_halo_bf16 (models/fcn8.py:34-71,
 * models/fcn_down.py:102-104, models/fcn_up.py:83-86); descriptor semantics as iiseg_conv_f32 with
 * d->C1 / d->C2 = the PHYSICAL channel counts of the C8 inputs (multiples of 16; d->C2 > 0 = channel
 * concat, x1 first) and weights packed by iiseg_conv_halo_bf16_pack for a descriptor with those
 * counts (rows of channels the reference layer does not have are zero).
 *   x1, x2      C8 inputs; with IISEG_CONV_UNPOOL x1 = `up` (B, C1/8, H/2, W/2, 8) and mask_in =
 *               (B, C1/8, H/2, W/2, 8) bytes, bit (y & 1) * 2 + (x & 1) of byte j: pre == pooled for
 *               channel 8 c8 + j at pixel (y, x) (DePool2D, layers/mylayers.py:88-115)
 *   add         NULL (add_kind 0), C8 bf16 (1) or C8 fp32 (2: (B, Cout/8, AH, AW, 8) floats)
 *   out         out_kind 0: not stored (pool_out given), 1: C8 bf16, 2: C8 fp32, 3: NCHW fp32
 *               (B, out_ctot, out_H, out_W), Cout <= 32, no add / pool -- the class-score layer
 *   pool_out    C8 bf16 (B, Cout/8, fullH/2, fullW/2, 8): 2x2 max-pool of the fp32 results, rules as
 *               iiseg_conv_pool_f32; mask_out: the DePool2D mask bytes of those windows, taken from
 *               the fp32 results before rounding (same decisions as the fp32-activation form)
 * Pixel tilings (chosen per launch from its geometry alone; every output is one fixed-order sum
 * whichever runs it): th x tw pixel tiles of one image (256 or 512 pixels, shape chosen for the fewest
 * tiles) or FLAT, 256 consecutive window pixels of the whole batch per workgroup (small windows).  With
 * pool_out the pixels of a tile are ordered by 2x2 pooling windows (even tile shapes; the flat list is
 * a list of windows), so the fused pool is available on every tiling.
 * iiseg_conv_c8_is_flat: 1 if the launch (without pool_out) uses the flat tiling.
 * (environment overrides for timing experiments: IISEG_C8_TILING=1|2, IISEG_C8_TALL=0|1,
 * IISEG_C8_SHAPE=th,tw)
 * iiseg_conv_c8_tiling: out4 = {0 rect-256 / 1 rect-512 / 2 flat, th (flat: patch rows), tw (flat:
 * patch row stride), quad order} of the launch with (pool != 0) or without a fused pool.
 * iiseg_nchw_to_c8 / iiseg_c8_to_nchw: layout converters (fp32 NCHW <-> C8 bf16, C8n chunks/image).
 * iiseg_pool_mask_c8: 2x2 max-pool (+ mask bytes, may be NULL) of the pooled-coordinate window
 * (y0, x0, wh, ww) from a stored piece `pre` (BC8 = B * chunks, PH, PW, 8; bf16, or with pre_f32 the
 * C8 fp32 results of out_kind 2: the same fp32 comparisons as the fused pool) whose corner is at
 * (py0, px0) of the (H, W) map, into the full (BC8, H/2, W/2, 8) pooled / mask tensors. */
int iiseg_conv_c8_supported(const iiseg_conv_desc* d);
int iiseg_conv_c8_is_flat(const iiseg_conv_desc* d);
int iiseg_conv_c8_tiling(const iiseg_conv_desc* d, int pool, int32_t* out4);
/* Test hook (process-wide, not for production use): kind -1 = automatic (default), 0 / 1 / 2 force the
 * rect-256 / rect-512 / flat tiling where it can run the launch; th, tw > 0 fix the RECT tile shape. */
int iiseg_conv_c8_force_tiling(int kind, int th, int tw);
int iiseg_conv_c8(void* stream, const iiseg_conv_desc* d, const void* x1, const void* x2,
                  const uint8_t* mask_in, const void* wp16, const float* bias, const void* add,
                  int add_kind, void* out, int out_kind, void* pool_out, uint8_t* mask_out);
/* iiseg_conv_c8 with x1 a CHANNEL SLICE of a wider C8 tensor: x1 points at the slice's first chunk plane of
 * image 0, x1_ctot = channels per image of the tensor it lies in (0: dense, d->C1) -- the conv reads a dense
 * block's stack in place (models/FCDenseNet.py:92: the ConcatLayer costs no copy) -- and, with
 * IISEG_CONV_ZINS, zero-inserted (d->H, d->W = the zero-inserted size, d->pad = 0).  Plain bf16 launches only
 * (no IISEG_CONV_UNPOOL / IISEG_CONV_X3, d->C2 == 0 with IISEG_CONV_ZINS). */
int iiseg_conv_c8_slice(void* stream, const iiseg_conv_desc* d, const void* x1, int32_t x1_ctot,
                        const void* x2, const uint8_t* mask_in, const void* wp16, const float* bias,
                        const void* add, int add_kind, void* out, int out_kind, void* pool_out,
                        uint8_t* mask_out);
/* DePool2D materialised on C8 tensors (layers/mylayers.py:88-115) for the pooled-coordinate window
 * (y0, x0, wh, ww): out (BC8, H, W, 8) bf16 <- up (BC8, H/2, W/2, 8) where the mask byte's bit (y & 1) * 2 +
 * (x & 1) is set, else 0; elements outside the window are left as they are (rows / columns >= 2 (H/2), 2 (W/2)
 * must be zero in `out`: they are never written).  The decoder levels with >= 1024 input channels unpool this way
 * and convolve the result as a plain layer (LDS-DMA staging) instead of selecting through registers per tile. */
int iiseg_unpool_c8(void* stream, const void* up, const uint8_t* mask, void* out, int BC8, int H, int W,
                    int y0, int x0, int wh, int ww);
/* The same layer for AT MOST 16 OUTPUT CHANNELS (csrc/conv_c8_m16.hip: v_mfma_f32_16x16x32_bf16, M = 16
 * channels, 16-pixel blocks, 256-pixel th x tw tiles, three workgroups per CU): the DAE's class-score
 * layer (models/fcn_up.py:83-86) and FC-DenseNet's growth-rate-16 dense-block layers
 * (models/FCDenseNet.py:61-146).  Single source (d->C2 == 0), d->C1 = the channels convolved (a multiple
 * of 16): the FIRST d->C1 channels of x1, a C8 tensor of in_ctot >= d->C1 channels per image (0: d->C1).
 *   mask_in     with IISEG_CONV_UNPOOL, as in iiseg_conv_c8
 *   bn_a, bn_b  both NULL, or d->C1 floats each: BatchNorm + ReLU applied to the input while it is staged,
 *               x <- max(bn_a[c] x + bn_b[c], 0) rounded to bf16 (iiseg_bn_fold_f32 makes the pair from
 *               beta, gamma, mean, inv_std); the zero-padding ring stays zero
 *   out         out_kind 1: C8 bf16, the 16 channels [out_c0, out_c0 + 16) of a (B, out_ctot / 8, out_H,
 *               out_W, 8) tensor (out_ctot 0: a dense 16-channel tensor), channels past Cout zero;
 *               out_kind 3: fp32 NCHW (B, out_ctot or Cout, out_H, out_W)
 * weights: iiseg_conv_halo_bf16_pack for this Cout (rows padded to 32).  No add, no pool.
 * iiseg_bn_stats_c8: batch statistics (mean, 1 / sqrt(biased var + eps); P10) of channels [c0, c0 + n) of
 * a C8 tensor of Ctot channels (all multiples of 8), two deterministic stages in double; workspace:
 * iiseg_bn_stats_c8_workspace_elems(n) doubles, caller-owned. */
int iiseg_conv_c8_m16_supported(const iiseg_conv_desc* d);
int iiseg_conv_c8_m16(void* stream, const iiseg_conv_desc* d, const void* x1, int in_ctot,
                      const uint8_t* mask_in, const float* bn_a, const float* bn_b, const void* wp16,
                      const float* bias, void* out, int out_kind);
/* The same with a caller-owned scratch buffer for SPLIT-K launches: on small maps (one tile per image, 40+
 * k-tiles in sequence: the deep dense blocks of FC-DenseNet at 7^2 .. 28^2) the channel range is dealt to
 * 16 / tiles-per-image workgroups per tile, each sums its share from zero into an fp32 slab, and a second
 * launch (one workgroup per tile) adds the slabs in slice order and writes the result -- a fixed association
 * that depends on the layer geometry only, never on the batch.  workspace: iiseg_conv_c8_m16_workspace_bytes(d)
 * bytes (it also holds the per-tile statistics below), 16-byte aligned.  Without a workspace the launch is
 * not split (iiseg_conv_c8_m16). */
int64_t iiseg_conv_c8_m16_workspace_bytes(const iiseg_conv_desc* d);
/* Two more fusions of a dense-block layer (models/FCDenseNet.py:88-92) ride on the same entry point:
 *   stat_mean / stat_inv_std (bf16 C8 output only, needs the workspace): the batch statistics (biased variance,
 *     stat_eps; P10) of the 16 produced channels -- of the stored bf16 values, per-tile sums in double out of the
 *     conv's epilogue, added in a fixed order by a one-workgroup reduction -- written to entries [out_c0,
 *     out_c0 + 16) of the two vectors (iiseg_bn_stats_c8's result up to the order of the double sums);
 *   fold_* (with the statistics): that reduction then also forms the (a, b) pair of the NEXT consumer's
 *     BatchNorm over the first fold_n channels of the stack from the two vectors (iiseg_bn_fold_f32's
 *     arithmetic, one launch less per layer). */
int iiseg_conv_c8_m16_ws(void* stream, const iiseg_conv_desc* d, const void* x1, int in_ctot,
                         const uint8_t* mask_in, const float* bn_a, const float* bn_b, const void* wp16,
                         const float* bias, void* out, int out_kind, void* workspace,
                         int64_t workspace_bytes, float* stat_mean, float* stat_inv_std, double stat_eps,
                         const float* fold_beta, const float* fold_gamma, float* fold_a, float* fold_b,
                         int fold_n);
/* 1x1 convolution on a bf16 C8 tensor (csrc/conv1x1_c8.hip): FC-DenseNet's TransitionDown (BN -> ReLU -> 1x1
 * conv -> 2x2 max-pool, FC_DenseNet.layers.TransitionDown at models/FCDenseNet.py:95) and the SoftmaxLayer's
 * 1x1 score convolution (models/FCDenseNet.py:134) on the dense block's stack.
 *   x        (B, in_ctot / 8, H, W, 8) bf16; the layer reads its first Cin channels (Cin % 16 == 0)
 *   bn_a/b   NULL, or the folded BatchNorm of those channels (iiseg_bn_fold_f32): x <- max(a x + b, 0)
 *   wp       iiseg_conv1x1_c8_pack of w[Cout][Cin_w] (iiseg_conv1x1_c8_weight_bytes bytes)
 *   pool 1:  out = bf16 C8 (B, out_ctot / 8, H / 2, W / 2, 8); channels [out_c0, out_c0 + Cout) receive
 *            maxpool2x2(conv) + bias (ignore_border), a partial last chunk is zero-padded
 *   pool 0:  out = fp32 NCHW (B, Cout, H, W) = conv + bias (out_ctot, out_c0 ignored) */
int64_t iiseg_conv1x1_c8_weight_bytes(int Cout, int Cin);
int iiseg_conv1x1_c8_pack(void* stream, const float* w, int64_t stride_o, int64_t stride_c, int Cout,
                          int Cin_w, int Cin, void* wp);
int iiseg_conv1x1_c8(void* stream, const void* x, int B, int Cin, int in_ctot, int H, int W,
                     const float* bn_a, const float* bn_b, const void* wp, const float* bias, int Cout,
                     int pool, void* out, int out_ctot, int out_c0);
int iiseg_bn_fold_f32(void* stream, const float* beta, const float* gamma, const float* mean,
                      const float* inv_std, float* a, float* b, int n);
int64_t iiseg_bn_stats_c8_workspace_elems(int n);
int iiseg_bn_stats_c8(void* stream, const void* x, int B, int Ctot, int c0, int n, int H, int W,
                      double eps, float* mean, float* inv_std, double* workspace);
int iiseg_nchw_to_c8(void* stream, const float* x, void* out, int B, int C, int H, int W, int C8n);
int iiseg_c8_to_nchw(void* stream, const void* x, float* out, int B, int C, int H, int W, int C8n);
/* The same between a dense fp32 (B, C, H, W) tensor and the chunk planes [c8_0, c8_0 + ceil(C / 8)) of a
 * WIDER C8 tensor of C8tot planes per image (a dense block's stack: models/FCDenseNet.py:92,119-127). */
int iiseg_nchw_to_c8_slice(void* stream, const float* x, void* out, int B, int C, int H, int W, int C8tot,
                           int c8_0);
int iiseg_c8_slice_to_nchw(void* stream, const void* x, float* out, int B, int C, int H, int W, int C8tot,
                           int c8_0);
int iiseg_pool_mask_c8(void* stream, const void* pre, int pre_f32, void* pooled, uint8_t* mask,
                       int BC8, int PH, int PW, int py0, int px0, int H, int W, int y0, int x0, int wh,
                       int ww);
/* The helpers of the IISEG_CONV_X3 mode: fp32 NCHW <-> hi / lo pair (B, 2 C8n, H, W, 8), and the
 * pool of a C8 fp32 piece `pre` (B, C8n, PH, PW, 8) into a pooled hi / lo pair (B, 2 C8n, H/2, W/2, 8)
 * + the (B, C8n, H/2, W/2, 8) mask bytes (same comparisons as iiseg_pool_mask_c8 with pre_f32). */
/* iiseg_conv_c8_split_weights: the filter of an IISEG_CONV_X3 layer, w[co][c][3][3] (element strides
 * stride_o / stride_c as in iiseg_conv_pack_f32) -> out (Cout, 2 Cp, 3, 3) floats = [W_hi | W_lo] with
 * Cp = Cin rounded up to 16, ready for iiseg_conv_halo_bf16_pack with C1 = 2 Cp. */
int iiseg_conv_c8_split_weights(void* stream, const float* w, int64_t stride_o, int64_t stride_c,
                                int Cout, int Cin, float* out);
int iiseg_nchw_to_c8x3(void* stream, const float* x, void* out, int B, int C, int H, int W, int C8n);
int iiseg_c8x3_to_nchw(void* stream, const void* x, float* out, int B, int C, int H, int W, int C8n);
int iiseg_pool_mask_c8x3(void* stream, const void* pre, void* pooled, uint8_t* mask, int B, int C8n,
                         int PH, int PW, int py0, int px0, int H, int W, int y0, int x0, int wh,
                         int ww);

/* im2col + split-K GEMM form of iiseg_conv_f32 for 'valid' (pad 0, dil 1) KxK layers computed in
 * full into a dense output (FCN-8's fc6, models/fcn8.py:75-76): same packed weights `wp` (d->Kpad,
 * d->Mpad from iiseg_conv_plan), the gather-free MFMA GEMM kernel of the Winograd path, partial
 * sums of the K slices added in a fixed order.  workspace: iiseg_conv_gemm_workspace_elems floats;
 * stages: IISEG_WINO_INPUT (im2col) | IISEG_WINO_GEMM | IISEG_WINO_OUTPUT, IISEG_WINO_ALL normally. */
int iiseg_conv_gemm_supported(const iiseg_conv_desc* d);
int64_t iiseg_conv_gemm_workspace_elems(const iiseg_conv_desc* d);
int iiseg_conv_gemm_f32(void* stream, const iiseg_conv_desc* d, const float* x, const float* wp,
                        const float* bias, float* workspace, float* out, uint32_t stages);

/* ---------------------------------------------------------------------------------------
 * 2x2/2 max-pool, ignore_border (floor).  Replaces Pool2DLayer(x, 2): models/fcn8.py:38-72,
 * models/fcn_down.py:122.   x (B,C,H,W) -> out (B,C,H/2,W/2)
 * ------------------------------------------------------------------------------------- */
int iiseg_maxpool2x2_f32(void* stream, const float* x, float* out, int32_t BC, int32_t H,
                         int32_t W);

/* The same for pooled outputs [y0,y0+wh) x [x0,x0+ww) only, written in place into the (H/2, W/2)
 * planes of `out` (loop-invariant borders of the refinement loop are not recomputed). */
int iiseg_maxpool2x2_window_f32(void* stream, const float* x, float* out, int32_t BC, int32_t H,
                                int32_t W, int32_t y0, int32_t x0, int32_t wh, int32_t ww);
int iiseg_maxpool2x2_window_f64(void* stream, const double* x, double* out, int32_t BC, int32_t H,
                                int32_t W, int32_t y0, int32_t x0, int32_t wh, int32_t ww);

/* Equality-mask unpool, materialised (the fused form is IISEG_CONV_UNPOOL).  Replaces
 * DePool2D.get_output_for, layers/mylayers.py:88-115:
 * out[y,x] = (y < 2h && x < 2w && pre[y,x] == pooled[y/2,x/2]) ? up[y/2,x/2] : 0 */
int iiseg_unpool_eqmask_f32(void* stream, const float* up, const float* pre,
                            const float* pooled, float* out, int32_t BC, int32_t H, int32_t W);
/* The same, restricted to the window [y0,y0+wh) x [x0,x0+ww) of the (H,W) planes (in place:
 * elements outside the window are not written). */
int iiseg_unpool_eqmask_window_f32(void* stream, const float* up, const float* pre,
                                   const float* pooled, float* out, int32_t BC, int32_t H,
                                   int32_t W, int32_t y0, int32_t x0, int32_t wh, int32_t ww);

/* ---------------------------------------------------------------------------------------
 * Small-channel transposed convolution (gather form), crop='valid', linear.
 * Replaces Deconv2DLayer (Theano CorrMM_gradInputs): models/fcn8.py:90,100,109;
 * models/fcn_up.py:41-45.  w is the reference layout W[Cin][Cout][K][K]; the kernel applies
 * the spatial flip of Lasagne's filter_flip=True gradient form.
 * Only the window [oy0,oy0+OH) x [ox0,ox0+OW) of the (H-1)*s+K output is produced.
 * Optional: out += add[b, c, ay0+oy, ax0+ox] (add is (B,Cout,AH,AW)); Cout <= 32.
 * ------------------------------------------------------------------------------------- */
typedef struct iiseg_deconv_desc {
    int32_t B, Cin, H, W, Cout, K, stride;
    int32_t oy0, ox0, OH, OW;
    int32_t AH, AW, ay0, ax0;
} iiseg_deconv_desc;

int iiseg_deconv_f32(void* stream, const iiseg_deconv_desc* d, const float* x, const float* w,
                     const float* bias, const float* add, float* out);

/* The same layer by output phase (csrc/deconv_phase.hip): K = 2 * stride, stride 2 or 8, at most 16 channels
 * on either side -- the three FCN-8 upsamplers (models/fcn8.py:90,100,109).  A thread owns an input position
 * and a row phase; the weights are scalar operands.  Same sums in the same order as iiseg_deconv_*.
 * iiseg_deconv_phase_supported: 1 when the request (with / without a skip tensor, f32 / f64) has this form;
 * iiseg_deconv_phase_weight_elems: elements of the packed weight buffer [py][dy][dx][Cin][px][Cout padded];
 * iiseg_deconv_phase_pack_*: W[Cin][Cout][K][K] -> that buffer (once per layer);
 * iiseg_deconv_phase_*: the launch, arguments as iiseg_deconv_* with the packed weights for w. */
int iiseg_deconv_phase_supported(const iiseg_deconv_desc* d, int has_add, int is_f64);
int64_t iiseg_deconv_phase_weight_elems(const iiseg_deconv_desc* d);
int iiseg_deconv_phase_pack_f32(void* stream, const iiseg_deconv_desc* d, const float* w, float* wp);
int iiseg_deconv_phase_pack_f64(void* stream, const iiseg_deconv_desc* d, const double* w, double* wp);
int iiseg_deconv_phase_f32(void* stream, const iiseg_deconv_desc* d, const float* x, const float* wp,
                           const float* bias, const float* add, float* out);
int iiseg_deconv_phase_f64(void* stream, const iiseg_deconv_desc* d, const double* x, const double* wp,
                           const double* bias, const double* add, double* out);

/* ---------------------------------------------------------------------------------------
 * Channel softmax of a center-cropped score map.  Replaces the crop + dimshuffle + reshape +
 * softmax + reshape + dimshuffle tail of models/fcn8.py:115-130,187-191 and
 * models/fcn_up.py:104-113,154-169.   score (B,C,SH,SW) -> out (B,C,H,W), window at (sy0,sx0),
 * C <= 32.  If `minuend` (B,C,H,W) is non-NULL the result is minuend - softmax, which is
 * de_fn's  -(pred_dae - y_hat)  of iterative_inference.py:203-204.
 * ------------------------------------------------------------------------------------- */
int iiseg_crop_softmax_f32(void* stream, const float* score, const float* minuend, float* out,
                           int32_t B, int32_t C, int32_t SH, int32_t SW, int32_t sy0,
                           int32_t sx0, int32_t H, int32_t W);

/* ---------------------------------------------------------------------------------------
 * Fused refinement step.  Replaces, per image, iterative_inference.py:203-204 (de = y - r),
 * :270 (y - step*de), :273 (clip) and the per-pixel part of :275 (||de||_2 over channels):
 *   r = softmax_c(score[window]); de = y - r; if active[b]: y = clip(y - step*de, 0, 1)
 *   partial[b][blk] = sum over the block's pixels of ||de||_2
 * y (B,C,H,W) is updated in place; active (B) int32; partial (B, nblk) float64 where
 * nblk = iiseg_refine_partials(H, W).
 * iiseg_refine_finalize then applies :275-277 per image: norm = sum(partial[b])/(H*W);
 * if active[b]: iters[b] += 1; if norm < eps: active[b] = 0   (update first, then test).
 * ------------------------------------------------------------------------------------- */
int iiseg_refine_partials(int32_t H, int32_t W);
int iiseg_refine_update_f32(void* stream, const float* score, float* y, const int32_t* active,
                            double* partial, int32_t B, int32_t C, int32_t SH, int32_t SW,
                            int32_t sy0, int32_t sx0, int32_t H, int32_t W, float step);
/* iiseg_refine_update_f32 that also writes the updated map as a bf16 C8 tensor y8 (B, C8n, H, W, 8)
 * -- the input format of the DAE's first layer under mma='bf16c8' (iiseg_conv_c8): the conversion
 * pass per refinement step disappears. */
int iiseg_refine_update_c8_f32(void* stream, const float* score, float* y, const int32_t* active,
                               double* partial, void* y8, int32_t C8n, int32_t B, int32_t C,
                               int32_t SH, int32_t SW, int32_t sy0, int32_t sx0, int32_t H, int32_t W,
                               float step);
int iiseg_refine_finalize(void* stream, const double* partial, int32_t* active, int32_t* iters,
                          double* last_norm, int32_t B, int32_t nblk, int32_t HW, double eps);

/* ---------------------------------------------------------------------------------------
 * The context module's last layers and the refinement update as ONE launch.  Replaces
 * models/contextmod_dae.py:98-105 (dilconv6: 3x3 'valid' + ReLU; dilconv7: 1x1 linear; softmax) followed
 * by iterative_inference.py:203-204, 270-277 -- i.e. iiseg_conv_f32 (3x3) + iiseg_conv_f32 (1x1) +
 * iiseg_refine_update_f32 -- for C <= 16 channels on either side (csrc/conv_small.hip):
 *   t = relu(conv3x3_valid(x; wp6, b6)); score = conv1x1(t; wp7, b7); then the refinement step above on
 *   y (B,C,H,W), in place, with the updated y of active images ALSO stored into channels
 *   [cat_c0, cat_c0 + C) of ycat (B, cat_ctot, cat_H, cat_W) at (cat_y0, cat_x0) (the ConcatLayer((h, y))
 *   buffer of contextmod_dae.py:55-59 the next step's first layer reads; NULL: no mirror).
 * x (B, C, H + 2, W + 2); wp6 / wp7: Wp[K][Mpad] of iiseg_conv_pack_f32 for the two layers; partial
 * (B, nblk) with nblk = iiseg_ctx_tail_partials(H, W) -- pass that nblk to iiseg_refine_finalize.
 * y is bit-identical to the three separate launches; the partials are summed per 16 x 64 tile.
 * ------------------------------------------------------------------------------------- */
int iiseg_ctx_tail_partials(int32_t H, int32_t W);
int iiseg_ctx_tail_f32(void* stream, const float* x, const float* wp6, int32_t Mpad6, const float* b6,
                       const float* wp7, int32_t Mpad7, const float* b7, float* y, const int32_t* active,
                       double* partial, float* ycat, int32_t cat_ctot, int32_t cat_c0, int32_t cat_H,
                       int32_t cat_W, int32_t cat_y0, int32_t cat_x0, int32_t B, int32_t C, int32_t H,
                       int32_t W, float step);

/* ---------------------------------------------------------------------------------------
 * Metrics accumulator.  Replaces val_fn, iterative_inference.py:206-210 = metrics.py:11-37
 * (jaccard), :40-65 (accuracy), :144-156 (squared_error, int-void branch):
 *   cm[i*(C+1) + j] += #(argmax_c y == i  and  argmax_c t == j), i < C, j <= C (j == C: void)
 *   sums[0] += sum_px mask * mean_c (y - t[:C])^2 ;  sums[1] += sum_px mask, mask = sum_c t[:C]
 * y (B,C,H,W), t one-hot (B,C+1,H,W) with void last.  cm: C*(C+1) int64, sums: 2 float64,
 * both ACCUMULATED into (zero them first); C <= 31.
 * ------------------------------------------------------------------------------------- */
int iiseg_confusion_f32(void* stream, const float* y, const float* t, int64_t* cm, double* sums,
                        int32_t B, int32_t C, int32_t HW);

/* Same accumulation restricted to images with active[b] != 0.  Used for the per-iteration Jaccard
 * of the validation driver (iterative_inference_valid.py:231,280-288: `valid_mat[:, :, it] +=
 * jacc_iter` only for images still iterating). */
int iiseg_confusion_masked_f32(void* stream, const float* y, const float* t, const int32_t* active,
                               int64_t* cm, double* sums, int32_t B, int32_t C, int32_t HW);
int iiseg_confusion_masked_f64(void* stream, const double* y, const double* t,
                               const int32_t* active, int64_t* cm, double* sums, int32_t B,
                               int32_t C, int32_t HW);

/* Non-finite input detection: *counter += number of NaN / Inf elements of x[0, n) (device counter, no
 * synchronisation).  The fp32 / bf16 kernels do not promise to propagate a NaN that enters the network (their
 * ReLU / max-pool epilogues are bare v_max_f32: build.py EXTRA_FLAGS), where the reference's Theano ops do
 * (models/fcn_down.py:102-104 `rectify`); the product detects non-finite values where they enter instead:
 * weights when a layer is built, image batches here. */
int iiseg_count_nonfinite_f32(void* stream, const float* x, int64_t n, int32_t* counter);
int iiseg_count_nonfinite_f64(void* stream, const double* x, int64_t n, int32_t* counter);

/* ---------------------------------------------------------------------------------------
 * True-gradient mode (BASELINE north_star; NOT in the reference, whose
 * "gradient" is the residual r - y):  E(y) = sum_{c,px} (r(y|h) - y)^2,
 * dE/dy = J_r^T 2(r - y) - 2(r - y).  The convolutions' backward-data passes run on the forward
 * conv entry points with channel-transposed, spatially flipped weights; these are the other
 * adjoints.  All tensors dense NCHW.
 *   sqerr_softmax_bwd : gscore = softmax backward of g_r = 2 (softmax(score window) - y)
 *   depool_bwd        : adjoint of DePool2D w.r.t. its input: masked 2x2 sum (mask pre == pooled)
 *   pool_relu_bwd     : adjoint of maxpool2x2(relu(z)) w.r.t. z given pre = relu(z): the gradient
 *                       goes to every position equal to the window maximum (F4 mask), relu'(0) = 0
 *   grad_update       : grad = gthrough - 2 (r - y); y <- clip(y - step * grad, 0, 1); norm
 *                       partials of grad as in iiseg_refine_update_*
 * ------------------------------------------------------------------------------------- */
int iiseg_sqerr_softmax_bwd_f32(void* stream, const float* score, const float* y, float* gscore,
                                int32_t B, int32_t C, int32_t SH, int32_t SW, int32_t sy0,
                                int32_t sx0, int32_t H, int32_t W);
int iiseg_sqerr_softmax_bwd_f64(void* stream, const double* score, const double* y, double* gscore,
                                int32_t B, int32_t C, int32_t SH, int32_t SW, int32_t sy0,
                                int32_t sx0, int32_t H, int32_t W);
int iiseg_depool_bwd_f32(void* stream, const float* gout, const float* pre, const float* pooled,
                         float* gup, int32_t BC, int32_t H, int32_t W);
int iiseg_depool_bwd_f64(void* stream, const double* gout, const double* pre, const double* pooled,
                         double* gup, int32_t BC, int32_t H, int32_t W);
int iiseg_pool_relu_bwd_f32(void* stream, const float* gpool, const float* pre, const float* pooled,
                            float* gz, int32_t BC, int32_t H, int32_t W);
int iiseg_pool_relu_bwd_f64(void* stream, const double* gpool, const double* pre,
                            const double* pooled, double* gz, int32_t BC, int32_t H, int32_t W);
int iiseg_grad_update_f32(void* stream, const float* score, const float* gthrough, float* y,
                          const int32_t* active, double* partial, int32_t B, int32_t C, int32_t SH,
                          int32_t SW, int32_t sy0, int32_t sx0, int32_t H, int32_t W, float step);
int iiseg_grad_update_f64(void* stream, const double* score, const double* gthrough, double* y,
                          const int32_t* active, double* partial, int32_t B, int32_t C, int32_t SH,
                          int32_t SW, int32_t sy0, int32_t sx0, int32_t H, int32_t W, double step);

/* ---------------------------------------------------------------------------------------
 * Element-wise helpers of the optional noise>0 mask emulation: with
 * dae_dict['noise'] > 0 the reference's DePool2D masks come from a re-evaluation of the down path
 * with GaussianNoiseLayer / DropoutLayer ACTIVE (layers/mylayers.py:91-93, models/fcn_down.py:60-63,
 * 108-111).  The caller supplies the random tensors.
 *   add_noise     : out = x + sigma * eps                       (GaussianNoiseLayer)
 *   dropout_apply : x = x * keep / (1 - p), keep in {0, 1}       (DropoutLayer, rescale=True)
 * ------------------------------------------------------------------------------------- */
int iiseg_add_noise_f32(void* stream, const float* x, const float* eps, float sigma, float* out,
                        int64_t n);
int iiseg_add_noise_f64(void* stream, const double* x, const double* eps, double sigma, double* out,
                        int64_t n);
int iiseg_dropout_apply_f32(void* stream, float* x, const float* keep, float p, int64_t n);
int iiseg_dropout_apply_f64(void* stream, double* x, const double* keep, double p, int64_t n);

/* ---------------------------------------------------------------------------------------
 * Batch-statistics BatchNorm (+ReLU).  Replaces lasagne BatchNormLayer under
 * batch_norm_use_averages=False (iterative_inference.py:187) followed by the rectify
 * NonlinearityLayer of FC-DenseNet's BN_ReLU_Conv (models/FCDenseNet.py:12,90,109,123).
 * x is (B, >=C, H, W) with `bstride` elements between images (a channel-slice view of a stack).
 *   stats : mean[c], inv_std[c] = 1/sqrt(biased_var + eps) over (B,H,W) for c < C
 *   apply : out[b,c] = max((x[b,c] - mean[c]) * (gamma[c] * inv_std[c]) + beta[c], 0), out dense
 * ------------------------------------------------------------------------------------- */
int64_t iiseg_bn_stats_workspace_elems(int32_t C); /* doubles of caller-owned workspace */
int iiseg_bn_stats_f32(void* stream, const float* x, int64_t bstride, int32_t B, int32_t C,
                       int32_t HW, float eps, float* mean, float* inv_std, double* workspace);
int iiseg_bn_relu_f32(void* stream, const float* x, int64_t bstride, int32_t B, int32_t C,
                      int32_t HW, const float* beta, const float* gamma, const float* mean,
                      const float* inv_std, float* out);
/* Stored-average BatchNorm of the DAE's bn=1 layers (lasagne BatchNormLayer at deterministic=True:
 * models/fcn_down.py:112-114, models/fcn_up.py:91-93 under iterative_inference.py:189), in place on
 * the window [y0,y0+wh) x [x0,x0+ww) of the (H,W) planes of x (B,C,H,W):
 *   x = (x - mean[c]) * (gamma[c] * inv_std[c]) + beta[c] */
int iiseg_bn_affine_window_f32(void* stream, float* x, int32_t B, int32_t C, int32_t H, int32_t W,
                               int32_t y0, int32_t x0, int32_t wh, int32_t ww, const float* beta,
                               const float* gamma, const float* mean, const float* inv_std);
int iiseg_bn_affine_window_f64(void* stream, double* x, int32_t B, int32_t C, int32_t H, int32_t W,
                               int32_t y0, int32_t x0, int32_t wh, int32_t ww, const double* beta,
                               const double* gamma, const double* mean, const double* inv_std);
int iiseg_bn_stats_f64(void* stream, const double* x, int64_t bstride, int32_t B, int32_t C,
                       int32_t HW, double eps, double* mean, double* inv_std, double* workspace);
int iiseg_bn_relu_f64(void* stream, const double* x, int64_t bstride, int32_t B, int32_t C,
                      int32_t HW, const double* beta, const double* gamma, const double* mean,
                      const double* inv_std, double* out);

/* ---------------------------------------------------------------------------------------
 * float64 variants (strict-parity mode).  The reference's CPU path computes in float64 (Theano
 * floatX default) and DePool2D compares activations for exact equality, so only
 * float64 arithmetic reproduces its mask decisions at near-tied pooling windows.
 * Same semantics and argument meaning as the _f32 entry points; tensors are float64.
 * iiseg_conv_*_f64 cover 1x1 and 3x3 filters (v_mfma_f64_16x16x4_f64); other filter shapes go
 * through iiseg_im2col_f64 + a 1x1 convolution (fc6 of models/fcn8.py:75-76), whose weight matrix
 * is the reference W[out][in*KH*KW] as it lies in memory.  No gather table is needed.
 * ------------------------------------------------------------------------------------- */
int iiseg_conv_plan_f64(iiseg_conv_desc* d);
int iiseg_conv_pack_f64(void* stream, const iiseg_conv_desc* d, const double* w, int64_t stride_o,
                        int64_t stride_c, double* wp);
int iiseg_conv_f64(void* stream, const iiseg_conv_desc* d, const double* x1, const double* x2,
                   const double* pre, const double* pooled, const double* wp, const double* bias,
                   const double* add, double* out);
/* iiseg_conv_f64 with the 2x2 max-pool (Pool2DLayer(x, 2), ignore_border) of the result fused into the epilogue
 * of the halo-tile kernel: pool_out = the FULL pooled tensor (B, Cout, fullH / 2, fullW / 2), written where the
 * computed window has whole pooling windows (as iiseg_conv_pool_f32; out is still stored: DePool2D compares the
 * float64 pre-pool map). */
int iiseg_conv_pool_f64_supported(const iiseg_conv_desc* d);
int iiseg_conv_pool_f64(void* stream, const iiseg_conv_desc* d, const double* x1, const double* x2,
                        const double* pre, const double* pooled, const double* wp, const double* bias,
                        const double* add, double* out, double* pool_out);
/* ... and with the DePool2D masks as bytes (iiseg_conv_mask_f32's convention: bit (y & 1) * 2 + (x & 1) of the
 * window's byte = pre == pooled, here compared in FLOAT64 by the encoder layer's epilogue, so the decisions are
 * those of the pre / pooled form): mask_out (uint8, pool_out's shape) is written next to the pool and `out`
 * may then be NULL -- the float64 pre-pool map is never stored; mask_in (uint8, x1's shape) replaces pre /
 * pooled in the DePool2D staging of the decoder layer. */
int iiseg_conv_mask_f64_supported(const iiseg_conv_desc* d);
int iiseg_conv_mask_f64(void* stream, const iiseg_conv_desc* d, const double* x1, const double* x2,
                        const double* pre, const double* pooled, const uint8_t* mask_in, const double* wp,
                        const double* bias, const double* add, double* out, double* pool_out,
                        uint8_t* mask_out);
/* 1 when iiseg_conv_f64 runs this (planned) request on the halo-tile kernel (conv_halo_f64.hip: plain
 * 3x3 layers, patch staged once per 4 input channels, 16-row MFMA tiles for Cout <= 16) instead of the
 * static-tap kernel; same packed weights, bit-identical results -- a scheduling fact, for profiles. */
int iiseg_conv_halo_f64_supported(const iiseg_conv_desc* d);
/* Split-K GEMM form of iiseg_conv_f64 for deep 1x1 layers computed in full into a dense output (fc6 after
 * iiseg_im2col_f64, fc7, score_fr: models/fcn8.py:75-85): x (B, C1, H, W) is laid out once as V[k][pixel],
 * then the LDS-DMA GEMM kernel of the float64 Winograd path runs on the same packed weights `wp`
 * (iiseg_conv_plan_f64 / _pack_f64); the partial sums of the K slices are added in a fixed order, and the
 * number of slices does not depend on the batch.  workspace: iiseg_conv_gemm_f64_workspace_elems doubles. */
int iiseg_conv_gemm_f64_supported(const iiseg_conv_desc* d);
int64_t iiseg_conv_gemm_f64_workspace_elems(const iiseg_conv_desc* d);
int iiseg_conv_gemm_f64(void* stream, const iiseg_conv_desc* d, const double* x, const double* wp,
                        const double* bias, double* workspace, double* out);
/* x (B,C,H,W) -> out (B, C*KH*KW, H-KH+1, W-KW+1), channel index c*KH*KW + ky*KW + kx */
int iiseg_im2col_f64(void* stream, const double* x, double* out, int32_t B, int32_t C, int32_t H,
                     int32_t W, int32_t KH, int32_t KW);
int iiseg_maxpool2x2_f64(void* stream, const double* x, double* out, int32_t BC, int32_t H,
                         int32_t W);
int iiseg_unpool_eqmask_f64(void* stream, const double* up, const double* pre,
                            const double* pooled, double* out, int32_t BC, int32_t H, int32_t W);
int iiseg_unpool_eqmask_window_f64(void* stream, const double* up, const double* pre,
                                   const double* pooled, double* out, int32_t BC, int32_t H,
                                   int32_t W, int32_t y0, int32_t x0, int32_t wh, int32_t ww);
int iiseg_deconv_f64(void* stream, const iiseg_deconv_desc* d, const double* x, const double* w,
                     const double* bias, const double* add, double* out);
int iiseg_crop_softmax_f64(void* stream, const double* score, const double* minuend, double* out,
                           int32_t B, int32_t C, int32_t SH, int32_t SW, int32_t sy0,
                           int32_t sx0, int32_t H, int32_t W);
int iiseg_refine_update_f64(void* stream, const double* score, double* y, const int32_t* active,
                            double* partial, int32_t B, int32_t C, int32_t SH, int32_t SW,
                            int32_t sy0, int32_t sx0, int32_t H, int32_t W, double step);
int iiseg_confusion_f64(void* stream, const double* y, const double* t, int64_t* cm, double* sums,
                        int32_t B, int32_t C, int32_t HW);

#ifdef __cplusplus
}
#endif
#endif /* IISEG_H */
