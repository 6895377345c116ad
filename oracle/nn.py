"""Oracle primitives: numpy restatement of the Lasagne/Theano layer semantics the hot path uses.

TEST INFRASTRUCTURE (see oracle/__init__.py).  All tensors are C-contiguous NCHW numpy arrays;
the dtype of the result is the dtype of `x` (float64 for the parity oracle, float32 for the
timed CPU baseline).  Pins Pn refer to SURVEY.md section 2.1.
"""
import numpy as np

from . import _cref


def conv2d(x, W, b=None, pad=0, dilation=1, relu=False):
    """Lasagne Conv2DLayer with flip_filters=False (P1): stride-1 cross-correlation.

    float64 inputs (the parity oracle) go through the plain-C fixed-order kernel of
    oracle/conv_ref.c, which gives bit-equal outputs for equal patches (the DePool2D equality
    masks need that, see the header of conv_ref.c); float32 inputs (the timed CPU baseline)
    go through `conv2d_blas`.  Same arguments as `conv2d_blas`.
    """
    if x.dtype == np.float64:
        return _cref.conv2d_f64(x, W, b, pad, dilation, relu)
    return conv2d_blas(x, W, b, pad, dilation, relu)


def conv2d_blas(x, W, b=None, pad=0, dilation=1, relu=False):
    """Stride-1 cross-correlation via one BLAS matmul per filter tap (im2col-GEMM class).

    W[out, in, kh, kw], b[out]; pad = int zero padding on both spatial axes ('same' == k//2,
    pad=100 literal, 'valid' == 0).  Default Lasagne nonlinearity is ReLU, callers pass
    relu=True for it (reference: every ConvLayer(...) without nonlinearity=, e.g.
    models/fcn8.py:34-85, models/fcn_down.py:102-104).  dilation>1 restates
    DilatedConv2DLayer's arithmetic (P11) for an already [out,in,kh,kw]-ordered W.

    Algorithm: one BLAS matmul per filter tap, accumulated tap by tap (the same
    im2col-GEMM class as Theano's CorrMM, without materialising the im2col matrix).
    """
    B, C, H, Wd = x.shape
    O, Ci, kh, kw = W.shape
    assert Ci == C, (Ci, C)
    # taps as contiguous [kh, kw, O, C] matrices so numpy's matmul takes the BLAS path
    Wt = np.ascontiguousarray(np.transpose(W.astype(x.dtype, copy=False), (2, 3, 0, 1)))
    if pad:
        xp = np.zeros((B, C, H + 2 * pad, Wd + 2 * pad), dtype=x.dtype)
        xp[:, :, pad:pad + H, pad:pad + Wd] = x
    else:
        xp = x
    Hp, Wp = xp.shape[2], xp.shape[3]
    OH = Hp - dilation * (kh - 1)
    OW = Wp - dilation * (kw - 1)
    assert OH > 0 and OW > 0
    out = np.empty((B, O, OH, OW), dtype=x.dtype)
    for n in range(B):
        acc = np.zeros((O, OH * OW), dtype=x.dtype)
        for ky in range(kh):
            for kx in range(kw):
                patch = xp[n, :, ky * dilation:ky * dilation + OH, kx * dilation:kx * dilation + OW]
                acc += Wt[ky, kx] @ np.ascontiguousarray(patch).reshape(C, OH * OW)
        if b is not None:
            acc += b.astype(x.dtype, copy=False)[:, None]
        if relu:
            np.maximum(acc, 0, out=acc)
        out[n] = acc.reshape(O, OH, OW)
    return out


def deconv2d(x, W, b=None, stride=1):
    """Lasagne Deconv2DLayer (TransposedConv2DLayer), crop='valid', linear (P3).

    W[in, out, kh, kw]; output spatial size (in-1)*stride + k.  Lasagne builds the layer as
    the gradient-wrt-inputs of a *true convolution* (filter_flip = not flip_filters, with
    flip_filters=False), which equals torch.conv_transpose2d with the kernel flipped on both
    spatial axes:  out[c, i*s+a, j*s+b] += x[o, i, j] * W[o, c, kh-1-a, kw-1-b].
    Used at models/fcn8.py:90,100,109 and models/fcn_up.py:41-45.
    (Restated from Lasagne@45bb568; not verifiable against the real library here.)
    """
    B, Ci, H, Wd = x.shape
    Ci2, O, kh, kw = W.shape
    assert Ci2 == Ci
    s = stride
    OH, OW = (H - 1) * s + kh, (Wd - 1) * s + kw
    # flipped taps as contiguous [kh, kw, O, Ci] matrices (BLAS path)
    Wf = np.ascontiguousarray(np.transpose(W[:, :, ::-1, ::-1].astype(x.dtype, copy=False),
                                           (2, 3, 1, 0)))
    out = np.zeros((B, O, OH, OW), dtype=x.dtype)
    for n in range(B):
        xn = x[n].reshape(Ci, H * Wd)
        for a in range(kh):
            for c in range(kw):
                contrib = (Wf[a, c] @ xn).reshape(O, H, Wd)
                out[n, :, a:a + (H - 1) * s + 1:s, c:c + (Wd - 1) * s + 1:s] += contrib
    if b is not None:
        out += b.astype(x.dtype, copy=False)[None, :, None, None]
    return out


def maxpool2(x):
    """Pool2DLayer(x, 2): 2x2 max, stride 2, pad 0, ignore_border=True (P4): out=floor(in/2).

    Reference: models/fcn8.py:38,45,54,63,72; models/fcn_down.py:122.
    """
    B, C, H, W = x.shape
    h, w = H // 2, W // 2
    v = x[:, :, :2 * h, :2 * w].reshape(B, C, h, 2, w, 2)
    return v.max(axis=(3, 5))


def depool_eqmask(up, pre, pooled):
    """DePool2D.get_output_for, layers/mylayers.py:88-115 (F4, P5).

    `up` (B,C,h,w) is repeated 2x on both axes (:95-98), pasted at the top-left of zeros of the
    pre-pool shape (:103-109), and multiplied by T.grad(None, wrt=pre, known_grads={pooled:
    ones}) (:111-114).  Theano's CPU MaxPoolGrad adds the upstream 1 to EVERY position of a
    window that equals the window maximum (all ties), and positions of a trailing odd row /
    column belong to no window (ignore_border=True) so their mask is 0.
    Deterministic masks only (noise == 0), see SURVEY.md F4.
    """
    B, C, H, W = pre.shape
    h, w = pooled.shape[2], pooled.shape[3]
    assert up.shape == pooled.shape, (up.shape, pooled.shape)
    assert (h, w) == (H // 2, W // 2)
    rep_up = np.repeat(np.repeat(up, 2, axis=2), 2, axis=3)
    rep_pool = np.repeat(np.repeat(pooled, 2, axis=2), 2, axis=3)
    out = np.zeros_like(pre)
    mask = pre[:, :, :2 * h, :2 * w] == rep_pool
    out[:, :, :2 * h, :2 * w] = np.where(mask, rep_up, 0).astype(pre.dtype, copy=False)
    return out


def center_crop(x, H, W):
    """Lasagne autocrop 'center' on the two spatial axes (P6): offset = (dim - target)//2."""
    oh = (x.shape[2] - H) // 2
    ow = (x.shape[3] - W) // 2
    assert oh >= 0 and ow >= 0
    return x[:, :, oh:oh + H, ow:ow + W]


def crop_sum(a, b):
    """ElemwiseSumLayer((a, b), cropping=[None, None, 'center', 'center']) (P6):
    both inputs are center-cropped to the per-axis minimum, then summed.
    Reference: models/fcn8.py:94-97,104-107; models/fcn_up.py:98-102."""
    H = min(a.shape[2], b.shape[2])
    W = min(a.shape[3], b.shape[3])
    return center_crop(a, H, W) + center_crop(b, H, W)


def crop_like(ref, x):
    """CroppingLayer / ElemwiseMergeLayer((ref, x), merge_function=lambda input, deconv: deconv,
    cropping center) (P6): returns x center-cropped to the per-axis minimum of both.
    Reference: layers/mylayers.py:49-57; models/fcn8.py:115-119; models/fcn_up.py:105-113."""
    H = min(ref.shape[2], x.shape[2])
    W = min(ref.shape[3], x.shape[3])
    return center_crop(x, H, W)


def softmax_channels(x):
    """lasagne.nonlinearities.softmax on the (B*H*W, C) view (P7): max-subtracted softmax over
    the channel axis, returned in NCHW.  Reference: models/fcn8.py:122-130,187-191;
    models/fcn_up.py:154-169."""
    m = x.max(axis=1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=1, keepdims=True)


def concat_h_first(h, x):
    """ConcatLayer((h, x), axis=1) -- h channels FIRST (P13).  models/model_helpers.py:93-94."""
    assert h.shape[0] == x.shape[0] and h.shape[2:] == x.shape[2:], (h.shape, x.shape)
    return np.concatenate([h.astype(x.dtype, copy=False), x], axis=1)


def batchnorm_batchstats(x, beta, gamma, eps=1e-4):
    """BatchNormLayer under batch_norm_use_averages=False (P10): batch mean and biased variance
    over (B,H,W), eps 1e-4: (x-mean)*gamma/sqrt(var+eps)+beta.  iterative_inference.py:187."""
    mean = x.mean(axis=(0, 2, 3), keepdims=True)
    var = x.var(axis=(0, 2, 3), keepdims=True)
    inv_std = 1.0 / np.sqrt(var + eps)
    return (x - mean) * (gamma[None, :, None, None] * inv_std) + beta[None, :, None, None]
