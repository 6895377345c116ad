"""GPU parity of the float64 (strict-parity) path: the reference's CPU numerics (floatX=float64,
SURVEY P15).  With float64 arithmetic the HIP path takes the same DePool2D mask decisions as the
oracle, so the 1e-4 bound of north_star holds END TO END at full size (measured: ~1e-12)."""
import numpy as np
import pytest
import torch

from oracle import dae as odae, fcn8 as ofcn8, nn as onn, refine as orefine, metrics as ometrics
from iterative_inference_segm_amd import synthetic as S

pytestmark = pytest.mark.gpu
F64 = torch.float64


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def to64(p):
    return {k: tuple(np.asarray(a, dtype=np.float64) for a in v) for k, v in p.items()}


@pytest.fixture(scope='module')
def ops(built_lib):
    from iterative_inference_segm_amd import ops as _ops
    return _ops


CASES = [(2, 3, 17, 19, 11, 3, 1, 1, False), (1, 11, 20, 20, 64, 3, 5, 1, True),
         (3, 40, 13, 9, 130, 3, 1, 1, True), (2, 70, 7, 7, 33, 1, 0, 1, True),
         (1, 11, 40, 36, 11, 3, 0, 4, False), (2, 16, 9, 9, 200, 7, 0, 1, True)]


@pytest.mark.parametrize('case', CASES)
def test_conv_f64(ops, case):
    B, Cin, H, W, Cout, k, pad, dil, relu = case
    rng = np.random.default_rng(sum(case))
    x, Wt, b = rng.standard_normal((B, Cin, H, W)), rng.standard_normal((Cout, Cin, k, k)), \
        rng.standard_normal(Cout)
    ref = onn.conv2d(x, Wt, b, pad=pad, dilation=dil, relu=relu)
    got = host(ops.Conv(Wt, b, pad=pad, relu=relu, dil=dil, dtype=F64)(dev(x)))
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-12 * (1 + np.abs(ref).max())


def test_conv_f64_concat_unpool_window_add(ops):
    rng = np.random.default_rng(9)
    h, t = rng.standard_normal((2, 24, 10, 11)), rng.standard_normal((2, 8, 10, 11))
    Wt, b = rng.standard_normal((40, 32, 3, 3)), rng.standard_normal(40)
    ref = onn.conv2d(onn.concat_h_first(h, t), Wt, b, pad=1, relu=True)
    got = host(ops.Conv(Wt, b, pad=1, relu=True, dtype=F64)(dev(h), x2=dev(t)))
    assert np.abs(got - ref).max() <= 1e-12 * (1 + np.abs(ref).max())
    pre = np.maximum(rng.standard_normal((2, 70, 13, 13)), 0)
    pooled = onn.maxpool2(pre)
    up = rng.standard_normal(pooled.shape)
    Wt, b = rng.standard_normal((12, 70, 3, 3)), rng.standard_normal(12)
    other = rng.standard_normal((2, 12, 15, 14))
    full = onn.conv2d(onn.depool_eqmask(up, pre, pooled), Wt, b, pad=1)
    ref = onn.crop_sum(full[:, :, 1:12, 2:12], onn.center_crop(other, 11, 10))
    got = host(ops.Conv(Wt, b, pad=1, relu=False, dtype=F64)(
        dev(up), pre=dev(pre), pooled=dev(pooled), add=dev(other), add_off=(2, 2),
        window=(1, 2, 11, 10)))
    assert np.abs(got - ref).max() <= 1e-12 * (1 + np.abs(ref).max())
    assert np.array_equal(host(ops.maxpool2x2(dev(pre))), pooled)
    assert np.array_equal(host(ops.unpool_eqmask(dev(up), dev(pre), dev(pooled))),
                          onn.depool_eqmask(up, pre, pooled))


# the halo-tile float64 kernel (conv_halo_f64.hip: plain 3x3 layers): exact on integer data -- every
# product and partial sum is an integer below 2^53, so whatever the tile, window or MFMA row count the
# result must EQUAL the oracle's
HALO64 = [  # B, C1, C2, H, W, Cout, pad, unpool, window, relu
    (2, 5, 0, 45, 70, 20, 1, False, None, True),            # 64-row tiles, ragged tile edges, k padding
    (1, 64, 0, 23, 37, 11, 1, False, None, False),          # 16-row tiles
    (2, 8, 4, 19, 33, 64, 1, False, (3, 2, 13, 30), True),  # two sources, window
    (1, 12, 0, 21, 35, 11, 1, True, None, False),           # DePool2D input, odd map (no window for the last row / column)
    (2, 9, 0, 18, 40, 70, 1, True, (1, 5, 16, 33), True),   # DePool2D input, 64-row tiles, window
    (1, 3, 0, 30, 66, 16, 0, False, None, True),            # pad 0
    (1, 11, 0, 12, 12, 64, 5, False, None, True),           # pad > 1 (pad-100 layers)
]


@pytest.mark.parametrize('case', HALO64)
def test_conv_halo_f64_exact_on_integers(ops, case):
    B, C1, C2, H, W, Cout, pad, unpool, window, relu = case
    rng = np.random.default_rng(sum(x for x in case[:7]))
    Wt = rng.integers(-2, 3, size=(Cout, C1 + C2, 3, 3)).astype(np.float64)
    b = rng.integers(-4, 5, size=Cout).astype(np.float64)
    conv = ops.Conv(Wt, b, pad=pad, relu=relu, dtype=F64)
    kw = {}
    if unpool:
        pre = np.maximum(rng.integers(-3, 4, size=(B, C1, H, W)), 0).astype(np.float64)
        pooled = onn.maxpool2(pre)
        up = rng.integers(-3, 4, size=pooled.shape).astype(np.float64)
        xin = onn.depool_eqmask(up, pre, pooled)
        x1 = dev(up)
        kw.update(pre=dev(pre), pooled=dev(pooled))
    else:
        xin = rng.integers(-3, 4, size=(B, C1 + C2, H, W)).astype(np.float64)
        x1 = dev(xin[:, :C1])
        if C2:
            kw['x2'] = dev(xin[:, C1:])
    ref = onn.conv2d(xin, Wt, b, pad=pad, relu=False)
    other = rng.integers(-5, 6, size=ref.shape).astype(np.float64)
    ref = ref + other
    if relu:
        ref = np.maximum(ref, 0)
    if window is not None:
        y0, x0, oh, ow = window
        ref = ref[:, :, y0:y0 + oh, x0:x0 + ow]
        kw.update(window=window, add_off=(y0, x0))
    got = host(conv(x1, add=dev(other), **kw))
    assert got.shape == ref.shape and np.array_equal(got, ref)
    # placement into a larger map + channel slice leave everything else untouched
    if window is None and not unpool:
        big = torch.full((B, Cout + 3, ref.shape[2] + 4, ref.shape[3] + 5), 7.0, dtype=F64, device='cuda')
        conv(x1, add=dev(other), out=big, out_c0=2, place=(1, 3), **kw)
        hb = host(big)
        assert np.array_equal(hb[:, 2:2 + Cout, 1:1 + ref.shape[2], 3:3 + ref.shape[3]], ref)
        hb[:, 2:2 + Cout, 1:1 + ref.shape[2], 3:3 + ref.shape[3]] = 7.0
        assert np.all(hb == 7.0)


def test_conv_halo_f64_is_the_form_of_the_thin_layers(ops):
    """iiseg_conv_halo_f64_supported: plain 3x3 layers yes; dilated, transposed and 1x1 layers no."""
    import ctypes as C
    rng = np.random.default_rng(0)
    for k, dil, want in ((3, 1, 1), (3, 2, 0), (1, 1, 0)):
        conv = ops.Conv(rng.standard_normal((8, 6, k, k)), None, pad=0, relu=False, dil=dil, dtype=F64)
        d, _, _ = conv._plan(1, 6, 0, 20, 20, None, None, False)
        assert conv.lib.iiseg_conv_halo_f64_supported(C.byref(d)) == want


@pytest.mark.parametrize('case', [(3, 1100, 5, 6, 70, 1), (2, 24, 9, 10, 130, 7), (1, 2048, 7, 7, 11, 1)])
def test_conv_gemm_f64_exact_on_integers(ops, case):
    """Deep 1x1 layers / im2col'd KxK layers as split-K GEMMs (iiseg_conv_gemm_f64): exact on integer data,
    whatever the number of K slices."""
    import ctypes as C
    B, Cin, H, W, Cout, k = case
    rng = np.random.default_rng(sum(case))
    x = rng.integers(-3, 4, size=(B, Cin, H, W)).astype(np.float64)
    Wt = rng.integers(-2, 3, size=(Cout, Cin, k, k)).astype(np.float64)
    b = rng.integers(-4, 5, size=Cout).astype(np.float64)
    conv = ops.Conv(Wt, b, pad=0, relu=True, dtype=F64)
    d, _, _ = conv._plan(B, Cin, 0, H, W, None, None, False)
    assert conv.lib.iiseg_conv_gemm_f64_supported(C.byref(d)) == 1
    got = host(conv(dev(x)))
    assert np.array_equal(got, onn.conv2d(x, Wt, b, pad=0, relu=True))


@pytest.mark.parametrize('case', [(2, 12, 19, 37, 40, 1, None), (1, 64, 30, 70, 128, 1, (2, 4, 20, 58)),
                                  (2, 11, 13, 33, 64, 5, None)])
def test_conv_halo_f64_fused_pool(ops, case):
    """The 2x2 max-pool (ignore_border) in the float64 halo kernel's epilogue: the pooled tensor EQUALS the pool of
    the stored pre-pool map (which is still written: DePool2D compares it), on whole windows of the launch window;
    everything outside stays untouched."""
    B, Cin, H, W, Cout, pad, window = case
    rng = np.random.default_rng(sum(case[:6]))
    x, Wt, b = rng.standard_normal((B, Cin, H, W)), rng.standard_normal((Cout, Cin, 3, 3)), rng.standard_normal(Cout)
    conv = ops.Conv(Wt, b, pad=pad, relu=True, dtype=F64)
    assert conv.pool_fusable()
    fh, fw = conv.out_hw(H, W)
    pool = torch.full((B, Cout, fh // 2, fw // 2), -7.0, dtype=F64, device='cuda')
    kw = {}
    if window is not None:
        kw['window'] = conv.pool_window(H, W, window)
        assert kw['window'] is not None
    out = host(conv(dev(x), pool_out=pool, **kw))
    ref = onn.conv2d(x, Wt, b, pad=pad, relu=True)
    got_pool = host(pool)
    if window is None:
        assert np.abs(out - ref).max() <= 1e-12 * (1 + np.abs(ref).max())
        assert np.array_equal(got_pool, onn.maxpool2(out))
    else:
        y0, x0, oh, ow = kw['window']
        assert np.abs(out - ref[:, :, y0:y0 + oh, x0:x0 + ow]).max() <= 1e-12 * (1 + np.abs(ref).max())
        want = np.full_like(got_pool, -7.0)
        pw = onn.maxpool2(out)
        want[:, :, y0 // 2:y0 // 2 + pw.shape[2], x0 // 2:x0 // 2 + pw.shape[3]] = pw
        assert np.array_equal(got_pool, want)


def _eq_bits(pre, pooled):
    """bit (y & 1) * 2 + (x & 1) of the window's byte: pre == pooled (include/iiseg.h iiseg_conv_mask_f32)."""
    B, C, H, W = pre.shape
    h2, w2 = pooled.shape[2:]
    m = np.zeros(pooled.shape, np.uint8)
    for dy in range(2):
        for dx in range(2):
            m |= ((pre[:, :, dy:2 * h2:2, dx:2 * w2:2] == pooled).astype(np.uint8) << (dy * 2 + dx))
    return m


@pytest.mark.parametrize('case', [(2, 12, 21, 37, 40, 1), (1, 64, 30, 66, 128, 1)])
def test_conv_halo_f64_mask_bytes_encoder(ops, case):
    """float64 encoder side of the byte-mask DePool2D: conv + pool + mask_out with the pre-pool map NOT stored
    gives the pool of the stored-map launch and exactly the bytes of pre == pooled (float64 comparisons), also on
    a window."""
    B, Cin, H, W, Cout, pad = case
    rng = np.random.default_rng(sum(case))
    x, Wt, b = rng.standard_normal((B, Cin, H, W)), rng.standard_normal((Cout, Cin, 3, 3)), rng.standard_normal(Cout)
    conv = ops.Conv(Wt, b, pad=pad, relu=True, dtype=F64)
    assert conv.pool_fusable() and conv.mask_ok()
    fh, fw = conv.out_hw(H, W)
    pooled = torch.empty((B, Cout, fh // 2, fw // 2), dtype=F64, device='cuda')
    full = conv(dev(x), pool_out=pooled)
    ref_mask = _eq_bits(host(full), host(pooled))
    assert int((ref_mask == 15).sum()) > 0                     # ReLU zeros: windows with every bit set
    pool2 = torch.full_like(pooled, -3.0)
    mask = torch.full(pooled.shape, 0xAA, dtype=torch.uint8, device='cuda')
    assert conv(dev(x), pool_out=pool2, mask_out=mask, store_out=False) is None
    assert np.array_equal(host(pool2), host(pooled)) and np.array_equal(mask.cpu().numpy(), ref_mask)
    y0, x0, h, ww = win = conv.pool_window(H, W, (3, 5, 8, 22))
    pool2.fill_(-3.0); mask.fill_(0xAA)
    conv(dev(x), window=win, out=torch.empty(full.shape, dtype=F64, device='meta'), place=(y0, x0), pool_out=pool2,
         mask_out=mask, store_out=False)
    exp_p = np.full(pooled.shape, -3.0); exp_m = np.full(pooled.shape, 0xAA, np.uint8)
    q = (slice(None), slice(None), slice(y0 // 2, (y0 + h) // 2), slice(x0 // 2, (x0 + ww) // 2))
    exp_p[q], exp_m[q] = host(pooled)[q], ref_mask[q]
    assert np.array_equal(host(pool2), exp_p) and np.array_equal(mask.cpu().numpy(), exp_m)


@pytest.mark.parametrize('case', [(2, 24, 20, 26, 40), (1, 64, 33, 41, 11), (2, 16, 14, 15, 70)])
def test_conv_halo_f64_unpool_from_mask_bytes_is_bit_identical(ops, case):
    """float64 decoder side: the conv over the DePool2D of `up` from mask bytes == from pre / pooled, bit for bit
    (64-row and 16-row tiles, windows with the skip add, ties and all-equal windows)."""
    B, Cc, H, W, Cout = case
    rng = np.random.default_rng(sum(case))
    pre = np.maximum(rng.standard_normal((B, Cc, H, W)), 0)
    pooled = onn.maxpool2(pre)
    up = rng.standard_normal(pooled.shape)
    conv = ops.Conv(rng.standard_normal((Cout, Cc, 3, 3)) * 0.3, rng.standard_normal(Cout), pad=1, relu=False,
                    dtype=F64)
    assert conv.mask_ok()
    mask = torch.from_numpy(_eq_bits(pre, pooled)).cuda()
    ref = conv(dev(up), pre=dev(pre), pooled=dev(pooled))
    got = conv(dev(up), mask_in=mask, unpool_hw=(H, W))
    assert np.array_equal(host(got), host(ref))
    add = rng.standard_normal((B, Cout, H + 3, W + 2))
    for (y0, x0, h, ww) in [(1, 2, 9, 11), (0, 0, H, 5), (H - 4, W - 7, 4, 7)]:
        kw = dict(window=(y0, x0, h, ww), add=dev(add), add_off=(y0 + 1, x0))
        ref = conv(dev(up), pre=dev(pre), pooled=dev(pooled), **kw)
        got = conv(dev(up), mask_in=mask, unpool_hw=(H, W), **kw)
        assert np.array_equal(host(got), host(ref)), (y0, x0, h, ww)


def test_tail_deconv_metrics_f64(ops):
    rng = np.random.default_rng(10)
    x, Wt, b = rng.standard_normal((2, 11, 5, 6)), rng.standard_normal((11, 11, 16, 16)), \
        rng.standard_normal(11)
    ref = onn.deconv2d(x, Wt, b, stride=8)
    assert np.abs(host(ops.Deconv(Wt, b, 8, dtype=F64)(dev(x))) - ref).max() <= 1e-12 * 50
    score = 5 * rng.standard_normal((2, 11, 30, 28))
    y0 = rng.random((2, 11, 24, 20))
    r = onn.softmax_channels(onn.center_crop(score, 24, 20))
    assert np.abs(host(ops.crop_softmax(dev(score), 24, 20)) - r).max() <= 1e-14
    y = dev(y0)
    st = ops.RefineState(2, 24, 20, y.device)
    ops.refine_update(dev(score), y, st, 0.3)
    ops.refine_finalize(st, 1e-3)
    de = y0 - r
    assert np.abs(host(y) - np.clip(y0 - 0.3 * de, 0, 1)).max() <= 1e-14
    assert np.abs(host(st.last_norm) - np.linalg.norm(de, axis=1).mean(axis=(1, 2))).max() <= 1e-12
    labels = rng.integers(0, 12, size=(2, 24, 20))
    t = np.zeros((2, 12, 24, 20)); np.put_along_axis(t, labels[:, None], 1.0, axis=1)
    from iterative_inference_segm_amd.api import Metrics
    m = Metrics(11, 'cuda')
    ops.confusion_accumulate(dev(y0), dev(t), m.cm, m.sums)
    acc, jacc, mse = m.result()
    acc_r, jacc_r, mse_r = ometrics.val_fn(y0, t, 11, [11])
    assert np.array_equal(jacc, jacc_r) and abs(acc - acc_r) < 1e-12 and abs(mse - mse_r) < 1e-12


def test_full_size_end_to_end_strict(built_lib):
    """BASELINE config-1/2 network, one 224x224 image, ALL 10 refinement steps of the bench
    workload, float64 on the GPU vs the float64 oracle, free-running: identical mask decisions,
    refined map within 1e-4 (north_star) -- in fact ~1e-11.  This is the path (and bench.py's
    `strict_f64` leg the number) that carries the end-to-end 1e-4 claim."""
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp, dp = S.make_fcn8_params(), S.make_dae_params()
    ii = IterativeInference(FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'], dtype=F64),
                            StandardDAE(dp, 11, dtype=F64), 11, [11], dtype=F64)
    X = S.make_images(1, 224, 224, seed=1234)
    out = ii.pred_fcn_fn(X)
    H, Y = out[:-1], out[-1]
    h_ref, y_ref = ofcn8.fcn8_forward(to64(fp), X.astype(np.float64), layer=['pool4', 'probs_dimshuffle'])
    assert np.abs(host(Y) - y_ref).max() <= 1e-10
    dp64 = to64(dp)
    dae_fn = lambda hh, yy: odae.dae_forward(dp64, hh, yy)
    yii_ref, it_ref = orefine.refine_batch(dae_fn, [h_ref], y_ref, 0.1, 10)
    Yii, iters, _ = ii.refine(H, Y, 0.1, 10)
    err = np.abs(host(Yii) - yii_ref).max()
    print('float64 end-to-end max-abs err after 10 steps: %.3e' % err)
    assert list(host(iters)) == list(it_ref) == [10]
    assert err <= 1e-4


def test_reference_function_is_chaotic_at_float32_resolution(built_lib):
    """Why no float32 implementation can promise 1e-4 free-running through DePool2D with these
    (synthetic, non-contractive) weights: the reference's OWN function -- float64 arithmetic
    throughout, here the oracle-pinned float64 HIP path -- is run on y0 and on y0 * (1 + 1e-6 u),
    |u| <= 1: a perturbation of about ten float32 ulps, below the rounding any float32 conv stack
    accumulates.  Equality-mask decisions at near-tied pooling windows (layers/mylayers.py:111-114)
    flip, every flip moves a skip-sized value by a pixel, and the loop amplifies it: after the 10
    steps of the bench workload the two float64 results differ by > 0.1 and most pixels are outside
    1e-4 (numbers printed; table for 1e-7..1e-5 and the fp32 path in profiles/r02_sensitivity.md)."""
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp, dp = S.make_fcn8_params(), S.make_dae_params()
    ii = IterativeInference(FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'], dtype=F64),
                            StandardDAE(dp, 11, dtype=F64), 11, [11], dtype=F64)
    X = S.make_images(2, 224, 224, seed=1234)
    out = ii.pred_fcn_fn(X)
    H, Y = out[:-1], out[-1]
    u = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, size=tuple(Y.shape))).cuda()
    Yp = (Y * (1 + 1e-6 * u)).clamp(0, 1)
    assert float((Y - Yp).abs().max()) <= 1e-6
    a = host(ii.refine(H, Y, 0.1, 10, early_stop=False)[0])
    b = host(ii.refine(H, Yp, 0.1, 10, early_stop=False)[0])
    e = np.abs(a - b)
    frac = float((e.max(axis=1) <= 1e-4).mean())
    print('float64 path, y0 vs y0 (1 + 1e-6 u) after 10 steps: max %.3e mean %.3e, pixels within '
          '1e-4 %.4f, argmax agreement %.5f'
          % (e.max(), e.mean(), frac, float((a.argmax(1) == b.argmax(1)).mean())))
    assert e.max() > 1e-2 and frac < 0.5


WINO64_CASES = [  # B, Cin, H, W, Cout, pad, relu, window, anchor
    (2, 128, 12, 14, 128, 1, True, None, (0, 0)),
    (1, 144, 9, 21, 70, 1, False, None, (0, 0)),            # Cout padded to the 64-channel tile
    (3, 256, 7, 7, 320, 1, True, None, (1, 1)),             # odd tile anchor
    (2, 128, 6, 6, 128, 5, True, None, (0, 0)),             # wide zero padding (pad-100 rule, scaled)
    (2, 160, 20, 18, 96, 1, True, (3, 5, 10, 9), (1, 1)),   # window, anchored at its parity
]


@pytest.mark.parametrize('case', WINO64_CASES)
def test_conv_wino_f64(ops, case, monkeypatch):
    """float64 Winograd F(2x2,3x3) (csrc/conv_wino_f64.hip) vs the oracle, incl. windows and tile
    anchors; and bit-identical between a windowed and the full-map launch with the same anchor."""
    B, Cin, H, W, Cout, pad, relu, window, anchor = case
    monkeypatch.setattr(ops, 'WINO_F64_MIN_COUT', 64)       # (the default sends Cout < 128 to the halo kernel)
    rng = np.random.default_rng(sum(case[:6]))
    x, Wt, b = rng.standard_normal((B, Cin, H, W)), rng.standard_normal((Cout, Cin, 3, 3)), \
        rng.standard_normal(Cout)
    ref = onn.conv2d(x, Wt, b, pad=pad, relu=relu)
    conv = ops.Conv(Wt, b, pad=pad, relu=relu, dtype=F64)
    assert conv.wino_f64
    full = host(conv(dev(x), anchor=anchor))
    assert np.abs(full - ref).max() <= 1e-12 * (1 + np.abs(ref).max())
    assert conv._U is not None                              # took the Winograd path
    if window is not None:
        y0, x0, h, w = window
        got = host(conv(dev(x), window=window, anchor=anchor))
        assert np.array_equal(got, full[:, :, y0:y0 + h, x0:x0 + w])


def test_conv_wino_f64_concat_add_placement_and_exact_ties(ops, monkeypatch):
    monkeypatch.setattr(ops, 'WINO_F64_MIN_COUT', 64)
    rng = np.random.default_rng(31)
    h, t = rng.standard_normal((2, 96, 10, 11)), rng.standard_normal((2, 64, 10, 11))
    Wt, b = rng.standard_normal((72, 160, 3, 3)), rng.standard_normal(72)
    other = rng.standard_normal((2, 72, 14, 15))
    ref = onn.conv2d(onn.concat_h_first(h, t), Wt, b, pad=1) + other[:, :, 2:12, 3:14]
    conv = ops.Conv(Wt, b, pad=1, relu=False, dtype=F64)
    y0, x0, hh, ww = 2, 4, 7, 6
    out = torch.full((2, 72, 10, 11), 5.0, dtype=F64, device='cuda')
    conv(dev(h), x2=dev(t), add=dev(other), add_off=(2 + y0, 3 + x0), window=(y0, x0, hh, ww), out=out,
         place=(y0, x0))
    want = np.full_like(ref, 5.0)
    want[:, :, y0:y0 + hh, x0:x0 + ww] = ref[:, :, y0:y0 + hh, x0:x0 + ww]
    assert np.abs(host(out) - want).max() <= 1e-12 * (1 + np.abs(ref).max())
    # equal patches -> bit-equal outputs: a map that is constant inside (the weights-only pad-100
    # border of the encoder maps) gives one value per channel over the whole interior, whatever the
    # position of a pixel inside its 2x2 Winograd tile -- DePool2D's exact ties depend on it
    xc = np.broadcast_to(rng.standard_normal((1, 160, 1, 1)), (1, 160, 16, 16)).copy()
    got = host(ops.Conv(Wt, b, pad=1, relu=True, dtype=F64)(dev(xc)))[:, :, 1:-1, 1:-1]
    assert np.array_equal(got, np.broadcast_to(got[:, :, :1, :1], got.shape))


@pytest.mark.parametrize('shape,window,anchor', [((2, 128, 13, 15), None, (0, 0)),
                                                 ((1, 144, 16, 12), (3, 2, 9, 8), (1, 0)),
                                                 ((2, 128, 9, 9), (0, 0, 9, 9), (1, 1))])
def test_conv_wino_f64_fused_unpool(ops, shape, window, anchor, monkeypatch):
    """DePool2D (layers/mylayers.py:88-115) applied inside the float64 Winograd input transform: vs the
    oracle, and bit-identical to the same Winograd layer run on the materialised unpooled map (odd
    sizes leave the last row / column of `pre` outside every pooling window -> zeros)."""
    B, Cc, H, W = shape
    monkeypatch.setattr(ops, 'WINO_F64_MIN_COUT', 64)
    rng = np.random.default_rng(H * W + Cc)
    pre = np.maximum(rng.standard_normal(shape), 0)          # ReLU zeros -> genuine ties
    pooled = onn.maxpool2(pre)
    up = rng.standard_normal(pooled.shape)
    Wt, b = rng.standard_normal((80, Cc, 3, 3)), rng.standard_normal(80)
    un = onn.depool_eqmask(up, pre, pooled)
    ref = onn.conv2d(un, Wt, b, pad=1, relu=True)
    conv = ops.Conv(Wt, b, pad=1, relu=True, dtype=F64)
    assert conv.wino_f64
    kw = dict(window=window) if window is not None else {}
    got = host(conv(dev(up), pre=dev(pre), pooled=dev(pooled), anchor=anchor, **kw))
    mat = host(conv(dev(un), anchor=anchor, **kw))
    assert conv._U is not None
    if window is not None:
        y0, x0, h, w = window
        ref = ref[:, :, y0:y0 + h, x0:x0 + w]
    assert np.abs(got - ref).max() <= 1e-12 * (1 + np.abs(ref).max())
    assert np.array_equal(got, mat)


_HALO_OFF_SCRIPT = r'''
import sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from iterative_inference_segm_amd import synthetic as S, ops
from iterative_inference_segm_amd.api import IterativeInference
from iterative_inference_segm_amd.dae import StandardDAE
from iterative_inference_segm_amd.fcn8 import FCN8
from oracle import dae as odae, fcn8 as ofcn8, refine as orefine
F64 = torch.float64
to64 = lambda p: {k: tuple(np.asarray(a, np.float64) for a in v) for k, v in p.items()}
fp = S.make_fcn8_params(width_div=16, fc_channels=64, seed=1)
dp = S.make_dae_params(h_channels=(fp['conv4_3'][0].shape[0],), n_filters=4, seed=2)
X = S.make_images(2, 64, 48, seed=3).astype(np.float64)
fcn = FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'], dtype=F64)
dae = StandardDAE(dp, 11, concat_h=['pool4'], n_filters=4, dtype=F64)
# with the halo-tile kernel off no float64 layer may promise the fused pool or the mask bytes
convs = [c for c in list(vars(dae).values()) + list(vars(fcn).values()) if isinstance(c, ops.Conv)]
for holder in (dae, fcn):
    for v in vars(holder).values():
        if isinstance(v, dict):
            convs += [c for c in v.values() if isinstance(c, ops.Conv)]
        if isinstance(v, (list, tuple)):
            convs += [c for c in v if isinstance(c, ops.Conv)]
assert convs, 'no Conv layers found'
assert not any(c.pool_fusable() or c.mask_ok() for c in convs if c.dtype == F64)
ii = IterativeInference(fcn, dae, 11, [11], dtype=F64)
out = ii.pred_fcn_fn(X)
Yii, iters, _ = ii.refine(out[:-1], out[-1], 0.1, 2)
torch.cuda.synchronize()
h_ref, y_ref = ofcn8.fcn8_forward(to64(fp), X, layer=['pool4', 'probs_dimshuffle'])
dp64 = to64(dp)
yii_ref, it_ref = orefine.refine_batch(lambda hh, yy: odae.dae_forward(dp64, hh, yy, n_filters=4),
                                       [h_ref], y_ref, 0.1, 2)
err = float(np.abs(Yii.cpu().numpy() - yii_ref).max())
assert err <= 1e-9, err
print('F64_HALO_OFF_OK %%.3g' %% err)
'''


def test_float64_path_with_only_the_halo_kernel_switched_off(built_lib):
    """IISEG_F64_HALO=0 ALONE (pool fusion and mask bytes left at their defaults) must fall
    back to the static-tap kernel with separate pool / pre-pooled masks -- not raise 'conv + pool fusion is not
    available' -- and still match the oracle.  The switch is read once per process: a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IISEG_F64_HALO='0')
    r = subprocess.run([sys.executable, '-c', _HALO_OFF_SCRIPT % {'root': root}], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'F64_HALO_OFF_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
