"""GPU: the 16-bit MFMA path (bf16 operands, fp32 accumulation; VERDICT row N1, BASELINE north_star
"fp16 MFMA peak" / configs[2] "bf16 with fp32 accumulate") against the float64 oracle.

Two kinds of checks:
  * EXACT: inputs and weights chosen so that every bf16 rounding is exact (small integers; weights
    multiples of 4 so that G g G^T stays integral).  Then the bf16 path must reproduce the oracle to
    fp32 rounding -- this pins every index of the k8-chunk layouts, the MFMA operand maps, the
    Winograd transforms and all epilogue / window / placement / concat / DePool2D fusions;
  * STATISTICAL: random data, error relative to the RMS of the reference output (8 significant
    bits per operand: expected ~2^-9 * sqrt(2) per product, averaged over K terms).
"""
import numpy as np
import pytest
import torch

from oracle import nn as onn

pytestmark = pytest.mark.gpu


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


@pytest.fixture(autouse=True)
def _one_kernel_per_layer(request, monkeypatch):
    """The tests named *wino* pin the Winograd form: with the per-geometry choice between the two
    16-bit forms (ops.BF16_PICKS / the static rule) a layer could run the direct kernel instead."""
    from iterative_inference_segm_amd import ops as _ops
    if 'wino' in request.node.name or 'form_choice' in request.node.name:
        # (by default layers below 256 channels go straight to the direct kernel)
        monkeypatch.setattr(_ops, 'BF16_WINO_MIN_CIN', 128)
        monkeypatch.setattr(_ops, 'BF16_WINO_MIN_COUT', 128)
    if 'wino' in request.node.name:
        monkeypatch.setattr(_ops, 'BF16_FORCE', 'wino')


@pytest.fixture(scope='module')
def ops(built_lib):
    from iterative_inference_segm_amd import ops as _ops
    return _ops


def ints(rng, *shape, lo=-3, hi=4, mult=1):
    return (rng.integers(lo, hi, size=shape) * mult).astype(np.float64)


def rel_rms(got, ref):
    return float(np.sqrt(((got - ref) ** 2).mean()) / (np.sqrt((ref ** 2).mean()) + 1e-30))


CASES = [  # B, Cin, H, W, Cout, pad, relu
    (2, 128, 12, 14, 128, 1, True),
    (1, 192, 9, 21, 160, 1, False),       # channels padded to the 64-channel k-tile, Cout to 128
    (3, 256, 7, 7, 512, 1, True),
    (1, 130, 10, 9, 130, 1, True),        # ragged channel counts
    (2, 128, 6, 6, 128, 5, True),         # wide zero padding (the pad-100 rule, scaled)
]


@pytest.mark.parametrize('case', CASES)
def test_wino_bf16_exact_on_integer_data(ops, case):
    B, Cin, H, W, Cout, pad, relu = case
    rng = np.random.default_rng(sum(case))
    x = ints(rng, B, Cin, H, W)
    Wt = ints(rng, Cout, Cin, 3, 3, lo=-1, hi=2, mult=4)      # G g G^T integral
    b = ints(rng, Cout)
    ref = onn.conv2d(x, Wt, b, pad=pad, relu=relu)
    conv = ops.Conv(Wt, b, pad=pad, relu=relu, mma='bf16')
    assert conv.wino_bf16
    got = host(conv(dev(x)))
    assert got.shape == ref.shape
    assert np.array_equal(got, ref.astype(np.float32)), np.abs(got - ref).max()


def test_wino_bf16_fusions_exact(ops):
    rng = np.random.default_rng(7)
    # two-source concat (h first), ragged split
    h, t = ints(rng, 2, 100, 10, 11), ints(rng, 2, 60, 10, 11)
    Wt, b = ints(rng, 128, 160, 3, 3, lo=-1, hi=2, mult=4), ints(rng, 128)
    ref = onn.conv2d(onn.concat_h_first(h, t), Wt, b, pad=1, relu=True)
    got = host(ops.Conv(Wt, b, pad=1, relu=True, mma='bf16')(dev(h), x2=dev(t)))
    assert np.array_equal(got, ref.astype(np.float32))
    # DePool2D input + skip-add with crop + window + placement, every patch-origin parity
    pre = np.maximum(ints(rng, 2, 128, 13, 15), 0)
    pooled = onn.maxpool2(pre)
    up = ints(rng, *pooled.shape)
    Wt, b = ints(rng, 128, 128, 3, 3, lo=-1, hi=2, mult=4), ints(rng, 128)
    other = ints(rng, 2, 128, 17, 16)
    full = onn.conv2d(onn.depool_eqmask(up, pre, pooled), Wt, b, pad=1)
    conv = ops.Conv(Wt, b, pad=1, relu=False, mma='bf16')
    for (oy, ox, oh, ow), anchor in [((1, 2, 11, 10), (0, 0)), ((2, 1, 9, 12), (1, 0)),
                                     ((0, 0, 13, 15), (0, 1)), ((3, 3, 8, 8), (1, 1))]:
        ref = full[:, :, oy:oy + oh, ox:ox + ow] + other[:, :, 2 + oy:2 + oy + oh, 1 + ox:1 + ox + ow]
        out = torch.full((2, 128, 20, 21), -9.0, device='cuda')
        conv(dev(up), pre=dev(pre), pooled=dev(pooled), add=dev(other), add_off=(2 + oy, 1 + ox),
             window=(oy, ox, oh, ow), out=out, place=(4, 5), anchor=anchor)
        o = host(out)
        assert np.array_equal(o[:, :, 4:4 + oh, 5:5 + ow], ref.astype(np.float32)), (oy, ox, anchor)
        o[:, :, 4:4 + oh, 5:5 + ow] = -9.0
        assert np.all(o == -9.0)                           # nothing outside the placement written
    # output channel slice of a wider tensor
    x = ints(rng, 1, 128, 8, 8)
    ref = onn.conv2d(x, Wt, b, pad=1)
    out = torch.full((1, 300, 8, 8), 5.0, device='cuda')
    ops.Conv(Wt, b, pad=1, relu=False, mma='bf16')(dev(x), out=out, out_c0=100)
    o = host(out)
    assert np.array_equal(o[:, 100:228], ref.astype(np.float32))
    assert np.all(o[:, :100] == 5.0) and np.all(o[:, 228:] == 5.0)


@pytest.mark.parametrize('shape', [(64, 1024, 12, 12, 2048), (8, 256, 60, 60, 128), (64, 512, 19, 19, 1024)])
def test_wino_bf16_statistical_at_layer_size(ops, shape):
    """Real layer sizes of configs[1] (conv6_1 window, up_conv3-like, conv5_1 y-half), random data:
    relative RMS error vs the float64 oracle of the SAME conv; and windows stay bit-identical."""
    B, Cin, H, W, Cout = shape
    rng = np.random.default_rng(Cin)
    x = rng.random((B, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    conv = ops.Conv(Wt, b, pad=1, relu=True, mma='bf16')
    xt = dev(x)
    got = host(conv(xt))
    nb = min(B, 2)
    ref = onn.conv2d(x[:nb].astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), pad=1, relu=True)
    err = rel_rms(got[:nb], ref)
    print('bf16 wino %s: relative RMS error %.2e, max abs %.2e' % (shape, err, np.abs(got[:nb] - ref).max()))
    assert err <= 6e-3
    # a window of the layer, same anchor: bit-identical to the full map (fixed-order sums per tile)
    win = (2, 4, H - 4, W - 6)
    part = host(conv(xt, window=win))
    assert np.array_equal(part, got[:, :, 2:H - 2, 4:W - 2])
    # and bit-identical across batch compositions
    assert np.array_equal(host(conv(xt[1:2].contiguous())), got[1:2])


HALO_CASES = [  # B, Cin, H, W, Cout, pad, relu
    (2, 3, 20, 37, 64, 4, True),          # FCN conv1_1-like: 3 channels zero-padded to the k-tile
    (1, 11, 33, 70, 64, 3, True),         # DAE conv1_1-like
    (2, 64, 17, 40, 64, 1, True),
    (1, 64, 30, 33, 128, 1, True),        # two 64-channel output tiles
    (2, 64, 25, 45, 11, 1, False),        # up_conv1-like: 11 output channels in a 32-row tile
    (1, 48, 19, 19, 16, 1, False),        # FC-DenseNet growth-rate-16 layer, ragged Cin
    (1, 100, 9, 9, 40, 1, True),
]


@pytest.mark.parametrize('case', HALO_CASES)
def test_halo_bf16_exact_on_integer_data(ops, case):
    B, Cin, H, W, Cout, pad, relu = case
    rng = np.random.default_rng(sum(case) + 1)
    x = ints(rng, B, Cin, H, W)
    Wt, b = ints(rng, Cout, Cin, 3, 3), ints(rng, Cout)
    ref = onn.conv2d(x, Wt, b, pad=pad, relu=relu)
    conv = ops.Conv(Wt, b, pad=pad, relu=relu, mma='bf16')
    assert conv.halo_bf16 and not conv.wino_bf16
    got = host(conv(dev(x)))
    assert got.shape == ref.shape
    assert np.array_equal(got, ref.astype(np.float32)), np.abs(got - ref).max()


def test_halo_bf16_fusions_exact(ops):
    rng = np.random.default_rng(17)
    # two-source concat (h first; C1 a multiple of the 16-channel k-tile)
    h, t = ints(rng, 2, 32, 21, 35), ints(rng, 2, 24, 21, 35)
    Wt, b = ints(rng, 64, 56, 3, 3), ints(rng, 64)
    ref = onn.conv2d(onn.concat_h_first(h, t), Wt, b, pad=1, relu=True)
    got = host(ops.Conv(Wt, b, pad=1, relu=True, mma='bf16')(dev(h), x2=dev(t)))
    assert np.array_equal(got, ref.astype(np.float32))
    # DePool2D input (odd trailing row / column) + skip-add with crop + window + placement
    pre = np.maximum(ints(rng, 2, 64, 27, 41), 0)
    pooled = onn.maxpool2(pre)
    up = ints(rng, *pooled.shape)
    Wt, b = ints(rng, 40, 64, 3, 3), ints(rng, 40)
    other = ints(rng, 2, 40, 31, 43)
    full = onn.conv2d(onn.depool_eqmask(up, pre, pooled), Wt, b, pad=1)
    conv = ops.Conv(Wt, b, pad=1, relu=False, mma='bf16')
    for (oy, ox, oh, ow) in [(1, 2, 23, 37), (0, 0, 27, 41), (5, 7, 9, 30)]:
        ref = full[:, :, oy:oy + oh, ox:ox + ow] + other[:, :, 2 + oy:2 + oy + oh, 1 + ox:1 + ox + ow]
        out = torch.full((2, 40, 35, 50), -9.0, device='cuda')
        conv(dev(up), pre=dev(pre), pooled=dev(pooled), add=dev(other), add_off=(2 + oy, 1 + ox),
             window=(oy, ox, oh, ow), out=out, place=(4, 5))
        o = host(out)
        assert np.array_equal(o[:, :, 4:4 + oh, 5:5 + ow], ref.astype(np.float32)), (oy, ox)
        o[:, :, 4:4 + oh, 5:5 + ow] = -9.0
        assert np.all(o == -9.0)
    # 2x2 max-pool in the epilogue: full map (odd sizes) and an even-origin window placed in place
    x = ints(rng, 2, 64, 23, 37)
    Wt, b = ints(rng, 64, 64, 3, 3), ints(rng, 64)
    conv = ops.Conv(Wt, b, pad=2, relu=True, mma='bf16')
    ref = onn.conv2d(x, Wt, b, pad=2, relu=True)            # (2, 64, 25, 39)
    pw = conv.pool_window(23, 37)
    assert pw == (0, 0, 25, 39)
    pool = torch.full((2, 64, 12, 19), -1.0, device='cuda')
    got = host(conv(dev(x), pool_out=pool))
    assert np.array_equal(got, ref.astype(np.float32))
    assert np.array_equal(host(pool), onn.maxpool2(ref).astype(np.float32))
    win = conv.pool_window(23, 37, (5, 9, 11, 13))
    assert win[0] % 2 == 0 and win[1] % 2 == 0
    full_out = torch.zeros((2, 64, 25, 39), device='cuda')
    pool2 = torch.full((2, 64, 12, 19), -1.0, device='cuda')
    conv(dev(x), window=win, out=full_out, place=(win[0], win[1]), pool_out=pool2)
    y0, x0, hh, ww = win
    assert np.array_equal(host(full_out)[:, :, y0:y0 + hh, x0:x0 + ww],
                          ref[:, :, y0:y0 + hh, x0:x0 + ww].astype(np.float32))
    p2, pr = host(pool2), onn.maxpool2(ref)
    ys, xs = slice(y0 // 2, (y0 + hh) // 2), slice(x0 // 2, (x0 + ww) // 2)
    assert np.array_equal(p2[:, :, ys, xs], pr[:, :, ys, xs].astype(np.float32))
    p2[:, :, ys, xs] = -1.0
    assert np.all(p2 == -1.0)


def test_bf16_dae_forward_statistical(built_lib):
    """The whole standard DAE of configs[1] (64 filters, pool4, 224x224) with bf16 MFMA operands
    against the float64 oracle on the SAME h, y: with the oracle's DePool2D masks injected (the
    discontinuity taken out) the reconstruction must agree to the bf16 error level; free (own masks)
    the argmax must still agree on almost every pixel."""
    from oracle import dae as odae
    from iterative_inference_segm_amd import ops, synthetic as S
    from iterative_inference_segm_amd.dae import StandardDAE
    from _parity_helpers import eq_masks, to64
    dp = S.make_dae_params()
    rng = np.random.default_rng(5)
    y = rng.random((1, 11, 224, 224)).astype(np.float32); y /= y.sum(1, keepdims=True)
    h = rng.random((1, 512, 26, 26)).astype(np.float32)
    r_ref, net = odae.dae_forward(to64(dp), [h.astype(np.float64)], y.astype(np.float64),
                                  return_net=True)
    dae = StandardDAE(dp, 11, mma='bf16')
    yt, ht = torch.from_numpy(y).cuda(), torch.from_numpy(h).cuda()
    free = host(dae(ht, yt))
    override = {}
    for p in range(1, 7):
        mo = eq_masks(net['pre%d' % p], net['pool%d' % p])
        full = np.zeros(net['pre%d' % p].shape, dtype=np.float32)
        full[:, :, :mo.shape[2], :mo.shape[3]] = mo
        override[p] = (torch.from_numpy(full).cuda(),
                       torch.ones(net['pool%d' % p].shape, dtype=torch.float32, device='cuda'))
    score = dae.scores([ht], yt, mask_override=override)
    forced = host(ops.crop_softmax(score, 224, 224, off=(0, 0)))
    e = np.abs(forced - r_ref)
    agree_f = float((forced.argmax(1) == r_ref.argmax(1)).mean())
    agree = float((free.argmax(1) == r_ref.argmax(1)).mean())
    print('bf16 DAE forward: teacher-forced max %.2e mean %.2e argmax %.5f | free-running mean %.2e '
          'argmax %.5f' % (e.max(), e.mean(), agree_f, np.abs(free - r_ref).mean(), agree))
    assert e.mean() <= 2e-3 and agree_f >= 0.99
    # own masks: 8-bit operands flip far more near-tied pooling windows than fp32 does (measured
    # 0.85 agreement after ONE forward); the loop's statistical criterion is the mIoU of north_star
    assert agree >= 0.8


def test_wino_bf16_split_form_exact(ops):
    """>= 1024 channels: the GEMMs run as their own kernel (256 x 128 or 128 x 128 blocks, fp32
    products M through the workspace) followed by the output-transform kernel.  Exact on integer
    data, incl. DePool2D input, skip-add, window and placement."""
    rng = np.random.default_rng(23)
    x = ints(rng, 2, 1024, 6, 7, lo=-1, hi=2)
    Wt, b = ints(rng, 256, 1024, 3, 3, lo=-1, hi=2, mult=4), ints(rng, 256)
    ref = onn.conv2d(x, Wt, b, pad=1, relu=True)
    got = host(ops.Conv(Wt, b, pad=1, relu=True, mma='bf16')(dev(x)))
    assert np.array_equal(got, ref.astype(np.float32))
    pre = np.maximum(ints(rng, 1, 1024, 9, 9, lo=-1, hi=2), 0)
    pooled = onn.maxpool2(pre)
    up = ints(rng, *pooled.shape, lo=-1, hi=2)
    Wt, b = ints(rng, 128, 1024, 3, 3, lo=-1, hi=2, mult=4), ints(rng, 128)
    other = ints(rng, 1, 128, 12, 12)
    full = onn.conv2d(onn.depool_eqmask(up, pre, pooled), Wt, b, pad=1)
    ref = full[:, :, 1:8, 2:9] + other[:, :, 3:10, 3:10]
    out = torch.full((1, 128, 10, 11), -9.0, device='cuda')
    ops.Conv(Wt, b, pad=1, relu=False, mma='bf16')(
        dev(up), pre=dev(pre), pooled=dev(pooled), add=dev(other), add_off=(3, 3),
        window=(1, 2, 7, 7), out=out, place=(2, 3), anchor=(1, 0))
    o = host(out)
    assert np.array_equal(o[:, :, 2:9, 3:10], ref.astype(np.float32))
    o[:, :, 2:9, 3:10] = -9.0
    assert np.all(o == -9.0)


@pytest.mark.parametrize('case', [(3, 40, 9, 11, 200, 7), (2, 1030, 5, 4, 130, 1), (4, 1024, 3, 3, 11, 1)])
def test_gemm_bf16_valid_layers_exact(ops, case):
    """fc6 (7x7 'valid'), fc7 / score_fr (deep 1x1) as im2col + bf16 GEMM; ragged K / Cout."""
    B, Cin, H, W, Cout, k = case
    rng = np.random.default_rng(sum(case))
    x = ints(rng, B, Cin, H, W, lo=-2, hi=3)
    Wt, b = ints(rng, Cout, Cin, k, k, lo=-1, hi=2), ints(rng, Cout)
    ref = onn.conv2d(x, Wt, b, pad=0, relu=True)
    conv = ops.Conv(Wt, b, pad=0, relu=True, mma='bf16')
    got = host(conv(dev(x)))
    assert got.shape == ref.shape
    assert np.array_equal(got, ref.astype(np.float32)), np.abs(got - ref).max()
    assert conv._W16 is not None                      # took the bf16 GEMM path


def test_bf16_mode_statistical_parity_64_images(built_lib):
    """Mode (i) of the 16-bit path on BASELINE configs[1] (FCN-8 + 64-filter DAE, 224x224, 10 steps)
    against the float64 path (= the reference's CPU numerics) over 64 images -- north_star's
    criterion for reduced precision: mIoU within +-0.05 of the reference (mIoU against the synthetic
    labels is a consistency metric: weights are random); plus the agreement of the FCN-8 output,
    where no feedback loop amplifies the operand rounding."""
    from iterative_inference_segm_amd import synthetic as S
    from iterative_inference_segm_amd.api import IterativeInference, Metrics
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp, dp = S.make_fcn8_params(), S.make_dae_params()

    def make(dtype, mma=None):
        return IterativeInference(
            FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'], dtype=dtype, mma=mma),
            StandardDAE(dp, 11, dtype=dtype, mma=mma), 11, [11], dtype=dtype)
    ii16, ii64 = make(torch.float32, 'bf16'), make(torch.float64)
    assert ii16.dae.enc['conv4_1'].wino_bf16 and ii16.dae.enc['conv2_1'].halo_bf16
    cm = {k: np.zeros((11, 12)) for k in ('bf16', 'f64')}
    agree_fcn, agree_ii = [], []
    for i in range(4):                                   # 4 batches of 16 = 64 images
        X = S.make_images(16, 224, 224, seed=500 + i)
        T = S.make_labels(16, 224, 224, seed=600 + i)
        res = {}
        for k, ii in (('bf16', ii16), ('f64', ii64)):
            out = ii.pred_fcn_fn(X)
            Yii = ii.refine(out[:-1], out[-1], 0.1, 10, early_stop=False)[0]
            m = ii.val_device(Yii, T)
            cm[k] += m.cm.cpu().numpy().reshape(11, 12)
            res[k] = (host(out[-1]), host(Yii))
        agree_fcn.append(float((res['bf16'][0].argmax(1) == res['f64'][0].argmax(1)).mean()))
        agree_ii.append(float((res['bf16'][1].argmax(1) == res['f64'][1].argmax(1)).mean()))
    miou = {}
    for k in cm:
        c = cm[k][:, :11]
        tp = np.diag(c)
        with np.errstate(invalid='ignore', divide='ignore'):
            miou[k] = float(np.nanmean(tp / (c.sum(1) + c.sum(0) - tp)))
    print('bf16 vs float64 over 64 images: mIoU %.5f vs %.5f, FCN argmax agreement %.4f, refined '
          '(10 chaotic steps) %.4f' % (miou['bf16'], miou['f64'], np.mean(agree_fcn), np.mean(agree_ii)))
    assert abs(miou['bf16'] - miou['f64']) <= 0.05
    assert np.mean(agree_fcn) >= 0.97


def test_config3_densenet_runs_in_bf16(built_lib):
    """BASELINE configs[2] as written ("bf16 with fp32 accumulate"): FC-DenseNet103 + standard DAE
    (padding 0, h = pool4 464 ch), 224x224, batch 32, 10 steps with bf16 MFMA operands.  The dense
    blocks' 3x3 growth-rate-16 convs run on the bf16 halo kernel, the DAE's wide layers on the bf16
    Winograd kernels.  Statistical check against the fp32 path on the segmentation output."""
    from iterative_inference_segm_amd import synthetic as S
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan
    params = S.make_densenet_params(layer_plan())
    dp = S.make_dae_params(h_channels=(464,))

    def make(mma):
        return IterativeInference(FCDenseNet(params, 11, layer=['pool4'], mma=mma),
                                  StandardDAE(dp, 11, padding=0, mma=mma), 11, [11])
    ii16, ii32 = make('bf16'), make(None)
    n16 = sum(1 for e in ii16.fcn.layers if e['conv'].halo_bf16)
    assert n16 >= 90                                       # the 3x3 BN_ReLU_Conv layers
    X = S.make_images(32, 224, 224, seed=303)
    o16, o32 = ii16.pred_fcn_fn(X), ii32.pred_fcn_fn(X)
    y16, y32 = host(o16[-1]), host(o32[-1])
    agree = float((y16.argmax(1) == y32.argmax(1)).mean())
    print('DenseNet103 bf16 vs fp32: argmax agreement %.4f, mean |dy| %.2e' % (agree, np.abs(y16 - y32).mean()))
    assert np.abs(y16.sum(1) - 1).max() <= 1e-5 and agree >= 0.9
    Yii, iters, norms = ii16.refine(o16[:-1], o16[-1], 0.1, 10, early_stop=False)
    a = host(Yii)
    assert list(host(iters)) == [10] * 32 and a.min() >= 0 and a.max() <= 1 and np.isfinite(a).all()


def test_bf16_form_choice_is_shared_and_batch_independent(ops, monkeypatch):
    """Where both 16-bit forms can run a layer the choice is a function of the launch geometry alone
    (an optional table ops.BF16_PICKS -- none is shipped -- else the static rule): two Conv objects of the same shape make
    the same choice (two engines, two ranks, two runs agree bit for bit), and so does the same layer
    on another batch size (an image alone == in a batch).  Nothing is timed at run time: the table
    is not written to.  Both forms are reachable (forced) and differ in their bits, i.e. the
    agreement above is not an accident of a single code path."""
    rng = np.random.default_rng(5)
    B, Cin, H, W, Cout = 6, 128, 30, 34, 128
    x = rng.random((B, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    before = dict(ops.BF16_PICKS)
    assert not ops.BF16_TUNE
    c1, c2 = (ops.Conv(Wt, b, pad=1, relu=True, mma='bf16') for _ in range(2))
    a = host(c1(dev(x)))
    assert ops.BF16_PICKS == before
    assert np.array_equal(host(c2(dev(x))), a)
    assert np.array_equal(host(c2(dev(x[2:3]))), a[2:3])
    ref = onn.conv2d(x[:1].astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), pad=1, relu=True)
    assert rel_rms(a[:1], ref) <= 6e-3
    forced = {}
    for form in ('wino', 'halo'):
        monkeypatch.setattr(ops, 'BF16_FORCE', form)
        forced[form] = host(ops.Conv(Wt, b, pad=1, relu=True, mma='bf16')(dev(x)))
        assert rel_rms(forced[form][:1], ref) <= 6e-3
    assert not np.array_equal(forced['wino'], forced['halo'])
    pick = before.get(ops._bf16_key(Cin, Cout, Cin, 0, False, H, W)) or ops._bf16_static_pick(Cin, W)
    assert np.array_equal(forced[pick], a)


def test_config3_densenet_on_c8_stacks(built_lib):
    """BASELINE configs[2] on the 16-bit format it names: FCDenseNet(mma='bf16c8') -- the dense blocks'
    stacks as bf16 C8 tensors, every BN_ReLU_Conv of a dense block one launch of the 16-row C8 kernel
    with BatchNorm + ReLU applied on the way in (models/FCDenseNet.py:61-146) -- + the DAE on C8, 224x224,
    batch 32, 10 steps.  Statistical check of the segmentation output and the h map against the fp32
    path (bf16 activations between ~100 batch-normalised layers: the same criterion as the 'bf16' mode's
    test above), and the loop runs on its h."""
    from iterative_inference_segm_amd import synthetic as S
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan
    params = S.make_densenet_params(layer_plan())
    dp = S.make_dae_params(h_channels=(464,))

    def make(mma):
        return IterativeInference(FCDenseNet(params, 11, layer=['pool4'], mma=mma),
                                  StandardDAE(dp, 11, padding=0, mma=mma), 11, [11])
    ii8, ii32 = make('bf16c8'), make(None)
    assert ii8.fcn.c8
    n8 = sum(1 for e in ii8.fcn.layers if e['conv'].c8)
    assert n8 == 92                                        # 91 dense-block layers + the first conv
    # ... and the other 11 convolutions on C8 kernels of their own: 5 TransitionDown + the score layer
    # (conv1x1_c8.hip), 5 TransitionUp (zero-inserted block on conv_c8_kernel)
    assert sum(1 for e in ii8.fcn.layers if 'conv8' in e) == 11
    X = S.make_images(32, 224, 224, seed=303)
    o8, o32 = ii8.pred_fcn_fn(X), ii32.pred_fcn_fn(X)
    y8, y32 = host(o8[-1]), host(o32[-1])
    h8, h32 = host(o8[0]), host(o32[0])
    assert h8.shape == (32, 464, 14, 14)
    agree = float((y8.argmax(1) == y32.argmax(1)).mean())
    eh = rel_rms(h8, h32)
    print('DenseNet103 on C8 stacks vs fp32: argmax agreement %.4f, mean |dy| %.2e, h rel rms %.2e'
          % (agree, np.abs(y8 - y32).mean(), eh))
    assert np.abs(y8.sum(1) - 1).max() <= 1e-5 and agree >= 0.9 and eh <= 0.1
    Yii, iters, norms = ii8.refine(o8[:-1], o8[-1], 0.1, 10, early_stop=False)
    a = host(Yii)
    assert list(host(iters)) == [10] * 32 and a.min() >= 0 and a.max() <= 1 and np.isfinite(a).all()


def test_config3_as_written_parity_on_the_damped_set(built_lib):
    """BASELINE configs[2] AS WRITTEN (FC-DenseNet103 + standard DAE, 224x224, batch 32, 10 steps, bf16 operands
    and bf16 C8 activations, fp32 accumulate) with north_star's number: labels = argmax of the float64-refined
    map (the float64 HIP path is pinned to the CPU oracle, tests/test_gpu_configs.py), so mIoU(float64) = 1 and
    "mIoU within +-0.05 of the CPU reference" reads |mIoU(bf16c8) - 1| <= 0.05.  On the DAMPED set of this config
    (synthetic.make_damped_densenet_set: contractive DAE loop, confident y0) -- on the chaotic default set only
    the smoke check of test_config3_densenet_on_c8_stacks (agreement >= 0.9) is meaningful.
    Measured (profiles/r05_parity_c3_32.md, same seeds): mIoU 0.951, refined argmax agreement 0.979 (the y0 of
    103 batch-normalised layers on bf16 activations agrees with float64 on 0.979 of the pixels, h rel rms
    3.5e-2; fp32 MFMA: 0.999998 / mIoU 1.000) -- the mIoU bound holds, an argmax agreement of 0.99 does NOT
    and is not claimed."""
    from iterative_inference_segm_amd import synthetic as S
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan
    params, dp = S.make_damped_densenet_set(layer_plan())

    def make(dtype, mma=None):
        return IterativeInference(FCDenseNet(params, 11, layer=['pool4'], dtype=dtype, mma=mma),
                                  StandardDAE(dp, 11, padding=0, dtype=dtype, mma=mma), 11, [11], dtype=dtype)
    X = S.make_images(32, 224, 224, seed=7000)
    ii64 = make(torch.float64)
    o = ii64.pred_fcn_fn(X)
    ref = ii64.refine(o[:-1], o[-1], 0.1, 10, early_stop=False)[0]
    conf = float(o[-1].amax(1).mean())
    T = S.labels_from_map(host(ref), seed=8000)
    ref_arg = ref.argmax(1)
    del ii64, o
    torch.cuda.empty_cache()
    ii8 = make(torch.float32, 'bf16c8')
    o8 = ii8.pred_fcn_fn(X)
    y8 = ii8.refine(o8[:-1], o8[-1], 0.1, 10, early_stop=False)[0]
    agree = float((y8.argmax(1) == ref_arg).float().mean())
    _, jacc, _ = ii8.val_fn(y8, T)
    with np.errstate(invalid='ignore', divide='ignore'):
        miou = float(np.nanmean(jacc[0] / jacc[1]))
    print('configs[2] as written, damped set, batch 32: mIoU %.5f (float64: 1 by construction), refined argmax '
          'agreement %.6f, mean max-probability of y0 %.3f' % (miou, agree, conf))
    assert abs(miou - 1.0) <= 0.05            # north_star: mIoU within +-0.05 of the (float64 = CPU) reference
    assert agree >= 0.97 and conf >= 0.7


def test_densenet_c8_small_against_float64(built_lib):
    """A small FC-DenseNet (2 pools, blocks [2, 2, 2, 2, 2]) on C8 stacks against the float64 HIP path on
    the same weights: every stage of the C8 forward (first conv into the stack slice, dense layers with
    input-side BN + ReLU, TransitionDown / TransitionUp through the layout converters, skip copy with
    its statistics) within the 16-bit mode's per-layer error -- a wiring error (wrong slice, wrong
    statistics) shows as an O(1) deviation."""
    from iterative_inference_segm_amd import synthetic as S
    from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan
    nl = [2, 2, 2, 2, 2]
    plan = layer_plan(n_layers_per_block=nl, n_pool=2)
    params = S.make_densenet_params(plan, seed=11)
    X = S.make_images(4, 36, 44, seed=5)
    n8 = FCDenseNet(params, 11, layer=['pool1', 'pool2'], n_layers_per_block=nl, n_pool=2, mma='bf16c8')
    n64 = FCDenseNet(params, 11, layer=['pool1', 'pool2'], n_layers_per_block=nl, n_pool=2,
                     dtype=torch.float64)
    assert n8.c8
    o8 = n8(torch.from_numpy(X).cuda())
    o64 = n64(torch.from_numpy(X).cuda().double())
    for a, b, name in zip(o8, o64, ('pool1', 'pool2', 'probs')):
        e = rel_rms(host(a), host(b))
        print('small DenseNet on C8 vs float64: %s rel rms %.2e' % (name, e))
        assert e <= 3e-2, name
    agree = float((host(o8[-1]).argmax(1) == host(o64[-1]).argmax(1)).mean())
    assert agree >= 0.97


def test_densenet_c8_forward_from_a_captured_graph_is_bit_identical(built_lib, monkeypatch):
    """FCDenseNet(mma='bf16c8') replays the forward of an input geometry from a captured HIP graph from its third
    call on (densenet.FWD_GRAPH): every call -- eager first, capture + replay second, replays after, a second
    geometry in between, the first one again -- returns the bits of the eager launches, and what an earlier call
    returned is not touched by a later one."""
    from iterative_inference_segm_amd import densenet as D, synthetic as S
    nl = [2, 2, 2, 2, 2]
    params = S.make_densenet_params(D.layer_plan(n_layers_per_block=nl, n_pool=2), seed=11)

    def net():
        return D.FCDenseNet(params, 11, layer=['input', 'pool1', 'pool2'], n_layers_per_block=nl, n_pool=2, mma='bf16c8')
    shapes = [(4, 36, 44), (4, 36, 44), (4, 36, 44), (2, 52, 40), (4, 36, 44), (2, 52, 40), (2, 52, 40), (4, 36, 44)]
    Xs = [torch.from_numpy(S.make_images(b, h, w, seed=20 + i)).cuda() for i, (b, h, w) in enumerate(shapes)]
    monkeypatch.setattr(D, 'FWD_GRAPH', False)
    eager = net()
    want = [[t.clone() for t in eager(X)] for X in Xs]
    assert not eager._c8_graphs
    monkeypatch.setattr(D, 'FWD_GRAPH', True)
    n = net()
    got = [n(X) for X in Xs]
    torch.cuda.synchronize()
    assert len(n._c8_graphs) == 2 and all(c['graph'] is not None for c in n._c8_graphs.values())
    for k, (g, w) in enumerate(zip(got, want)):
        assert len(g) == len(w) == 4
        for a, b in zip(g, w):
            assert torch.equal(a, b), k
