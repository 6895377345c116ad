"""mma='bf16x3' (IISEG_CONV_X3, csrc/conv_c8_bf16.hip): the fp32-class mode of the 16-bit matrix pipe.
Activations and weights are bf16 hi / lo pairs (16 significant bits), a layer accumulates
x_lo W_hi + x_hi W_lo + x_hi W_hi in fp32.  On integer data with at most 16 significant bits every
step is exact, so results must agree BIT FOR BIT with the float64 oracle: that pins the pair layout,
the three products of a channel tile, the split of outputs / pooled maps, the pair form of the skip addend and of the
DePool2D input; real-valued data then gives the error model (1e-5 class, against 4e-3 for one bf16
operand), and the end-to-end tests hold the mode to the 1e-4 of `north_star` on the damped set."""
import numpy as np
import pytest
import torch

from oracle import nn as onn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops(built_lib):
    from iterative_inference_segm_amd import ops as _ops
    return _ops


def host(t):
    torch.cuda.synchronize()
    return t.float().cpu().numpy() if t.dtype == torch.bfloat16 else t.cpu().numpy()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def from_pair(ops, t, C):
    """hi / lo pair (B, 2 C8, H, W, 8) -> (B, C, H, W) float64 on the host (hi + lo, exact in fp32)."""
    torch.cuda.synchronize()
    return ops.c8x3_to_float(t).cpu().numpy().astype(np.float64)[:, :C]


def from_c8f32(t, C):
    a = host(t).astype(np.float64)
    B, C8, H, W, _ = a.shape
    return a.transpose(0, 1, 4, 2, 3).reshape(B, C8 * 8, H, W)[:, :C]


def wide_ints(rng, *shape, small=3, big=1200, p_big=0.15):
    """Integers most of which are small, some need the lo half (e.g. 257, 1001: 9-11 significant bits)."""
    a = rng.integers(-small, small + 1, size=shape).astype(np.float64)
    b = rng.integers(-big, big + 1, size=shape).astype(np.float64)
    return np.where(rng.random(shape) < p_big, b, a)


def small_ints(rng, *shape, lo=-1, hi=2):
    return rng.integers(lo, hi, size=shape).astype(np.float64)


def _masks(pre):
    pooled = onn.maxpool2(pre)
    h, w = pooled.shape[2], pooled.shape[3]
    rep = np.repeat(np.repeat(pooled, 2, 2), 2, 3)
    eq = (pre[:, :, :2 * h, :2 * w] == rep)
    bits = (eq[:, :, 0::2, 0::2] * 1 + eq[:, :, 0::2, 1::2] * 2 + eq[:, :, 1::2, 0::2] * 4 +
            eq[:, :, 1::2, 1::2] * 8)
    return pooled, bits.astype(np.uint8)


def mask_from_c8(m):
    a = host(m)
    B, C8, H, W, _ = a.shape
    return a.transpose(0, 1, 4, 2, 3).reshape(B, C8 * 8, H, W)


def test_pair_converter_is_exact_on_16_bit_values(ops):
    rng = np.random.default_rng(0)
    x = rng.integers(-40000, 40001, size=(3, 11, 9, 13)).astype(np.float64)
    x8 = ops.nchw_to_c8(dev(x), x3=True)
    assert tuple(x8.shape) == (3, 4, 9, 13, 8) and ops.is_c8(x8)
    full = from_pair(ops, x8, 16)
    assert np.array_equal(full[:, :11], x) and not full[:, 11:].any()
    hi = host(x8[:, :2]).astype(np.float64)
    assert np.abs(hi).max() > 256 and not np.array_equal(hi.transpose(0, 1, 4, 2, 3).reshape(3, 16, 9, 13)[:, :11], x)
    # real values: 16 significant bits
    v = rng.standard_normal((2, 16, 8, 8))
    got = from_pair(ops, ops.nchw_to_c8(dev(v), x3=True), 16)
    assert np.abs(got - v.astype(np.float32)).max() <= 2.0 ** -16 * np.abs(v).max()


CASES = [  # B, Cin, H, W, Cout, pad, relu, window
    (2, 16, 20, 45, 64, 1, True, None),            # RECT, ragged tiles, one k-tile per group
    (1, 48, 9, 70, 72, 1, False, None),            # channel tails, three column tiles
    (2, 32, 12, 12, 64, 5, True, None),            # wide zero padding
    (3, 64, 40, 40, 128, 1, True, (6, 10, 21, 27)),  # window of a larger map
    (5, 32, 13, 13, 64, 1, True, None),            # FLAT: tiles run across images
    (7, 64, 22, 22, 96, 1, False, (5, 6, 10, 10)),   # FLAT window
    (2, 11, 17, 19, 64, 1, True, None),            # 11 input channels (the DAE's first layer)
    (2, 64, 70, 40, 64, 1, True, None),            # tall tiles where the launch is large enough
]


@pytest.mark.parametrize('split', ['activations', 'weights'])
@pytest.mark.parametrize('case', CASES)
def test_conv_x3_exact_on_integer_data(ops, case, split):
    """Either the activations or the weights need their lo halves (the x_lo W_lo term is the one the
    mode drops); outputs are read back as fp32 chunks and as pairs."""
    B, Cin, H, W, Cout, pad, relu, window = case
    rng = np.random.default_rng(sum(case[:6]))
    if split == 'activations':
        x, Wt = wide_ints(rng, B, Cin, H, W), small_ints(rng, Cout, Cin, 3, 3)
    else:
        x, Wt = small_ints(rng, B, Cin, H, W, lo=-2, hi=3), wide_ints(rng, Cout, Cin, 3, 3, small=1, big=700)
    b = small_ints(rng, Cout, lo=-3, hi=4)
    conv = ops.Conv(Wt, b, pad=pad, relu=relu, mma='bf16x3')
    ref = onn.conv2d(x, Wt, b, pad=pad, relu=relu)
    x8 = ops.nchw_to_c8(dev(x), x3=True)
    kw = dict(window=window) if window is not None else {}
    if window is not None:
        y0, x0, h, w = window
        ref = ref[:, :, y0:y0 + h, x0:x0 + w]
    assert np.abs(ref).max() < 2 ** 24
    got32 = conv(x8, out_format='c8f32', **kw)
    assert np.array_equal(from_c8f32(got32, Cout), ref)
    got = conv(x8, **kw)
    assert ops.is_c8(got) and got.shape[1] == 2 * ops.c8_chunks(Cout)
    full = from_pair(ops, got, got.shape[1] * 4)
    # the pair holds 16 significant bits: exact where the value has no more than that
    ok16 = np.abs(ref) < 2 ** 16
    assert ok16.mean() > 0.9
    assert np.array_equal(full[:, :Cout][ok16], ref[ok16])
    assert np.abs(full[:, :Cout] - ref).max() <= 2.0 ** -16 * np.abs(ref).max()
    assert not full[:, Cout:].any()


def test_conv_x3_drops_only_the_lo_lo_term(ops):
    """Both operands with lo halves: the result is conv(x_hi, W_hi) + conv(x_lo, W_hi) + conv(x_hi, W_lo)
    exactly (oracle on the split operands)."""
    rng = np.random.default_rng(3)
    B, Cin, H, W, Cout = 2, 32, 14, 37, 64
    x, Wt = wide_ints(rng, B, Cin, H, W, big=600), wide_ints(rng, Cout, Cin, 3, 3, small=1, big=300)
    bf = lambda a: torch.from_numpy(a).to(torch.float32).to(torch.bfloat16).to(torch.float64).numpy()
    xh, Wh = bf(x), bf(Wt)
    xl, Wl = x - xh, Wt - Wh
    assert np.abs(xl).max() > 0 and np.abs(Wl).max() > 0 and np.array_equal(bf(xl), xl) and np.array_equal(bf(Wl), Wl)
    ref = onn.conv2d(x, Wt, None, pad=1) - onn.conv2d(xl, Wl, None, pad=1)
    conv = ops.Conv(Wt, None, pad=1, relu=False, mma='bf16x3')
    got = from_c8f32(conv(ops.nchw_to_c8(dev(x), x3=True), out_format='c8f32'), Cout)
    assert np.abs(ref).max() < 2 ** 24
    assert np.array_equal(got, ref)


def test_conv_x3_skip_add_placement_and_score_layer(ops):
    rng = np.random.default_rng(5)
    B, Cin, Cout, H, W = 2, 32, 64, 30, 41
    x = wide_ints(rng, B, Cin, H, W, big=400)
    Wt, b = small_ints(rng, Cout, Cin, 3, 3), small_ints(rng, Cout, lo=-3, hi=4)
    conv = ops.Conv(Wt, b, pad=1, relu=False, mma='bf16x3')
    ref = onn.conv2d(x, Wt, b, pad=1, relu=False)
    x8 = ops.nchw_to_c8(dev(x), x3=True)
    for add_fmt in ('pair', 'c8f32'):
        skip = wide_ints(rng, B, Cout, H + 4, W + 6, big=5000)
        s8 = ops.nchw_to_c8(dev(skip), x3=(add_fmt == 'pair'))
        if add_fmt == 'c8f32':
            s8 = ops.nchw_to_c8(dev(skip), x3=True)
            s8 = (s8[:, :s8.shape[1] // 2].float() + s8[:, s8.shape[1] // 2:].float()).contiguous()
        y0, x0, h, w = 4, 8, 19, 30
        out = torch.zeros((B, 2 * ops.c8_chunks(Cout), H, W, 8), dtype=torch.bfloat16, device='cuda')
        out[:, :ops.c8_chunks(Cout)] = 7.0
        conv(x8, add=s8, add_off=(2 + y0, 3 + x0), window=(y0, x0, h, w), out=out, place=(y0, x0))
        got = from_pair(ops, out, Cout)
        want = np.full_like(ref, 7.0)
        want[:, :, y0:y0 + h, x0:x0 + w] = (ref + skip[:, :, 2:2 + H, 3:3 + W])[:, :, y0:y0 + h, x0:x0 + w]
        assert np.abs(want).max() < 2 ** 16
        assert np.array_equal(got, want), add_fmt
    # class-score layer: NCHW fp32 output
    Ws, bs = small_ints(rng, 11, Cin, 3, 3), small_ints(rng, 11)
    score = ops.Conv(Ws, bs, pad=1, relu=False, mma='bf16x3')
    got = score(x8, window=(3, 5, 16, 30))
    assert got.dtype == torch.float32 and tuple(got.shape) == (B, 11, 16, 30)
    assert np.array_equal(host(got), onn.conv2d(x, Ws, bs, pad=1)[:, :, 3:19, 5:35].astype(np.float32))


@pytest.mark.parametrize('shape', [(2, 32, 38, 45, 64), (4, 32, 15, 15, 64)])   # RECT fused / FLAT + pool kernel
def test_conv_x3_pool_and_mask_bytes(ops, shape):
    B, Cin, H, W, Cout = shape
    rng = np.random.default_rng(H)
    x = wide_ints(rng, B, Cin, H, W, big=300)
    Wt, b = small_ints(rng, Cout, Cin, 3, 3), small_ints(rng, Cout, lo=-3, hi=4)
    conv = ops.Conv(Wt, b, pad=1, relu=True, mma='bf16x3')
    pre = onn.conv2d(x, Wt, b, pad=1, relu=True)
    assert np.abs(pre).max() < 2 ** 16
    pooled, bits = _masks(pre)
    p8 = ops.empty_c8(B, Cout, H // 2, W // 2, 'cuda', x3=True)
    m8 = torch.zeros((B, ops.c8_chunks(Cout), H // 2, W // 2, 8), dtype=torch.uint8, device='cuda')
    assert conv(ops.nchw_to_c8(dev(x), x3=True), pool_out=p8, mask_out=m8, store_out=False) is None
    assert np.array_equal(from_pair(ops, p8, Cout), pooled)
    assert np.array_equal(mask_from_c8(m8)[:, :Cout], bits)
    # a window of whole pooling windows leaves the rest untouched
    p8.zero_(); m8.fill_(255)
    win = conv.pool_window(H, W, (5, 9, 7, 12))
    conv(ops.nchw_to_c8(dev(x), x3=True), pool_out=p8, mask_out=m8, store_out=False, window=win)
    y0, x0, h, w = win[0] // 2, win[1] // 2, win[2] // 2, win[3] // 2
    want_p = np.zeros_like(pooled); want_m = np.full_like(bits, 255)
    want_p[:, :, y0:y0 + h, x0:x0 + w] = pooled[:, :, y0:y0 + h, x0:x0 + w]
    want_m[:, :, y0:y0 + h, x0:x0 + w] = bits[:, :, y0:y0 + h, x0:x0 + w]
    assert np.array_equal(from_pair(ops, p8, Cout), want_p)
    assert np.array_equal(mask_from_c8(m8)[:, :Cout], want_m)


@pytest.mark.parametrize('shape', [(2, 32, 37, 44, 64), (6, 64, 13, 13, 32), (2, 128, 33, 33, 11)])
def test_conv_x3_depool_input_from_mask_bytes(ops, shape):
    """DePool2D (layers/mylayers.py:88-115) as the input staging: `up` pair + mask bytes."""
    B, Cin, H, W, Cout = shape
    rng = np.random.default_rng(W)
    pre = small_ints(rng, B, Cin, H, W, lo=0, hi=3)          # many ties
    pooled, bits = _masks(pre)
    up = wide_ints(rng, B, Cin, H // 2, W // 2, big=500)
    unp = onn.depool_eqmask(up, pre, pooled)
    Wt, b = small_ints(rng, Cout, Cin, 3, 3), small_ints(rng, Cout, lo=-3, hi=4)
    conv = ops.Conv(Wt, b, pad=1, relu=False, mma='bf16x3')
    ref = onn.conv2d(unp, Wt, b, pad=1, relu=False)
    m = np.zeros((B, ops.c8_chunks(Cin), H // 2, W // 2, 8), dtype=np.uint8)
    m[:] = bits.reshape(B, Cin // 8, 8, H // 2, W // 2).transpose(0, 1, 3, 4, 2)
    mt = torch.from_numpy(m).cuda()
    up8 = ops.nchw_to_c8(dev(up), x3=True)
    nchw = Cout % 8 != 0
    rd = (lambda t: host(t).astype(np.float64)) if nchw else (lambda t: from_c8f32(t, Cout))
    kw = {} if nchw else dict(out_format='c8f32')
    assert np.array_equal(rd(conv(up8, mask_in=mt, unpool_hw=(H, W), **kw)), ref)
    assert np.array_equal(rd(conv(up8, mask_in=mt, unpool_hw=(H, W), window=(3, 2, 9, 10), **kw)),
                          ref[:, :, 3:12, 2:12])


def test_conv_x3_error_model_at_layer_size(ops):
    """Real-valued data at a configs[1] layer size, against the float64 oracle: the split mode next to
    the fp32 MFMA path and the one-operand-rounding bf16 C8 mode."""
    rng = np.random.default_rng(9)
    B, Cin, H, W, Cout = 4, 128, 60, 60, 128
    x = rng.standard_normal((B, Cin, H, W))
    Wt = rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))
    b = 0.1 * rng.standard_normal(Cout)
    ref = onn.conv2d(x, Wt, b, pad=1, relu=True)
    rms = np.sqrt((ref ** 2).mean())
    c3 = ops.Conv(Wt, b, pad=1, relu=True, mma='bf16x3')
    got3 = from_pair(ops, c3(ops.nchw_to_c8(dev(x), x3=True)), Cout)
    got32 = host(ops.Conv(Wt, b, pad=1, relu=True)(dev(x))).astype(np.float64)
    c8 = ops.Conv(Wt, b, pad=1, relu=True, mma='bf16c8')
    a = host(c8(ops.nchw_to_c8(dev(x)))).astype(np.float64)
    got8 = a.transpose(0, 1, 4, 2, 3).reshape(B, -1, H, W)[:, :Cout]
    e3, e32, e8 = (float(np.sqrt(((g - ref) ** 2).mean()) / rms) for g in (got3, got32, got8))
    m3 = float(np.abs(got3 - ref).max() / np.abs(ref).max())
    print('128 -> 128 at 60x60, relative RMS error vs float64: bf16x3 %.2e (max %.2e), fp32 MFMA %.2e, '
          'bf16 C8 %.2e' % (e3, m3, e32, e8))
    assert e3 <= 1e-5 and m3 <= 2e-5 and e3 <= e8 / 100


def _damped_engine(dtype, mma=None, fcn_mma=None):
    from iterative_inference_segm_amd import synthetic as S
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp, dp, temp = S.make_damped_set()
    return IterativeInference(
        FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'], temperature=temp, dtype=dtype, mma=fcn_mma),
        StandardDAE(dp, 11, dtype=dtype, mma=mma), 11, [11], dtype=dtype)


def test_x3_engine_free_running_fixed_tolerance(built_lib):
    """configs[1] on the damped set (tests/test_gpu_damped.py), the 64 images of scripts/parity_report.py,
    10 steps of 0.1, early stop off, the product path (`refine()`, own masks, HIP-graph replay).  'x3' is
    the mode as bench.py runs it: the FCN-8 (once per batch) on its fp32 MFMA kernels, the DAE loop (10
    forwards per batch) on hi / lo pairs.  FIXED criteria, not the fp32 path's: >= 0.998 of the pixels within
    the 1e-4 of north_star (measured 0.9986; the fp32 path, held to >= 0.999 on the same images, measures
    0.9993), mean error <= 1e-5, argmax agreement >= 0.9999 with the float64 path.  'x3all' also runs the
    FCN-8's 3x3 layers on pairs: its 13 layers and the sharpened softmax of this set put 0.7 % of the
    pixels beyond 1e-4 (>= 0.99 asserted) -- which is why the mode keeps the FCN-8 in fp32."""
    from iterative_inference_segm_amd import synthetic as S
    TOL = 1e-4
    NIMG = 64                                   # the set of scripts/parity_report.py (8 batches of 8)
    modes = {'f64': (torch.float64, None, None), 'f32': (torch.float32, None, None),
             'x3': (torch.float32, 'bf16x3', None), 'x3all': (torch.float32, 'bf16x3', 'bf16x3')}
    acc = {k: dict(within=0, px=0, mx=0.0, sm=0.0, agree=0, r1mx=0.0) for k in modes if k != 'f64'}
    engines = {k: _damped_engine(dt, mma, fm) for k, (dt, mma, fm) in modes.items()}
    for bi in range(NIMG // 8):
        X = S.make_images(8, 224, 224, seed=5000 + bi)
        res = {}
        for k, ii in engines.items():
            out = ii.pred_fcn_fn(X)
            r1 = ii.pred_dae_fn(*out)
            res[k] = (r1.double(), ii.refine(out[:-1], out[-1], 0.1, 10, early_stop=False)[0].double())
        for k in acc:
            e = (res[k][1] - res['f64'][1]).abs()
            a = acc[k]
            a['within'] += int((e.amax(1) <= TOL).sum()); a['px'] += e.shape[0] * e.shape[2] * e.shape[3]
            a['mx'] = max(a['mx'], float(e.max())); a['sm'] += float(e.sum())
            a['agree'] += int((res[k][1].argmax(1) == res['f64'][1].argmax(1)).sum())
            a['r1mx'] = max(a['r1mx'], float((res[k][0] - res['f64'][0]).abs().max()))
    stats = {}
    for k, a in acc.items():
        stats[k] = (a['within'] / a['px'], a['mx'], a['sm'] / (a['px'] * 11), a['agree'] / a['px'])
        print('%s vs float64, %d images: one reconstruction max %.2e; after 10 steps pixels within 1e-4 '
              '%.5f, max %.2e, mean %.2e, argmax agreement %.6f' % ((k, NIMG, a['r1mx']) + stats[k]))
    # fixed criteria: the fp32 path >= 0.999 (measured 0.99930), the pair mode >= 0.998 (measured 0.99861:
    # fp32-CLASS, not fp32 -- DESIGN 3.8), with the FCN-8 on pairs too >= 0.99
    assert stats['f32'][0] >= 0.999
    frac, emax, emean, agree = stats['x3']
    assert frac >= 0.998 and emean <= 1e-5 and agree >= 0.9999 and emax <= 2e-3
    frac, emax, emean, agree = stats['x3all']
    assert frac >= 0.99 and emean <= 1e-5 and agree >= 0.9999


def test_x3_engine_work_eliminations_are_bit_identical(built_lib):
    """The exact work eliminations (decoder windows, loop-invariant encoder maps, weights-only border
    stores across batches, HIP-graph replay) under mma='bf16x3', against the same engine recomputing
    every layer in full for every step, eagerly -- bit for bit, three batches (as
    tests/test_gpu_c8.py does for the one-operand mode: same kernel, same fixed-order sums)."""
    from iterative_inference_segm_amd import synthetic as S
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8

    def small():
        fp = S.make_fcn8_params(width_div=4, fc_channels=1024, seed=481)
        dp = S.make_dae_params(h_channels=(fp['conv4_3'][0].shape[0],), n_filters=16, seed=482,
                               out_gain=0.25, dec_gain=0.35)
        return IterativeInference(
            FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'], mma='bf16x3'),
            StandardDAE(dp, 11, n_filters=16, mma='bf16x3'), 11, [11])
    ii_a, ii_b = small(), small()
    assert ii_a.dae.x3 and ii_a.dae.c8
    ii_b.dae.dce = ii_b.dae.licm = ii_b.dae.fold_border = False
    ii_b.fcn.fold_border = False
    B, H, W = 3, 64, 80
    ii_a.prepare(B, H, W)
    for i in range(3):
        X = S.make_images(B, H, W, seed=490 + i)
        oa, ob = ii_a.pred_fcn_fn(X), ii_b.pred_fcn_fn(X)
        for a, b in zip(oa, ob):
            assert np.array_equal(host(a), host(b)), 'FCN-8 output differs, batch %d' % i
        ra = ii_a.refine(oa[:-1], oa[-1], 0.2, 5, eps=1e-4, graph=True, first_reconstruction=True)
        rb = ii_b.refine(ob[:-1], ob[-1], 0.2, 5, eps=1e-4, graph=False, first_reconstruction=True)
        for a, b in zip(ra, rb):
            assert np.array_equal(host(a), host(b)), 'refined result differs, batch %d' % i


def test_conv_x3_random_geometries(ops):
    """40 seeded random launches of iiseg_conv_c8 with IISEG_CONV_X3 against the oracle, bit for bit on
    integer data whose lo halves are populated (activations in even cases, weights in odd ones):
    random channel counts, map sizes 5..70, paddings 1..6, batch 1..9, full maps and random windows
    with placement into a larger pair tensor, skip-add in both formats, ReLU, DePool2D input, fused /
    two-pass pool + mask bytes.  All tilings come up."""
    import ctypes
    from iterative_inference_segm_amd._lib import ConvDesc, CONV_X3
    rng = np.random.default_rng(4048)
    n_flat = 0
    for case in range(40):
        B = int(rng.integers(1, 10))
        Cin = 16 * int(rng.integers(1, 5))
        Cout = 8 * int(rng.integers(1, 13))
        H, W = int(rng.integers(5, 71)), int(rng.integers(5, 71))
        pad = int(rng.choice([1, 1, 1, 2, 6]))
        relu = bool(rng.integers(0, 2))
        unpool = case % 4 == 3
        if unpool:
            H, W = max(H, 6), max(W, 6)
        wide_x = case % 2 == 0
        Wt = small_ints(rng, Cout, Cin, 3, 3) if wide_x else wide_ints(rng, Cout, Cin, 3, 3, small=1, big=300)
        b = small_ints(rng, Cout, lo=-3, hi=4)
        conv = ops.Conv(Wt, b, pad=pad, relu=relu, mma='bf16x3')
        act = (lambda *s: wide_ints(rng, *s, small=2, big=300)) if wide_x else \
            (lambda *s: small_ints(rng, *s, lo=-2, hi=3))
        kw = {}
        if unpool:
            pre = small_ints(rng, B, Cin, H, W, lo=0, hi=3)
            pooled, bits = _masks(pre)
            up = act(B, Cin, H // 2, W // 2)
            x = onn.depool_eqmask(up, pre, pooled)
            m = np.zeros((B, ops.c8_chunks(Cin), H // 2, W // 2, 8), dtype=np.uint8)
            m[:] = bits.reshape(B, Cin // 8, 8, H // 2, W // 2).transpose(0, 1, 3, 4, 2)
            x8 = ops.nchw_to_c8(dev(up), x3=True)
            kw.update(mask_in=torch.from_numpy(m).cuda(), unpool_hw=(H, W))
        else:
            x = act(B, Cin, H, W)
            x8 = ops.nchw_to_c8(dev(x), x3=True)
        ref = onn.conv2d(x, Wt, b, pad=pad, relu=False)
        fh, fw = ref.shape[2], ref.shape[3]
        if case % 3 == 1:
            y0, x0 = int(rng.integers(0, fh // 2 + 1)), int(rng.integers(0, fw // 2 + 1))
            h, w = int(rng.integers(1, fh - y0 + 1)), int(rng.integers(1, fw - x0 + 1))
        else:
            y0, x0, h, w = 0, 0, fh, fw
        window = (y0, x0, h, w)
        add = None
        if case % 5 in (2, 4):
            add = wide_ints(rng, B, Cout, fh + 3, fw + 2, big=2000)
            ref = ref + add[:, :, 1:1 + fh, 2:2 + fw]
            a8 = ops.nchw_to_c8(dev(add), x3=True)
            if case % 5 == 4:
                n = a8.shape[1] // 2
                a8 = (a8[:, :n].float() + a8[:, n:].float()).contiguous()
            kw.update(add=a8, add_off=(1 + y0, 2 + x0))
        if relu:
            ref = np.maximum(ref, 0)
        assert np.abs(ref).max() < 2 ** 16, (case, np.abs(ref).max())     # exact as a pair
        if case % 2 == 0:
            out = torch.zeros((B, 2 * ops.c8_chunks(Cout), fh, fw, 8), dtype=torch.bfloat16, device='cuda')
            out[:, :ops.c8_chunks(Cout)] = -9.0
            got8 = conv(x8, window=window, out=out, place=(y0, x0), **kw)
            want = np.full((B, Cout, fh, fw), -9.0)
            want[:, :, y0:y0 + h, x0:x0 + w] = ref[:, :, y0:y0 + h, x0:x0 + w]
        else:
            got8 = conv(x8, window=window, **kw)
            want = ref[:, :, y0:y0 + h, x0:x0 + w]
        got = from_pair(ops, got8, Cout)
        assert np.array_equal(got, want), (case, B, Cin, Cout, H, W, pad, window, unpool,
                                           np.abs(got - want).max())
        d = ConvDesc()
        d.B, d.C1, d.C2, d.H, d.W = B, Cin, 0, H, W
        d.Cout, d.KH, d.KW, d.pad, d.dil = Cout, 3, 3, pad, 1
        d.oy0, d.ox0, d.OH, d.OW = window
        d.flags = CONV_X3
        n_flat += int(conv.lib.iiseg_conv_c8_is_flat(ctypes.byref(d)))
        if add is None and fh >= 4 and fw >= 4:
            pw = conv.pool_window(H, W, window)
            if pw is not None and pw[2] >= 2 and pw[3] >= 2:
                pre_full = np.maximum(onn.conv2d(x, Wt, b, pad=pad, relu=False), 0) if relu else \
                    onn.conv2d(x, Wt, b, pad=pad, relu=False)
                pooled, bits = _masks(pre_full)
                p8 = torch.zeros((B, 2 * ops.c8_chunks(Cout), fh // 2, fw // 2, 8), dtype=torch.bfloat16,
                                 device='cuda')
                m8 = torch.zeros((B, ops.c8_chunks(Cout), fh // 2, fw // 2, 8), dtype=torch.uint8,
                                 device='cuda')
                conv(x8, window=pw, pool_out=p8, mask_out=m8, store_out=False,
                     **{k: v for k, v in kw.items() if k in ('mask_in', 'unpool_hw')})
                qy0, qx0 = pw[0] // 2, pw[1] // 2
                qh = min((pw[0] + pw[2]) // 2, fh // 2) - qy0
                qw = min((pw[1] + pw[3]) // 2, fw // 2) - qx0
                gp, gm = from_pair(ops, p8, Cout), mask_from_c8(m8)[:, :Cout]
                assert np.array_equal(gp[:, :, qy0:qy0 + qh, qx0:qx0 + qw],
                                      pooled[:, :, qy0:qy0 + qh, qx0:qx0 + qw]), (case, 'pool')
                assert np.array_equal(gm[:, :, qy0:qy0 + qh, qx0:qx0 + qw],
                                      bits[:, :, qy0:qy0 + qh, qx0:qx0 + qw]), (case, 'mask')
    print('random X3 launches: %d of 40 on the flat tiling' % n_flat)
    assert 0 <= n_flat <= 39


@pytest.mark.parametrize('case', [(5, 96, 13, 13, 64, 1, True), (3, 256, 22, 22, 128, 1, False),
                                  (9, 128, 10, 10, 72, 1, True)])
def test_conv_x3_long_k_flat_exact_on_integer_data(ops, case):
    """IISEG_CONV_X3 on flat tiles with long k-loops (18 to 48 steps): bit for bit against the oracle
    with the lo halves in the activations, then in the weights."""
    B, Cin, H, W, Cout, pad, relu = case
    rng = np.random.default_rng(sum(case[:6]))
    for split in ('activations', 'weights'):
        if split == 'activations':
            x, Wt = wide_ints(rng, B, Cin, H, W, big=600), small_ints(rng, Cout, Cin, 3, 3)
        else:
            x, Wt = small_ints(rng, B, Cin, H, W, lo=-2, hi=3), wide_ints(rng, Cout, Cin, 3, 3, small=1, big=300)
        b = small_ints(rng, Cout, lo=-3, hi=4)
        conv = ops.Conv(Wt, b, pad=pad, relu=relu, mma='bf16x3')
        ref = onn.conv2d(x, Wt, b, pad=pad, relu=relu)
        x8 = ops.nchw_to_c8(dev(x), x3=True)
        assert np.abs(ref).max() < 2 ** 24
        assert np.array_equal(from_c8f32(conv(x8, out_format='c8f32'), Cout), ref)
        got = from_pair(ops, conv(x8), Cout)
        ok16 = np.abs(ref) < 2 ** 16
        assert ok16.mean() > 0.9 and np.array_equal(got[ok16], ref[ok16])


def test_x3_engine_on_360x480_frames(built_lib):
    """configs[3]'s geometry (360x480 CamVid frames) on the damped set, 2 frames, 10 steps free-running,
    against the float64 path, under the same two fixed criteria as at 224x224
    (test_x3_engine_free_running_fixed_tolerance): the fp32 path >= 0.999 of the pixels within 1e-4
    (measured 0.99929), the bf16x3 mode >= 0.998 (measured 0.99870, max 3.2e-4, mean 1.4e-6, argmax
    agreement 0.999997) -- the mode is fp32-CLASS, not fp32."""
    from iterative_inference_segm_amd import synthetic as S
    X = S.make_images(2, 360, 480, seed=4321)
    res = {}
    for k, (dt, mma) in {'f64': (torch.float64, None), 'f32': (torch.float32, None),
                         'x3': (torch.float32, 'bf16x3')}.items():
        ii = _damped_engine(dt, mma)
        out = ii.pred_fcn_fn(X)
        res[k] = ii.refine(out[:-1], out[-1], 0.1, 10, early_stop=False)[0].double()
        del ii
        torch.cuda.empty_cache()
    floor = {'f32': 0.999, 'x3': 0.998}
    for k in ('f32', 'x3'):
        e = (res[k] - res['f64']).abs()
        frac = float((e.amax(1) <= 1e-4).double().mean())
        agree = float((res[k].argmax(1) == res['f64'].argmax(1)).double().mean())
        print('360x480, %s vs float64 after 10 steps: pixels within 1e-4 %.5f, max %.2e, mean %.2e, argmax '
              'agreement %.6f' % (k, frac, float(e.max()), float(e.mean()), agree))
        assert frac >= floor[k] and float(e.mean()) <= 1e-5 and agree >= 0.9999


KNOBS = [  # concat_h, h channels, additional_pool, skip, unpool_type, padding
    (['pool4'], (9,), 2, True, 'trackind', 100),
    (['pool4'], (9,), 1, False, 'inverse', 100),
    (['input'], (3,), 2, True, 'trackind', 100),
    (['pool3', 'pool4'], (7, 9), 1, True, 'trackind', 100),
    (['pool4'], (9,), 2, True, 'trackind', 0),
]


@pytest.mark.parametrize('knobs', KNOBS)
def test_x3_dae_knobs_vs_oracle(built_lib, knobs):
    """dae_dict knobs the bf16x3 / C8 plan supports (concat points incl. `input` and two at once,
    additional_pool, skip, unpool_type trackind / inverse, padding 100 / 0), small standard DAE, one
    reconstruction r(y | h) and the de it gives, against the float64 oracle on the same inputs."""
    from oracle import dae as odae
    from iterative_inference_segm_amd import synthetic as S
    from iterative_inference_segm_amd.dae import StandardDAE
    concat_h, hch, ap, skip, ut, padding = knobs
    rng = np.random.default_rng(len(concat_h) * 7 + ap)
    Hh, Ww = (48, 64) if padding else (128, 192)      # six poolings need room without the pad-100 border
    y = rng.random((2, 11, Hh, Ww)).astype(np.float32); y /= y.sum(1, keepdims=True)
    hs = []
    for name, c in zip(concat_h, hch):
        if name == 'input':
            hs.append(rng.random((2, c, Hh, Ww)).astype(np.float32))
        else:
            s = 2 ** int(name[-1])
            q = (Hh + 2 * padding - 2) if padding else Hh
            r = (Ww + 2 * padding - 2) if padding else Ww
            hs.append(rng.random((2, c, q // s, r // s)).astype(np.float32))
    multi = len(concat_h) > 1
    dp = S.make_dae_params(h_channels=hch, concat_h=concat_h, n_filters=8, additional_pool=ap,
                           unpool_type=ut, seed=77, out_gain=0.25, dec_gain=0.35)
    kw = dict(concat_h=concat_h, n_filters=8, additional_pool=ap, skip=skip, unpool_type=ut,
              padding=padding)
    if multi:
        kw['pad_multi_concat'] = True
    to64 = lambda p: {k: tuple(np.asarray(a, np.float64) for a in v) for k, v in p.items()}
    try:
        r_ref = odae.dae_forward(to64(dp), [h.astype(np.float64) for h in hs], y.astype(np.float64), **kw)
    except Exception as e:       # a knob combination the reference's shapes do not admit
        pytest.skip('oracle: %s' % e)
    dae = StandardDAE(dp, 11, mma='bf16x3', **kw)
    assert dae.x3 and dae.c8
    got = host(dae(*[torch.from_numpy(h).cuda() for h in hs], torch.from_numpy(y).cuda())).astype(np.float64)
    e = np.abs(got - r_ref)
    frac = float((e.max(axis=1) <= 1e-4).mean())
    print('%s: r max err %.2e mean %.2e, pixels within 1e-4 %.5f' % (knobs, e.max(), e.mean(), frac))
    assert frac >= 0.999 and e.mean() <= 1e-5
