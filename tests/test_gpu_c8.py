"""bf16 C8 activations (csrc/conv_c8_bf16.hip, include/iiseg.h iiseg_conv_c8): the direct 3x3 kernel,
the layout converters and the pool + mask kernel against the float64 oracle.  On small-integer data
every bf16 rounding is exact, so results must agree BIT FOR BIT -- that pins every index of the chunk
layout, both pixel tilings (RECT 8 x 32 tiles, FLAT 256-pixel runs over the batch), the LDS-DMA
staging incl. zero padding, the DePool2D input from mask bytes (layers/mylayers.py:88-115), the
fused pool / mask-byte epilogue, skip-add, windows and placement."""
import numpy as np
import pytest
import torch

from oracle import nn as onn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops(built_lib):
    from iterative_inference_segm_amd import ops as _ops
    return _ops


# Pixel tilings forced through the test hook iiseg_conv_c8_force_tiling (kind, th, tw): automatic;
# 256- / 512-pixel tiles in the shape the planner picks; the flat list; and fixed tile shapes (a forced
# shape the launch cannot use -- odd with a fused pool, patch too large -- falls back to the planner's).
TILINGS = [(-1, 0, 0), (0, 0, 0), (1, 0, 0), (2, 0, 0), (-1, 8, 32), (-1, 16, 32), (-1, 6, 38),
           (-1, 13, 38), (-1, 10, 22), (-1, 2, 116), (-1, 34, 6)]


@pytest.fixture(params=TILINGS, ids=lambda t: 'auto' if t == (-1, 0, 0) else 'k%d_%dx%d' % t)
def tiling(request, built_lib):
    from iterative_inference_segm_amd import _lib
    lib = _lib.load()
    assert lib.iiseg_conv_c8_force_tiling(*request.param) == 0
    yield request.param
    assert lib.iiseg_conv_c8_force_tiling(-1, 0, 0) == 0


def host(t):
    torch.cuda.synchronize()
    return t.float().cpu().numpy() if t.dtype == torch.bfloat16 else t.cpu().numpy()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def from_c8(t, C):
    """(B, C8, H, W, 8) -> (B, C, H, W) float64 on the host."""
    a = host(t).astype(np.float64)
    B, C8, H, W, _ = a.shape
    return a.transpose(0, 1, 4, 2, 3).reshape(B, C8 * 8, H, W)[:, :C]


def ints(rng, *shape, lo=-3, hi=4):
    return rng.integers(lo, hi, size=shape).astype(np.float64)


def layer(ops, rng, Cin, Cout, pad=1, relu=True):
    # (sums must stay below 256 in magnitude to be exact in bf16: narrower weights for deep layers)
    W = ints(rng, Cout, Cin, 3, 3, lo=-2, hi=3) if Cin < 32 else ints(rng, Cout, Cin, 3, 3, lo=-1, hi=2)
    b = ints(rng, Cout)
    return W, b, ops.Conv(W, b, pad=pad, relu=relu, mma='bf16c8')


def test_converters_round_trip(ops):
    rng = np.random.default_rng(0)
    x = ints(rng, 3, 11, 9, 13, lo=-100, hi=100)
    x8 = ops.nchw_to_c8(dev(x))
    assert tuple(x8.shape) == (3, 2, 9, 13, 8) and ops.is_c8(x8)
    full = from_c8(x8, 16)
    assert np.array_equal(full[:, :11], x) and not full[:, 11:].any()
    assert np.array_equal(host(ops.c8_to_nchw(x8, 11)), x.astype(np.float32))


CASES = [  # B, Cin, H, W, Cout, pad, relu, window
    (2, 16, 20, 45, 64, 1, True, None),            # RECT, ragged tiles
    (1, 48, 9, 70, 72, 1, False, None),            # channel tails, three column tiles
    (2, 32, 12, 12, 64, 5, True, None),            # wide zero padding (the pad-100 rule, scaled)
    (3, 64, 40, 40, 128, 1, True, (6, 10, 21, 27)),  # window of a larger map
    (5, 32, 13, 13, 64, 1, True, None),            # FLAT: tiles run across images
    (7, 64, 22, 22, 96, 1, False, (5, 6, 10, 10)),   # FLAT window
    (2, 128, 17, 19, 64, 1, True, None),           # FLAT, 8 k-tiles
]


@pytest.mark.parametrize('case', CASES)
def test_conv_c8_exact_on_integer_data(ops, case, tiling):
    B, Cin, H, W, Cout, pad, relu, window = case
    rng = np.random.default_rng(sum(case[:6]))
    x = ints(rng, B, Cin, H, W) if Cin < 32 else ints(rng, B, Cin, H, W, lo=-2, hi=3)
    if Cin >= 128:
        x = ints(rng, B, Cin, H, W, lo=-1, hi=2)
    Wt, b, conv = layer(ops, rng, Cin, Cout, pad, relu)
    ref = onn.conv2d(x, Wt, b, pad=pad, relu=relu)
    got8 = conv(ops.nchw_to_c8(dev(x)), window=window)
    if window is not None:
        y0, x0, h, w = window
        ref = ref[:, :, y0:y0 + h, x0:x0 + w]
    assert ops.is_c8(got8) and got8.shape[1] == ops.c8_chunks(Cout)
    got = from_c8(got8, got8.shape[1] * 8)
    assert np.abs(ref).max() <= 256                   # every value bf16-exact
    assert np.array_equal(got[:, :Cout], ref), np.abs(got[:, :Cout] - ref).max()
    assert not got[:, Cout:].any()                    # padding channels stay zero


def test_conv_c8_placement_and_skip_add(ops, tiling):
    rng = np.random.default_rng(5)
    B, Cin, Cout, H, W = 2, 32, 64, 30, 41
    x = ints(rng, B, Cin, H, W)
    Wt, b, conv = layer(ops, rng, Cin, Cout, relu=False)
    ref = onn.conv2d(x, Wt, b, pad=1, relu=False)
    for add_fmt in ('c8', 'c8f32'):
        skip = ints(rng, B, Cout, H + 4, W + 6)
        s8 = ops.nchw_to_c8(dev(skip))
        if add_fmt == 'c8f32':
            s8 = s8.float()
        y0, x0, h, w = 4, 8, 19, 30
        out = torch.full((B, ops.c8_chunks(Cout), H, W, 8), 7.0, dtype=torch.bfloat16, device='cuda')
        conv(ops.nchw_to_c8(dev(x)), add=s8, add_off=(2 + y0, 3 + x0), window=(y0, x0, h, w), out=out,
             place=(y0, x0))
        got = from_c8(out, Cout)
        want = np.full_like(ref, 7.0)
        want[:, :, y0:y0 + h, x0:x0 + w] = (ref + skip[:, :, 2:2 + H, 3:3 + W])[:, :, y0:y0 + h, x0:x0 + w]
        assert np.array_equal(got, want)
    # fp32 C8 output (the loop-invariant h half), dense
    o32 = conv(ops.nchw_to_c8(dev(x)), out_format='c8f32')
    a = host(o32).astype(np.float64).transpose(0, 1, 4, 2, 3).reshape(B, -1, H, W)[:, :Cout]
    assert np.array_equal(a, ref)


def test_conv_c8_class_score_layer_nchw_output(ops, tiling):
    rng = np.random.default_rng(6)
    B, Cin, H, W = 2, 64, 21, 37
    x = ints(rng, B, Cin, H, W, lo=-2, hi=3)
    Wt = ints(rng, 11, Cin, 3, 3, lo=-1, hi=2)
    b = ints(rng, 11)
    conv = ops.Conv(Wt, b, pad=1, relu=False, mma='bf16c8')
    ref = onn.conv2d(x, Wt, b, pad=1, relu=False)
    got = conv(ops.nchw_to_c8(dev(x)), window=(3, 5, 16, 30))
    assert got.dtype == torch.float32 and tuple(got.shape) == (B, 11, 16, 30)
    assert np.array_equal(host(got), ref[:, :, 3:19, 5:35].astype(np.float32))


def _masks(pre):
    """oracle: pooled map and mask bytes (bit (y & 1) * 2 + (x & 1): pre == pooled) of `pre`."""
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


@pytest.mark.parametrize('shape', [(2, 32, 38, 45, 64), (4, 32, 15, 15, 64), (3, 16, 26, 20, 128)])
def test_conv_c8_pool_and_mask_bytes(ops, shape, tiling):
    B, Cin, H, W, Cout = shape
    rng = np.random.default_rng(H)
    x = ints(rng, B, Cin, H, W, lo=-2, hi=3)
    Wt, b, conv = layer(ops, rng, Cin, Cout)
    pre = onn.conv2d(x, Wt, b, pad=1, relu=True)
    pooled, bits = _masks(pre)
    p8 = ops.empty_c8(B, Cout, H // 2, W // 2, 'cuda')
    m8 = torch.zeros(p8.shape, dtype=torch.uint8, device='cuda')
    assert conv(ops.nchw_to_c8(dev(x)), pool_out=p8, mask_out=m8, store_out=False) is None
    assert np.array_equal(from_c8(p8, Cout), pooled)
    assert np.array_equal(mask_from_c8(m8)[:, :Cout], bits)
    # a window of whole pooling windows leaves the rest of the pooled / mask tensors untouched
    p8.fill_(3.0); m8.fill_(255)
    win = conv.pool_window(H, W, (5, 9, 7, 12))
    conv(ops.nchw_to_c8(dev(x)), pool_out=p8, mask_out=m8, store_out=False, window=win)
    y0, x0, h, w = win[0] // 2, win[1] // 2, win[2] // 2, win[3] // 2
    want_p = np.full_like(pooled, 3.0); want_m = np.full_like(bits, 255)
    want_p[:, :, y0:y0 + h, x0:x0 + w] = pooled[:, :, y0:y0 + h, x0:x0 + w]
    want_m[:, :, y0:y0 + h, x0:x0 + w] = bits[:, :, y0:y0 + h, x0:x0 + w]
    assert np.array_equal(from_c8(p8, Cout), want_p)
    assert np.array_equal(mask_from_c8(m8)[:, :Cout], want_m)


@pytest.mark.parametrize('shape', [(2, 32, 37, 44, 64), (6, 64, 13, 13, 32)])   # RECT / FLAT
def test_conv_c8_depool_input_from_mask_bytes(ops, shape, tiling):
    """DePool2D (layers/mylayers.py:88-115) as the conv's input staging: up (C8) + mask bytes."""
    B, Cin, H, W, Cout = shape
    rng = np.random.default_rng(W)
    pre = ints(rng, B, Cin, H, W, lo=0, hi=3)          # many ties
    pooled, bits = _masks(pre)
    up = ints(rng, B, Cin, H // 2, W // 2)
    unp = onn.depool_eqmask(up, pre, pooled)
    Wt, b, conv = layer(ops, rng, Cin, Cout, relu=False)
    ref = onn.conv2d(unp, Wt, b, pad=1, relu=False)
    m = np.zeros((B, ops.c8_chunks(Cin), H // 2, W // 2, 8), dtype=np.uint8)
    m[:] = bits.reshape(B, Cin // 8, 8, H // 2, W // 2).transpose(0, 1, 3, 4, 2)
    got = conv(ops.nchw_to_c8(dev(up)), mask_in=torch.from_numpy(m).cuda(), unpool_hw=(H, W))
    assert np.array_equal(from_c8(got, Cout), ref)
    got = conv(ops.nchw_to_c8(dev(up)), mask_in=torch.from_numpy(m).cuda(), unpool_hw=(H, W),
               window=(3, 2, 9, 10))
    assert np.array_equal(from_c8(got, Cout), ref[:, :, 3:12, 2:12])


def test_conv_c8_statistical_at_layer_size(ops):
    """Real-valued data at a configs[1] layer size: relative RMS error of the bf16 operand rounding."""
    rng = np.random.default_rng(9)
    B, Cin, H, W, Cout = 4, 128, 60, 60, 128
    x = rng.standard_normal((B, Cin, H, W))
    Wt = rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))
    b = 0.1 * rng.standard_normal(Cout)
    conv = ops.Conv(Wt, b, pad=1, relu=True, mma='bf16c8')
    ref = onn.conv2d(x, Wt, b, pad=1, relu=True)
    got = from_c8(conv(ops.nchw_to_c8(dev(x))), Cout)
    rel = float(np.sqrt(((got - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()))
    print('C8 conv 128 -> 128 at 60x60: relative RMS error %.2e' % rel)
    assert rel <= 8e-3


def _engine(mma, dtype=torch.float32, nf=16, div=4):
    from iterative_inference_segm_amd import synthetic as S
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp = S.make_fcn8_params(width_div=div, fc_channels=4096 // div, seed=481)
    dp = S.make_dae_params(h_channels=(fp['conv4_3'][0].shape[0],), n_filters=nf, seed=482,
                           out_gain=0.25, dec_gain=0.35)
    return IterativeInference(FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'], dtype=dtype, mma=mma),
                              StandardDAE(dp, 11, n_filters=nf, dtype=dtype, mma=mma), 11, [11],
                              dtype=dtype)


def test_c8_engine_work_eliminations_are_bit_identical(built_lib):
    """FCN-8 + standard DAE with bf16 C8 activations: the exact work eliminations (decoder windows,
    loop-invariant encoder maps, weights-only border stores across batches, HIP-graph replay) against
    the same engine recomputing every layer in full for every step, eagerly -- bit for bit, three
    batches.  Every output element is one fixed-order sum whichever tile, tiling (8 x 32 / flat) or
    window computes it, and the form a layer runs is a function of its geometry alone."""
    from iterative_inference_segm_amd import synthetic as S
    ii_a, ii_b = _engine('bf16c8'), _engine('bf16c8')
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


def test_c8_engine_close_to_float64(built_lib):
    """The same small network with C8 activations against the float64 path: FCN-8 output and one DAE
    reconstruction within the bf16 error model, next to the mma='bf16' mode (fp32 activations)."""
    from iterative_inference_segm_amd import synthetic as S
    X = S.make_images(2, 64, 80, seed=470)
    out = {}
    for k, (mma, dt) in {'c8': ('bf16c8', torch.float32), 'bf16': ('bf16', torch.float32),
                         'f64': (None, torch.float64)}.items():
        ii = _engine(mma, dt)
        o = ii.pred_fcn_fn(X)
        out[k] = (host(o[0]).astype(np.float64), host(o[-1]).astype(np.float64),
                  host(ii.pred_dae_fn(o[0], o[-1])).astype(np.float64))
    for k in ('c8', 'bf16'):
        eh = np.abs(out[k][0] - out['f64'][0]).mean() / np.abs(out['f64'][0]).mean()
        ey = np.abs(out[k][1] - out['f64'][1]).max()
        er = np.abs(out[k][2] - out['f64'][2]).max()
        agree = (out[k][2].argmax(1) == out['f64'][2].argmax(1)).mean()
        print('%s vs float64: h rel mean err %.2e, y0 max err %.2e, r max err %.2e, r argmax agreement '
              '%.4f' % (k, eh, ey, er, agree))
    eh = np.abs(out['c8'][0] - out['f64'][0]).mean() / np.abs(out['f64'][0]).mean()
    assert eh <= 2e-2 and np.abs(out['c8'][1] - out['f64'][1]).max() <= 5e-2
    assert np.abs(out['c8'][2] - out['f64'][2]).max() <= 5e-2


@pytest.mark.parametrize('force', [(-1, 0, 0), (1, 0, 0), (2, 0, 0)], ids=['auto', 'rect512', 'flat'])
def test_conv_c8_random_geometries(ops, force):
    """40 seeded random launches of iiseg_conv_c8 against the oracle, bit for bit on integer data:
    random channel counts (multiples of 16 in, multiples of 8 out), map sizes from 5 to 70, paddings 1
    to 6, batch 1 to 9, full maps and random windows with placement into a larger tensor, with and
    without skip-add (both formats), ReLU, DePool2D input, fused pool (where the window allows it).
    Every pixel tiling comes up: the planner's choice, and the 512-pixel tiles / the flat list forced
    wherever they can run the launch (the fused pool then runs on pooling-window-ordered tiles of each)."""
    from iterative_inference_segm_amd import _lib
    assert _lib.load().iiseg_conv_c8_force_tiling(*force) == 0
    try:
        _random_geometries(ops, force)
    finally:
        assert _lib.load().iiseg_conv_c8_force_tiling(-1, 0, 0) == 0


def _random_geometries(ops, force):
    rng = np.random.default_rng(2024)
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
        Wt = ints(rng, Cout, Cin, 3, 3, lo=-1, hi=2)
        b = ints(rng, Cout)
        conv = ops.Conv(Wt, b, pad=pad, relu=relu, mma='bf16c8')
        kw = {}
        if unpool:
            pre = ints(rng, B, Cin, H, W, lo=0, hi=3)
            pooled, bits = _masks(pre)
            up = ints(rng, B, Cin, H // 2, W // 2, lo=-2, hi=3)
            x = onn.depool_eqmask(up, pre, pooled)
            m = np.zeros((B, ops.c8_chunks(Cin), H // 2, W // 2, 8), dtype=np.uint8)
            m[:] = bits.reshape(B, Cin // 8, 8, H // 2, W // 2).transpose(0, 1, 3, 4, 2)
            x8 = ops.nchw_to_c8(dev(up))
            kw.update(mask_in=torch.from_numpy(m).cuda(), unpool_hw=(H, W))
        else:
            x = ints(rng, B, Cin, H, W, lo=-2, hi=3)
            x8 = ops.nchw_to_c8(dev(x))
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
            add = ints(rng, B, Cout, fh + 3, fw + 2)
            ref = ref + add[:, :, 1:1 + fh, 2:2 + fw]
            a8 = ops.nchw_to_c8(dev(add))
            kw.update(add=a8.float() if case % 5 == 4 else a8, add_off=(1 + y0, 2 + x0))
        if relu:
            ref = np.maximum(ref, 0)
        assert np.abs(ref).max() <= 256
        from iterative_inference_segm_amd._lib import ConvDesc
        place = case % 2 == 0
        if place:
            out = torch.full((B, ops.c8_chunks(Cout), fh, fw, 8), -9.0, dtype=torch.bfloat16, device='cuda')
            got8 = conv(x8, window=window, out=out, place=(y0, x0), **kw)
            want = np.full((B, Cout, fh, fw), -9.0)
            want[:, :, y0:y0 + h, x0:x0 + w] = ref[:, :, y0:y0 + h, x0:x0 + w]
        else:
            got8 = conv(x8, window=window, **kw)
            want = ref[:, :, y0:y0 + h, x0:x0 + w]
        got = from_c8(got8, Cout)
        assert np.array_equal(got, want), (case, B, Cin, Cout, H, W, pad, window, unpool, np.abs(got - want).max())
        d = ConvDesc()
        d.B, d.C1, d.C2, d.H, d.W = B, Cin, 0, H, W
        d.Cout, d.KH, d.KW, d.pad, d.dil = Cout, 3, 3, pad, 1
        d.oy0, d.ox0, d.OH, d.OW = window
        import ctypes
        n_flat += int(conv.lib.iiseg_conv_c8_is_flat(ctypes.byref(d)))
        # fused / two-pass pool + mask bytes of the same layer on a window of whole pooling windows
        if add is None and fh >= 4 and fw >= 4:
            pw = conv.pool_window(H, W, window)
            if pw is not None and pw[2] >= 2 and pw[3] >= 2:
                pre_full = np.maximum(onn.conv2d(x, Wt, b, pad=pad, relu=False), 0) if relu else \
                    onn.conv2d(x, Wt, b, pad=pad, relu=False)
                pooled, bits = _masks(pre_full)
                p8 = torch.zeros((B, ops.c8_chunks(Cout), fh // 2, fw // 2, 8), dtype=torch.bfloat16, device='cuda')
                m8 = torch.zeros(p8.shape, dtype=torch.uint8, device='cuda')
                conv(x8, window=pw, pool_out=p8, mask_out=m8, store_out=False,
                     **{k: v for k, v in kw.items() if k in ('mask_in', 'unpool_hw')})
                qy0, qx0 = pw[0] // 2, pw[1] // 2
                qh = min((pw[0] + pw[2]) // 2, fh // 2) - qy0
                qw = min((pw[1] + pw[3]) // 2, fw // 2) - qx0
                gp, gm = from_c8(p8, Cout), mask_from_c8(m8)[:, :Cout]
                assert np.array_equal(gp[:, :, qy0:qy0 + qh, qx0:qx0 + qw],
                                      pooled[:, :, qy0:qy0 + qh, qx0:qx0 + qw]), (case, 'pool')
                assert np.array_equal(gm[:, :, qy0:qy0 + qh, qx0:qx0 + qw],
                                      bits[:, :, qy0:qy0 + qh, qx0:qx0 + qw]), (case, 'mask')
    print('random C8 launches: %d of 40 on the flat tiling (%s)' % (n_flat, force))
    # (small batches: the planner's cost model rarely prefers the flat list; forced, it must run most)
    assert n_flat <= 39 if force[0] == -1 else (n_flat == 0) if force[0] == 1 else n_flat >= 20


@pytest.mark.parametrize('shape', [(2, 32, 38, 44, 64), (5, 48, 14, 14, 128), (2, 16, 21, 27, 64)])
def test_conv_c8_pool_with_fp32_addend(ops, shape, tiling):
    """The y half of the conv behind the h concat (models/fcn_down.py:102-122 after model_helpers.py:93-94):
    conv + cached fp32 addend (the loop-invariant h half) + ReLU + 2x2 max-pool + mask bytes, only the pool
    stored -- the straight-line epilogue EPI_POOL_ADD2 of conv_c8_kernel on every tiling."""
    B, Cin, H, W, Cout = shape
    rng = np.random.default_rng(H + Cout)
    x = ints(rng, B, Cin, H, W, lo=-2, hi=3)
    Wt, b, conv = layer(ops, rng, Cin, Cout)
    addend = ints(rng, B, Cout, H + 3, W + 5, lo=-6, hi=7)
    pre = np.maximum(onn.conv2d(x, Wt, b, pad=1, relu=False) + addend[:, :, 1:1 + H, 2:2 + W], 0)
    pooled, bits = _masks(pre)
    a8 = ops.nchw_to_c8(dev(addend)).float()
    p8 = ops.empty_c8(B, Cout, H // 2, W // 2, 'cuda')
    m8 = torch.zeros(p8.shape, dtype=torch.uint8, device='cuda')
    assert conv(ops.nchw_to_c8(dev(x)), add=a8, add_off=(1, 2), pool_out=p8, mask_out=m8,
                store_out=False) is None
    assert np.array_equal(from_c8(p8, Cout), pooled)
    assert np.array_equal(mask_from_c8(m8)[:, :Cout], bits)
    # a window of whole pooling windows
    p8.fill_(3.0); m8.fill_(255)
    win = conv.pool_window(H, W, (4, 6, 8, 10))
    conv(ops.nchw_to_c8(dev(x)), add=a8, add_off=(1 + win[0], 2 + win[1]), pool_out=p8, mask_out=m8,
         store_out=False, window=win)
    y0, x0, h, w = win[0] // 2, win[1] // 2, win[2] // 2, win[3] // 2
    want_p = np.full_like(pooled, 3.0); want_m = np.full_like(bits, 255)
    want_p[:, :, y0:y0 + h, x0:x0 + w] = pooled[:, :, y0:y0 + h, x0:x0 + w]
    want_m[:, :, y0:y0 + h, x0:x0 + w] = bits[:, :, y0:y0 + h, x0:x0 + w]
    assert np.array_equal(from_c8(p8, Cout), want_p)
    assert np.array_equal(mask_from_c8(m8)[:, :Cout], want_m)


@pytest.mark.parametrize('shape', [(3, 256, 20, 20, 64, (2, 3, 15, 14)), (6, 192, 12, 12, 128, None),
                                   (2, 16, 30, 42, 64, None), (2, 128, 34, 36, 32, (0, 0, 34, 36))])
def test_conv_c8_depool_long_k_with_skip_add(ops, shape, tiling):
    """DePool2D staging (up chunk + mask bytes through registers) on 1 to 16 k-tiles with the decoder's skip
    addend, every tiling, bit for bit."""
    B, Cin, H, W, Cout, window = shape
    rng = np.random.default_rng(Cin + W)
    pre = ints(rng, B, Cin, H, W, lo=0, hi=3)
    pooled, bits = _masks(pre)
    up = ints(rng, B, Cin, H // 2, W // 2, lo=-1, hi=2)
    unp = onn.depool_eqmask(up, pre, pooled)
    Wt = ints(rng, Cout, Cin, 3, 3, lo=-1, hi=2)
    b = ints(rng, Cout)
    conv = ops.Conv(Wt, b, pad=1, relu=False, mma='bf16c8')
    ref = onn.conv2d(unp, Wt, b, pad=1, relu=False)
    skip = np.clip(-np.round(ref) + ints(rng, B, Cout, H, W, lo=-3, hi=4), -256, 256)
    tot = ref + skip
    assert np.abs(tot).max() <= 256
    m = np.zeros((B, ops.c8_chunks(Cin), H // 2, W // 2, 8), dtype=np.uint8)
    m[:] = bits.reshape(B, Cin // 8, 8, H // 2, W // 2).transpose(0, 1, 3, 4, 2)
    y0, x0, h, w = window if window is not None else (0, 0, H, W)
    kw = dict(window=window) if window is not None else {}
    got = conv(ops.nchw_to_c8(dev(up)), mask_in=torch.from_numpy(m).cuda(), unpool_hw=(H, W),
               add=ops.nchw_to_c8(dev(skip)), add_off=(y0, x0), **kw)
    assert np.array_equal(from_c8(got, Cout), tot[:, :, y0:y0 + h, x0:x0 + w])


def test_unpool_c8_materialised_and_the_deep_decoder_levels(ops, monkeypatch):
    """ops.unpool_c8 (DePool2D materialised on C8 tensors, layers/mylayers.py:88-115) against the oracle, a
    window of it, and the engine: the decoder levels with >= 1024 input channels that unpool this way and run
    their conv as a plain layer give the SAME score map, bit for bit, as the levels that unpool in the conv's
    patch staging (full-size standard DAE, h = pool4, two steps of one session)."""
    rng = np.random.default_rng(77)
    B, Cc, H, W = 3, 32, 13, 18
    pre = ints(rng, B, Cc, H, W, lo=0, hi=3)
    pooled, bits = _masks(pre)
    up = ints(rng, B, Cc, H // 2, W // 2, lo=-5, hi=6)
    ref = onn.depool_eqmask(up, pre, pooled)
    m = np.zeros((B, ops.c8_chunks(Cc), H // 2, W // 2, 8), dtype=np.uint8)
    m[:] = bits.reshape(B, Cc // 8, 8, H // 2, W // 2).transpose(0, 1, 3, 4, 2)
    m8 = torch.from_numpy(m).cuda()
    out = torch.zeros((B, ops.c8_chunks(Cc), H, W, 8), dtype=torch.bfloat16, device='cuda')
    ops.unpool_c8(ops.nchw_to_c8(dev(up)), m8, out)
    assert np.array_equal(from_c8(out, Cc), ref)
    out.fill_(7.0)
    ops.unpool_c8(ops.nchw_to_c8(dev(up)), m8, out, window=(1, 2, 3, 5))
    want = np.full_like(ref, 7.0)
    want[:, :, 2:8, 4:14] = ref[:, :, 2:8, 4:14]
    assert np.array_equal(from_c8(out, Cc), want)
    # the engine
    from iterative_inference_segm_amd import dae as D, synthetic as S
    dp = S.make_dae_params()
    h = torch.from_numpy(rng.random((2, 512, 26, 26), dtype=np.float32)).cuda()
    ys = [torch.softmax(torch.from_numpy(rng.standard_normal((2, 11, 224, 224)).astype(np.float32)).cuda(), 1)
          for _ in range(2)]
    res = {}
    for mincin in (0, 1024):
        monkeypatch.setattr(D, 'C8_UNPOOL_MIN_CIN', mincin)
        net = D.StandardDAE(dp, 11, mma='bf16c8')
        sess = net.new_session([h], ys[0])
        res[mincin] = [host(net.scores([h], y, session=sess)) for y in ys]
        assert ('unp6' in sess) == (mincin > 0) and ('unp4' not in sess)
    for a, b in zip(res[0], res[1024]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('case', [(3, 48, 9, 11, 64, 16, 96), (2, 32, 14, 14, 80, 0, 32), (5, 16, 5, 7, 48, 32, 64)])
def test_conv_c8_zero_inserted_channel_slice_input(ops, case, tiling):
    """FC-DenseNet's TransitionUp (models/FCDenseNet.py:118-121: Deconv2DLayer 3x3, stride 2, 'valid', P3) on the
    C8 kernel: the block to upsample is read IN PLACE -- a channel slice [c0, c0 + Cin) of a wider stack -- and
    zero-inserted by the kernel's own patch staging (IISEG_CONV_ZINS).  Against the oracle's transposed convolution
    on integer data, bit for bit, whole map and a centre-crop window written into a slice of a wider output stack."""
    B, Cin, H, W, Cout, c0, ctot = case
    rng = np.random.default_rng(H * W + Cin)
    stack = ints(rng, B, ctot, H, W, lo=-2, hi=3)
    Wd = ints(rng, Cin, Cout, 3, 3, lo=-1, hi=2)                       # Deconv2DLayer W[in, out, 3, 3]
    b = ints(rng, Cout)
    ref = onn.deconv2d(stack[:, c0:c0 + Cin], Wd, b, stride=2)          # (B, Cout, 2 H + 1, 2 W + 1)
    assert ref.shape[2:] == (2 * H + 1, 2 * W + 1) and np.abs(ref).max() <= 256
    conv = ops.Conv(np.ascontiguousarray(Wd.transpose(1, 0, 2, 3)), b, pad=0, relu=False, mma='bf16c8')
    s8 = ops.nchw_to_c8(dev(stack))
    view = s8[:, c0 // 8:(c0 + Cin) // 8]
    assert not view.is_contiguous() or c0 == 0 and Cin == ctot
    got = conv(view, zins=True)
    assert np.array_equal(from_c8(got, Cout), ref)
    # centre crop to (2 H, 2 W) straight into channels [16, 16 + Cout) of a wider stack
    out = torch.full((B, ops.c8_chunks(Cout) + 4, 2 * H, 2 * W, 8), 5.0, dtype=torch.bfloat16, device='cuda')
    conv(view, zins=True, window=(0, 1, 2 * H, 2 * W), out=out, out_c0=16)
    full = from_c8(out, out.shape[1] * 8)
    assert np.array_equal(full[:, 16:16 + Cout], ref[:, :, 0:2 * H, 1:1 + 2 * W])
    assert (full[:, :16] == 5.0).all() and (full[:, 16 + ops.c8_chunks(Cout) * 8:] == 5.0).all()


DEEP_CASES = [  # B, Cin, H, W, Cout, pad, relu, window  (flat tiling, 16 to 32 k-tiles: the deep layers)
    (5, 256, 13, 13, 64, 1, True, None),
    (3, 320, 22, 22, 128, 1, False, (5, 6, 10, 10)),
    (9, 272, 10, 10, 72, 1, True, None),            # 810 pixels: a partly filled second 512-pixel tile
    (2, 512, 31, 31, 64, 1, True, None),            # 31-wide rows: the widest patch rows of configs[1]
]


@pytest.fixture(params=[(2, 0, 0), (-1, 0, 0)], ids=['flat', 'auto'])
def flat_forced(request, built_lib):
    from iterative_inference_segm_amd import _lib
    lib = _lib.load()
    assert lib.iiseg_conv_c8_force_tiling(*request.param) == 0
    yield request.param
    assert lib.iiseg_conv_c8_force_tiling(-1, 0, 0) == 0


@pytest.mark.parametrize('case', DEEP_CASES)
def test_conv_c8_long_k_flat_exact_on_integer_data(ops, case, flat_forced):
    """Flat-tiled launches with long k-loops (256 to 512 input channels, 10^2 to 31^2 windows, tiles
    that run across images) against the oracle, bit for bit: fp32 chunk output, bf16 output with
    skip-add and placement."""
    B, Cin, H, W, Cout, pad, relu, window = case
    rng = np.random.default_rng(sum(case[:6]))
    x = ints(rng, B, Cin, H, W, lo=-1, hi=2)
    Wt = ints(rng, Cout, Cin, 3, 3, lo=-1, hi=2)
    b = ints(rng, Cout)
    conv = ops.Conv(Wt, b, pad=pad, relu=relu, mma='bf16c8')
    ref = onn.conv2d(x, Wt, b, pad=pad, relu=relu)
    fh, fw = ref.shape[2], ref.shape[3]
    y0, x0, h, w = window if window is not None else (0, 0, fh, fw)
    x8 = ops.nchw_to_c8(dev(x))
    kw = dict(window=window) if window is not None else {}
    got32 = conv(x8, out_format='c8f32', **kw)
    a = host(got32).astype(np.float64).transpose(0, 1, 4, 2, 3).reshape(B, -1, h, w)[:, :Cout]
    assert np.array_equal(a, ref[:, :, y0:y0 + h, x0:x0 + w])
    # bf16 output placed into a larger tensor, with a skip addend that keeps every value bf16-exact
    skip = -np.round(onn.conv2d(x, Wt, b, pad=pad, relu=False)) + ints(rng, B, Cout, fh, fw, lo=-3, hi=4)
    skip = np.clip(skip, -256, 256)
    tot = onn.conv2d(x, Wt, b, pad=pad, relu=False) + skip
    if relu:
        tot = np.maximum(tot, 0)
    assert np.abs(tot).max() <= 256
    out = torch.full((B, ops.c8_chunks(Cout), fh, fw, 8), -9.0, dtype=torch.bfloat16, device='cuda')
    conv(x8, add=ops.nchw_to_c8(dev(skip)), add_off=(y0, x0), window=(y0, x0, h, w), out=out,
         place=(y0, x0))
    want = np.full((B, Cout, fh, fw), -9.0)
    want[:, :, y0:y0 + h, x0:x0 + w] = tot[:, :, y0:y0 + h, x0:x0 + w]
    assert np.array_equal(from_c8(out, Cout), want)


# ---- layers with at most 16 output channels: csrc/conv_c8_m16.hip (16-row MFMA) ----------------------

M16_CASES = [  # B, Cin, H, W, Cout, pad, relu, window
    (2, 16, 20, 45, 16, 1, True, None),
    (1, 48, 9, 70, 8, 1, False, None),
    (2, 32, 12, 12, 16, 5, True, None),              # wide zero padding
    (3, 64, 40, 40, 11, 1, False, (6, 10, 21, 27)),  # window, fp32 NCHW scores
    (5, 128, 13, 13, 16, 1, True, None),
    (2, 80, 33, 17, 12, 1, False, (0, 0, 33, 17)),
]


@pytest.mark.parametrize('shape_env', [None, '4,58', '16,16', '3,70'])
@pytest.mark.parametrize('case', M16_CASES)
def test_conv_c8_m16_exact_on_integer_data(ops, case, shape_env, monkeypatch):
    """The 16-row kernel against the oracle, bit for bit on integer data: bf16 C8 output for channel
    counts that are multiples of 8, fp32 NCHW otherwise; windows; channels past Cout stay zero.
    (IISEG_M16_SHAPE is read once per process: the forced shapes take effect when this test is the
    first m16 launch -- scripts/ run it that way; here the planner's shapes are what is pinned.)"""
    B, Cin, H, W, Cout, pad, relu, window = case
    rng = np.random.default_rng(sum(case[:6]) + 7)
    x = ints(rng, B, Cin, H, W, lo=-2, hi=3) if Cin < 128 else ints(rng, B, Cin, H, W, lo=-1, hi=2)
    Wt = ints(rng, Cout, Cin, 3, 3, lo=-1, hi=2)
    b = ints(rng, Cout)
    conv = ops.Conv(Wt, b, pad=pad, relu=relu, mma='bf16c8')
    ref = onn.conv2d(x, Wt, b, pad=pad, relu=relu)
    got = conv(ops.nchw_to_c8(dev(x)), window=window)
    if window is not None:
        y0, x0, h, w = window
        ref = ref[:, :, y0:y0 + h, x0:x0 + w]
    assert np.abs(ref).max() <= 256
    if Cout % 8 == 0:
        assert ops.is_c8(got) and got.shape[1] == 2
        full = from_c8(got, 16)
        assert np.array_equal(full[:, :Cout], ref) and not full[:, Cout:].any()
    else:
        assert got.dtype == torch.float32 and np.array_equal(host(got), ref.astype(np.float32))


def test_conv_c8_m16_depool_input_and_placement(ops):
    """DePool2D input (up + mask bytes) into the class-score layer, placed into a larger fp32 map."""
    rng = np.random.default_rng(77)
    B, Cin, H, W, Cout = 2, 64, 37, 44, 11
    pre = ints(rng, B, Cin, H, W, lo=0, hi=3)
    pooled, bits = _masks(pre)
    up = ints(rng, B, Cin, H // 2, W // 2, lo=-2, hi=3)
    unp = onn.depool_eqmask(up, pre, pooled)
    Wt = ints(rng, Cout, Cin, 3, 3, lo=-1, hi=2)
    b = ints(rng, Cout)
    conv = ops.Conv(Wt, b, pad=1, relu=False, mma='bf16c8')
    ref = onn.conv2d(unp, Wt, b, pad=1, relu=False)
    m = np.zeros((B, ops.c8_chunks(Cin), H // 2, W // 2, 8), dtype=np.uint8)
    m[:] = bits.reshape(B, Cin // 8, 8, H // 2, W // 2).transpose(0, 1, 3, 4, 2)
    mt = torch.from_numpy(m).cuda()
    got = conv(ops.nchw_to_c8(dev(up)), mask_in=mt, unpool_hw=(H, W))
    assert np.array_equal(host(got), ref.astype(np.float32))
    out = torch.full((B, Cout, H, W), -5.0, device='cuda')
    conv(ops.nchw_to_c8(dev(up)), mask_in=mt, unpool_hw=(H, W), window=(3, 2, 19, 30), out=out, place=(3, 2))
    want = np.full_like(ref, -5.0)
    want[:, :, 3:22, 2:32] = ref[:, :, 3:22, 2:32]
    assert np.array_equal(host(out), want.astype(np.float32))


def test_conv_c8_m16_dense_block_layer(ops):
    """A dense-block layer of FC-DenseNet on a C8 stack (models/FCDenseNet.py:61-146, BN_ReLU_Conv):
    BatchNorm + ReLU applied to the first n channels of the stack on the way in, 3x3 conv to 16 new
    channels written into the stack's next slice -- bit for bit on integer data (integer BN scale /
    shift), the rest of the stack untouched; then the batch statistics of the new channels."""
    rng = np.random.default_rng(31)
    B, cap, n, H, W = 3, 96, 48, 21, 30
    stack = np.zeros((B, cap, H, W))
    stack[:, :n] = ints(rng, B, n, H, W, lo=-3, hi=4)
    stack[:, n:] = 9.0                                   # (later slices: must not be read, 16 of them written)
    a = rng.integers(1, 3, size=n).astype(np.float64)
    bsh = rng.integers(-2, 3, size=n).astype(np.float64)
    xin = np.maximum(stack[:, :n] * a[None, :, None, None] + bsh[None, :, None, None], 0)
    Wt = ints(rng, 16, n, 3, 3, lo=-1, hi=2)
    bias = ints(rng, 16)
    ref = onn.conv2d(xin, Wt, bias, pad=1, relu=False)
    assert np.abs(ref).max() <= 256 and np.abs(xin).max() <= 256
    conv = ops.Conv(Wt, bias, pad=1, relu=False, mma='bf16c8')
    s8 = ops.nchw_to_c8(dev(stack))
    at = torch.zeros(cap, device='cuda'); bt = torch.zeros(cap, device='cuda')
    at[:n] = torch.from_numpy(a).float().cuda(); bt[:n] = torch.from_numpy(bsh).float().cuda()
    r = conv(s8, in_c=n, bn=(at, bt), out=s8, out_c0=n)
    assert r is s8
    full = from_c8(s8, cap)
    assert np.array_equal(full[:, :n], stack[:, :n])
    assert np.array_equal(full[:, n:n + 16], ref)
    assert np.array_equal(full[:, n + 16:], stack[:, n + 16:])
    # statistics of the new 16 channels (biased variance, eps 1e-4: Lasagne BatchNormLayer, P10)
    mean = torch.zeros(cap, device='cuda'); inv = torch.zeros(cap, device='cuda')
    ops.bn_stats_c8(s8, n, 16, mean, inv, eps=1e-4)
    mref = ref.mean(axis=(0, 2, 3)); vref = ref.var(axis=(0, 2, 3))
    assert np.allclose(host(mean)[n:n + 16], mref, rtol=1e-6, atol=1e-6)
    assert np.allclose(host(inv)[n:n + 16], 1.0 / np.sqrt(vref + 1e-4), rtol=1e-6)
    assert not host(mean)[:n].any() and not host(mean)[n + 16:].any()
    # the (a, b) pair from beta, gamma, mean, inv_std
    beta = torch.rand(cap, device='cuda'); gamma = torch.rand(cap, device='cuda') + 0.5
    fa, fb = ops.bn_fold(beta, gamma, mean, inv, n + 16, cap=cap)
    g, be, mm, ii = (host(t).astype(np.float64) for t in (gamma, beta, mean, inv))
    assert np.allclose(host(fa)[:n + 16], (g * ii)[:n + 16], rtol=1e-6)
    assert np.allclose(host(fb)[:n + 16], (be - mm * g * ii)[:n + 16], rtol=1e-5, atol=1e-6)
    assert not host(fa)[n + 16:].any()


@pytest.mark.parametrize('case', [(3, 112, 48, 22, 30, 48),     # B, cap, n, H, W, Cout: one 64-channel tile, partial
                                  (2, 96, 80, 13, 18, 80),      # two m-tiles, odd map (ignore_border)
                                  (1, 32, 32, 40, 66, 32)])
def test_conv1x1_c8_transition_down(ops, case):
    """FC-DenseNet's TransitionDown on a C8 stack as ONE kernel (csrc/conv1x1_c8.hip; FC_DenseNet.layers
    TransitionDown, models/FCDenseNet.py:95): BatchNorm + ReLU of the first n channels on the way in, 1x1
    conv, 2x2 max-pool (ignore_border), + bias, bf16 C8 into a slice of the next stack -- bit for bit on
    integer data, the rest of the target untouched."""
    B, cap, n, H, W, Cout = case
    rng = np.random.default_rng(sum(case))
    stack = np.zeros((B, cap, H, W))
    stack[:, :n] = ints(rng, B, n, H, W, lo=-3, hi=4)
    stack[:, n:] = 9.0                                   # later slices: must not be read
    a = rng.integers(1, 3, size=n).astype(np.float64)
    bsh = rng.integers(-2, 3, size=n).astype(np.float64)
    xin = np.maximum(stack[:, :n] * a[None, :, None, None] + bsh[None, :, None, None], 0)
    Wt = ints(rng, Cout, n, 1, 1, lo=-1, hi=2)
    bias = ints(rng, Cout)
    ref = onn.maxpool2(onn.conv2d(xin, Wt, bias, pad=0, relu=False))
    # (the stored value is the fp32 result rounded to bf16: the same rounding on the reference)
    ref = torch.from_numpy(ref).float().bfloat16().float().numpy().astype(np.float64)
    conv = ops.Conv1x1C8(Wt, bias)
    s8 = ops.nchw_to_c8(dev(stack))
    at = torch.zeros(cap, device='cuda'); bt = torch.zeros(cap, device='cuda')
    at[:n] = torch.from_numpy(a).float().cuda(); bt[:n] = torch.from_numpy(bsh).float().cuda()
    ocap = Cout + 32
    nxt = torch.full((B, ocap // 8, H // 2, W // 2, 8), 5.0, dtype=torch.bfloat16, device='cuda')
    r = conv(s8, n, bn=(at, bt), pool=True, out=nxt, out_c0=16)
    assert r is nxt
    full = from_c8(nxt, ocap)
    assert np.array_equal(full[:, 16:16 + Cout], ref)
    assert np.all(full[:, :16] == 5.0) and np.all(full[:, 16 + Cout:] == 5.0)
    assert np.array_equal(from_c8(s8, cap), stack)       # the source stack is only read


def test_conv1x1_c8_score_layer(ops):
    """The SoftmaxLayer's 1x1 class-score convolution on the C8 stack (models/FCDenseNet.py:134): fp32 NCHW
    out, no BN, exact on integer data; pixel count not a multiple of the 512-pixel tile."""
    rng = np.random.default_rng(77)
    B, n, H, W, Cout = 2, 64, 19, 31, 11
    x = ints(rng, B, n, H, W, lo=-3, hi=4)
    Wt = ints(rng, Cout, n, 1, 1, lo=-2, hi=3)
    bias = ints(rng, Cout)
    got = host(ops.Conv1x1C8(Wt, bias)(ops.nchw_to_c8(dev(x)), n))
    assert got.shape == (B, Cout, H, W) and got.dtype == np.float32
    assert np.array_equal(got, onn.conv2d(x, Wt, bias, pad=0, relu=False))


@pytest.mark.parametrize('case', [(3, 7, 7, 144), (2, 14, 14, 208), (2, 28, 28, 96), (1, 20, 33, 64)])
def test_conv_c8_m16_split_k(ops, case):
    """Small maps with long channel loops (FC-DenseNet's deep dense blocks): the channel range of a tile dealt to
    several workgroups, slabs of fp32 partial sums added in slice order by a second launch (iiseg_conv_c8_m16_ws)
    -- exact on integer data, repeatable, the stack around the slice untouched."""
    import ctypes as C
    B, H, W, n = case
    rng = np.random.default_rng(sum(case))
    cap = n + 32
    stack = np.zeros((B, cap, H, W))
    stack[:, :n] = ints(rng, B, n, H, W, lo=-2, hi=3)
    stack[:, n:] = 9.0
    a = rng.integers(1, 3, size=n).astype(np.float64)
    bsh = rng.integers(-1, 2, size=n).astype(np.float64)
    xin = np.maximum(stack[:, :n] * a[None, :, None, None] + bsh[None, :, None, None], 0)
    Wt = (rng.random((16, n, 3, 3)) < 0.08) * ints(rng, 16, n, 3, 3, lo=-1, hi=2)    # sparse: sums stay bf16-exact
    bias = ints(rng, 16)
    ref = onn.conv2d(xin, Wt, bias, pad=1, relu=False)
    assert np.abs(ref).max() <= 256
    conv = ops.Conv(Wt, bias, pad=1, relu=False, mma='bf16c8')
    at = torch.zeros(cap, device='cuda'); bt = torch.zeros(cap, device='cuda')
    at[:n] = torch.from_numpy(a).float().cuda(); bt[:n] = torch.from_numpy(bsh).float().cuda()
    for rep in range(2):
        s8 = ops.nchw_to_c8(dev(stack))
        conv(s8, in_c=n, bn=(at, bt), out=s8, out_c0=n)
        full = from_c8(s8, cap)
        assert np.array_equal(full[:, n:n + 16], ref), rep
        assert np.array_equal(full[:, :n], stack[:, :n]) and np.array_equal(full[:, n + 16:], stack[:, n + 16:])
    d = ops.ConvDesc()
    d.B, d.C1, d.C2, d.H, d.W, d.Cout, d.KH, d.KW, d.pad, d.dil, d.OH, d.OW = B, n, 0, H, W, 16, 3, 3, 1, 1, H, W
    # (beyond the per-tile statistics -- 256 bytes a tile -- the scratch holds slabs only for split launches: one
    # tile per image and >= 8 k-tiles)
    split = conv.lib.iiseg_conv_c8_m16_workspace_bytes(C.byref(d)) > 256 * B * 4
    assert split == ((H, W) in ((7, 7), (14, 14)))


@pytest.mark.parametrize('shape', [(3, 21, 30), (4, 7, 7)])
def test_conv_c8_m16_fused_statistics_and_next_fold(ops, shape):
    """A dense-block layer with the BatchNorm bookkeeping inside its launches: the batch statistics of the 16
    produced channels out of the conv's epilogue (per-tile sums in double, added in a fixed order), and the (a, b)
    pair of the NEXT layer's BatchNorm formed by the reduction that finishes them.  Output bit-identical to the
    plain launch; statistics equal to bn_stats_c8's up to the order of the double sums; the folded pair equal to
    bn_fold on those statistics.  (7 x 7: a split-K launch -- the statistics come from its second phase.)"""
    B, H, W = shape
    rng = np.random.default_rng(B * H)
    cap, n = 96, 64
    stack = torch.zeros((B, cap, H, W), device='cuda')
    stack[:, :n] = torch.from_numpy(rng.standard_normal((B, n, H, W)).astype(np.float32)).cuda()
    s8 = ops.nchw_to_c8(stack)
    mean = torch.zeros(cap, device='cuda'); inv = torch.zeros(cap, device='cuda')
    ops.bn_stats_c8(s8, 0, n, mean, inv, eps=1e-4)
    gamma = torch.rand(cap, device='cuda') + 0.5; beta = torch.rand(cap, device='cuda') - 0.5
    g2 = torch.rand(cap, device='cuda') + 0.5; b2 = torch.rand(cap, device='cuda') - 0.5
    Wt = rng.standard_normal((16, n, 3, 3)).astype(np.float32) / np.sqrt(9 * n)
    conv = ops.Conv(Wt, rng.standard_normal(16).astype(np.float32), pad=1, relu=False, mma='bf16c8')
    a, b = ops.bn_fold(beta, gamma, mean, inv, n, cap=cap)
    ref8 = s8.clone()
    conv(ref8, in_c=n, bn=(a, b), out=ref8, out_c0=n)
    m_ref, i_ref = mean.clone(), inv.clone()
    ops.bn_stats_c8(ref8, n, 16, m_ref, i_ref, eps=1e-4)
    got8 = s8.clone()
    m_got, i_got = mean.clone(), inv.clone()
    a2 = torch.zeros(cap, device='cuda'); bb2 = torch.zeros(cap, device='cuda')
    conv(got8, in_c=n, bn=(a, b), out=got8, out_c0=n, stats=(m_got, i_got, 1e-4), fold=(b2, g2, a2, bb2, n + 16))
    assert torch.equal(got8, ref8)
    assert np.allclose(host(m_got), host(m_ref), rtol=1e-6, atol=1e-7)
    assert np.allclose(host(i_got), host(i_ref), rtol=1e-6)
    assert np.array_equal(host(m_got)[:n], host(mean)[:n]) and not host(m_got)[n + 16:].any()
    fa, fb = ops.bn_fold(b2, g2, m_got, i_got, n + 16, cap=cap)
    assert torch.equal(fa, a2) and torch.equal(fb, bb2)


def test_c8_full_config_image_bits_do_not_depend_on_the_batch(built_lib):
    """BASELINE configs[1] on bf16 C8 at full size: an image refined inside the batch of 64, in the reference's own
    batch of 10 (iterative_inference.py:117) and in a batch of 3 gives BIT-IDENTICAL maps -- other pixel tilings
    (the flat lists run across images), other workgroup counts, the three-stage ring the launches that leave the
    chip half empty run on: every output is one fixed-order sum.  Data-parallel shards of any size therefore
    reproduce the single-GPU result bit for bit."""
    from iterative_inference_segm_amd import synthetic as S
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp, dp = S.make_fcn8_params(), S.make_dae_params()

    def make():
        return IterativeInference(FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'], mma='bf16c8'),
                                  StandardDAE(dp, 11, mma='bf16c8'), 11, [11])
    X = S.make_images(64, 224, 224, seed=331)
    ii = make()
    o = ii.pred_fcn_fn(X)
    Y64, _, n64 = ii.refine(o[:-1], o[-1], 0.1, 10, early_stop=False)
    y0, Y64, n64 = host(o[-1]), host(Y64), host(n64)
    for sub in (list(range(10)), [5, 17, 63]):
        jj = make()
        o2 = jj.pred_fcn_fn(X[sub])
        assert np.array_equal(host(o2[-1]), y0[sub])
        Y2, _, n2 = jj.refine(o2[:-1], o2[-1], 0.1, 10, early_stop=False)
        assert np.array_equal(host(Y2), Y64[sub])
        assert np.array_equal(host(n2), n64[sub])
