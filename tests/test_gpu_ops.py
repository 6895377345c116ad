"""GPU parity: every HIP kernel, called through the C ABI (ctypes), against the float64 numpy
oracle on the same seeded inputs.  Tolerances: fp32 kernels vs float64 oracle, |err| <=
1e-5 * (1 + |ref|) scaled by the reduction depth for convolutions; integer/compare work
(pool, unpool masks, confusion counts) is bit-exact."""
import numpy as np
import pytest
import torch

from oracle import nn as onn
from oracle import metrics as ometrics

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops(built_lib):
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from iterative_inference_segm_amd import ops as _ops
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def rnd(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


def conv_tol(ref, K):
    return 2e-6 * np.sqrt(K) * (1.0 + np.abs(ref).max())


CONV_CASES = [
    # B, Cin, H, W, Cout, k, pad, dil, relu   (covers BM=32/64/128 variants, ragged tiles)
    (2, 3, 17, 19, 11, 3, 1, 1, False),
    (1, 11, 20, 20, 64, 3, 5, 1, True),     # big pad, BM=64
    (3, 40, 13, 9, 130, 3, 1, 1, True),     # BM=128, 2 m-tiles, ragged Cout
    (2, 16, 9, 9, 200, 7, 0, 1, True),      # 7x7 valid (fc6-like)
    (2, 70, 7, 7, 33, 1, 0, 1, True),       # 1x1, BM=64
    (1, 11, 40, 36, 11, 3, 0, 4, False),    # dilated, valid (contextmod-like)
    (1, 5, 300, 7, 12, 3, 1, 1, False),     # many pixel tiles, tall image
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_matches_oracle(ops, case):
    B, Cin, H, W, Cout, k, pad, dil, relu = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x, Wt, b = rnd(rng, B, Cin, H, W), rnd(rng, Cout, Cin, k, k) / np.sqrt(Cin * k * k), rnd(rng, Cout)
    ref = onn.conv2d(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64),
                     pad=pad, dilation=dil, relu=relu)
    conv = ops.Conv(Wt, b, pad=pad, relu=relu, dil=dil)
    got = host(conv(dev(x)))
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= conv_tol(ref, Cin * k * k)


def test_conv_iohw_layout(ops):
    """DilatedConv2DLayer weight layout W[in,out,kh,kw] (P11)."""
    rng = np.random.default_rng(5)
    x, Wio, b = rnd(rng, 1, 6, 12, 12), rnd(rng, 6, 9, 3, 3), rnd(rng, 9)
    ref = onn.conv2d(x.astype(np.float64), np.transpose(Wio, (1, 0, 2, 3)).astype(np.float64),
                     b.astype(np.float64), pad=0, dilation=2)
    got = host(ops.Conv(Wio, b, pad=0, relu=False, dil=2, layout='iohw')(dev(x)))
    assert np.abs(got - ref).max() <= conv_tol(ref, 54)


def test_conv_two_source_concat(ops):
    """h-first channel concat fused in the gather (model_helpers.py:93-94, P13)."""
    rng = np.random.default_rng(7)
    h, t = rnd(rng, 2, 24, 10, 11), rnd(rng, 2, 8, 10, 11)
    Wt, b = rnd(rng, 40, 32, 3, 3) / 17, rnd(rng, 40)
    ref = onn.conv2d(onn.concat_h_first(h.astype(np.float64), t.astype(np.float64)),
                     Wt.astype(np.float64), b.astype(np.float64), pad=1, relu=True)
    got = host(ops.Conv(Wt, b, pad=1, relu=True)(dev(h), x2=dev(t)))
    assert np.abs(got - ref).max() <= conv_tol(ref, 288)


def test_conv_window_and_add(ops):
    """Output window (center crop) + epilogue add with its own crop offset (P6)."""
    rng = np.random.default_rng(8)
    x = rnd(rng, 2, 9, 14, 15)
    Wt, b = rnd(rng, 11, 9, 3, 3) / 9, rnd(rng, 11)
    other = rnd(rng, 2, 11, 10, 9)
    full = onn.conv2d(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), pad=4)
    ref = onn.crop_sum(full, other.astype(np.float64))          # (2, 11, 10, 9)
    fh, fw = full.shape[2:]
    got = host(ops.Conv(Wt, b, pad=4, relu=False)(
        dev(x), add=dev(other), add_off=(0, 0),
        window=((fh - 10) // 2, (fw - 9) // 2, 10, 9)))
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= conv_tol(ref, 81)


WINO_CASES = [
    # B, Cin, H, W, Cout, pad, relu   (Cin % 16 == 0; Cout <= 128 -> 128-tile, > 128 -> 256-tile)
    (2, 16, 9, 11, 8, 1, True),
    (1, 32, 14, 6, 40, 0, False),
    (3, 48, 7, 7, 130, 5, True),
    (2, 64, 12, 13, 260, 1, True),
    (1, 16, 40, 33, 24, 100, True),     # the pad-100 geometry
    (70, 32, 5, 4, 16, 1, False),       # many images, few tiles each
]


@pytest.mark.parametrize('fused_max', [0, 4096])
@pytest.mark.parametrize('case', WINO_CASES)
def test_conv_winograd_matches_oracle(ops, case, fused_max, monkeypatch):
    """Winograd F(2x2,3x3) path (iiseg_conv_wino_f32) against the float64 oracle conv; both the
    GEMM + output-transform pair and the fused kernel (Cin % 32 == 0 cases)."""
    monkeypatch.setattr(ops, 'WINO_FUSED_MAX_CIN', fused_max)
    B, Cin, H, W, Cout, pad, relu = case
    rng = np.random.default_rng(hash(case) % 2**32)
    x, w, b = rnd(rng, B, Cin, H, W), rnd(rng, Cout, Cin, 3, 3) * 0.2, rnd(rng, Cout)
    conv = ops.Conv(w, b, pad=pad, relu=relu)
    conv.wino = True
    got = host(conv(dev(x)))
    assert conv._U is not None, 'Winograd path did not run'
    ref = onn.conv2d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), pad=pad)
    if relu:
        ref = np.maximum(ref, 0)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 4 * conv_tol(ref, Cin * 9)


@pytest.mark.parametrize('anchor', [(0, 0), (1, 1), (0, 1)])
@pytest.mark.parametrize('fused_max', [0, 4096])
def test_conv_winograd_windows_are_bit_identical(ops, fused_max, anchor, monkeypatch):
    monkeypatch.setattr(ops, 'WINO_FUSED_MAX_CIN', fused_max)
    _winograd_windows(ops, anchor)


def _winograd_windows(ops, anchor):
    """Any window of a layer (odd/even origins, placement into a larger plane, channel slice,
    two-source concat, skip add) gives exactly the values of the full-map launch."""
    rng = np.random.default_rng(77)
    B, C1, C2, H, W, Cout, pad = 2, 16, 48, 15, 18, 36, 3
    x1, x2 = rnd(rng, B, C1, H, W), rnd(rng, B, C2, H, W)
    w, b = rnd(rng, Cout, C1 + C2, 3, 3) * 0.2, rnd(rng, Cout)
    OH, OW = H + 2 * pad - 2, W + 2 * pad - 2
    add = rnd(rng, B, Cout, OH + 3, OW + 2)
    conv = ops.Conv(w, b, pad=pad, relu=True)
    conv.wino = True
    full = host(conv(dev(x1), x2=dev(x2), add=dev(add), add_off=(2, 1), anchor=anchor))
    direct = ops.Conv(w, b, pad=pad, relu=True)
    direct.wino = False
    ref = host(direct(dev(x1), x2=dev(x2), add=dev(add), add_off=(2, 1)))
    assert np.abs(full - ref).max() <= 4 * conv_tol(ref, (C1 + C2) * 9)
    for (y0, x0, h, ww) in [(0, 0, OH, OW), (1, 1, 5, 7), (2, 3, 4, 4), (3, 0, OH - 3, 1),
                            (OH - 1, OW - 1, 1, 1), (0, 5, 2, OW - 5)]:
        win = host(conv(dev(x1), x2=dev(x2), add=dev(add), add_off=(2 + y0, 1 + x0),
                        window=(y0, x0, h, ww), anchor=anchor))
        assert np.array_equal(win, full[:, :, y0:y0 + h, x0:x0 + ww]), (y0, x0, h, ww)
        big = torch.full((B, Cout + 3, OH + 4, OW + 5), -7.0, device='cuda')
        conv(dev(x1), x2=dev(x2), add=dev(add), add_off=(2 + y0, 1 + x0), window=(y0, x0, h, ww),
             out=big, out_c0=2, place=(y0 + 1, x0 + 2), anchor=anchor)
        bigh = host(big)
        assert np.array_equal(bigh[:, 2:2 + Cout, y0 + 1:y0 + 1 + h, x0 + 2:x0 + 2 + ww], win)
        bigh[:, 2:2 + Cout, y0 + 1:y0 + 1 + h, x0 + 2:x0 + 2 + ww] = -7.0
        assert (bigh == -7.0).all(), 'wrote outside the placement window'


@pytest.mark.parametrize('shape', [(2, 5, 8, 8), (1, 3, 9, 7), (2, 70, 13, 13), (1, 1, 211, 5)])
def test_conv_fused_unpool(ops, shape):
    """DePool2D as the conv's input gather == oracle depool + conv; includes tie masks
    (post-ReLU zeros) and odd sizes (trailing row/col zero)."""
    B, C, H, W = shape
    rng = np.random.default_rng(H * 100 + W)
    pre = np.maximum(rnd(rng, B, C, H, W), 0)            # lots of exact-zero ties
    pre[:, :, :2, :2] = 0.75                              # a constant (all-tie) window
    pooled = onn.maxpool2(pre)
    up = rnd(rng, *pooled.shape)
    Cout = 12
    Wt, b = rnd(rng, Cout, C, 3, 3) / np.sqrt(9 * C), rnd(rng, Cout)
    un = onn.depool_eqmask(up.astype(np.float64), pre.astype(np.float64), pooled.astype(np.float64))
    ref = onn.conv2d(un, Wt.astype(np.float64), b.astype(np.float64), pad=1)
    got = host(ops.Conv(Wt, b, pad=1, relu=False)(dev(up), pre=dev(pre), pooled=dev(pooled)))
    assert np.abs(got - ref).max() <= conv_tol(ref, 9 * C)
    # and the materialised kernel is bit-exact
    un_gpu = host(ops.unpool_eqmask(dev(up), dev(pre), dev(pooled)))
    assert np.array_equal(un_gpu, un.astype(np.float32))


@pytest.mark.parametrize('fused_max', [0, 4096])
@pytest.mark.parametrize('shape,cout', [((2, 32, 13, 12), 40), ((1, 64, 9, 7), 64), ((2, 16, 10, 11), 130)])
def test_conv_winograd_fused_unpool(ops, shape, cout, fused_max, monkeypatch):
    """DePool2D applied by the Winograd input transform (all four patch-origin parities via
    windows and tile anchors) == oracle depool + conv == the window of the full launch."""
    monkeypatch.setattr(ops, 'WINO_FUSED_MAX_CIN', fused_max)
    B, C, H, W = shape
    rng = np.random.default_rng(H * 100 + W)
    pre = np.maximum(rnd(rng, B, C, H, W), 0)
    pre[:, :, :2, :2] = 0.75
    pooled = onn.maxpool2(pre)
    up = rnd(rng, *pooled.shape)
    Wt, b = rnd(rng, cout, C, 3, 3) / np.sqrt(9 * C), rnd(rng, cout)
    un = onn.depool_eqmask(up.astype(np.float64), pre.astype(np.float64), pooled.astype(np.float64))
    ref = onn.conv2d(un, Wt.astype(np.float64), b.astype(np.float64), pad=1)
    conv = ops.Conv(Wt, b, pad=1, relu=False)
    conv.wino = True
    for anchor in [(0, 0), (1, 0), (0, 1), (1, 1)]:
        full = host(conv(dev(up), pre=dev(pre), pooled=dev(pooled), anchor=anchor))
        assert conv._U is not None
        assert np.abs(full - ref).max() <= 4 * conv_tol(ref, 9 * C)
        for (y0, x0, h, w) in [(1, 2, 5, 4), (2, 1, H - 2, W - 1), (H - 1, W - 1, 1, 1)]:
            win = host(conv(dev(up), pre=dev(pre), pooled=dev(pooled), window=(y0, x0, h, w),
                            anchor=anchor))
            assert np.array_equal(win, full[:, :, y0:y0 + h, x0:x0 + w]), (anchor, y0, x0)


@pytest.mark.parametrize('case', [(2, 5, 21, 19, 40, 3), (1, 12, 16, 70, 100, 1), (2, 8, 9, 9, 64, 5)])
def test_conv_with_fused_pool_is_bit_exact(ops, case):
    """The 2x2 max-pool fused into the halo kernel's epilogue (iiseg_conv_pool_f32) == the pool
    kernel applied to the conv output, bit for bit, for the full map and for windows widened to
    whole pooling windows (`Conv.pool_window`); odd map sizes leave the last row / column unpooled."""
    B, Cin, H, W, Cout, pad = case
    rng = np.random.default_rng(hash(case) % 2**32)
    x, w, b = rnd(rng, B, Cin, H, W), rnd(rng, Cout, Cin, 3, 3) * 0.3, rnd(rng, Cout)
    conv = ops.Conv(w, b, pad=pad, relu=True)
    full = conv(dev(x))
    ref_pool = host(ops.maxpool2x2(full))
    fh, fw = full.shape[2], full.shape[3]
    if conv.pool_window(H, W) is None:
        pytest.skip('pool fusion switched off (IISEG_POOL_FUSE=0 / IISEG_CONV_HALO=0)')
    assert conv.pool_window(H, W) == (0, 0, fh, fw)
    pooled = torch.full((B, Cout, fh // 2, fw // 2), -3.0, device='cuda')
    out = conv(dev(x), pool_out=pooled)
    assert np.array_equal(host(out), host(full)) and np.array_equal(host(pooled), ref_pool)
    for region in [(3, 5, 4, 6), (0, 1, fh, 3), (fh - 3, fw - 4, 3, 4), (2, 2, 1, 1)]:
        win = conv.pool_window(H, W, region)
        y0, x0, h, ww = win
        assert y0 % 2 == 0 and x0 % 2 == 0 and y0 <= region[0] and y0 + h >= region[0] + region[2]
        buf = torch.full_like(full, -5.0)
        pooled = torch.full((B, Cout, fh // 2, fw // 2), -3.0, device='cuda')
        conv(dev(x), window=win, out=buf, place=(y0, x0), pool_out=pooled)
        bh, ph = host(buf), host(pooled)
        assert np.array_equal(bh[:, :, y0:y0 + h, x0:x0 + ww], host(full)[:, :, y0:y0 + h, x0:x0 + ww])
        py0, px0 = y0 // 2, x0 // 2
        py1, px1 = min((y0 + h) // 2, fh // 2), min((x0 + ww) // 2, fw // 2)
        assert np.array_equal(ph[:, :, py0:py1, px0:px1], ref_pool[:, :, py0:py1, px0:px1])
        ph[:, :, py0:py1, px0:px1] = -3.0
        assert (ph == -3.0).all(), 'pooled values written outside the window'


FULL_LAYERS = [
    # name, Cin, Cout, H, k, pad, window   -- real configs[1] layer shapes (batch 4)
    ('dae.conv2_1 halo', 64, 128, 211, 3, 1, (48, 48, 116, 116)),
    ('dae.up_conv1 halo16', 64, 11, 422, 3, 1, (99, 99, 224, 224)),
    ('dae.conv4_1 wino fused', 256, 512, 52, 3, 1, (10, 10, 33, 33)),
    ('dae.conv6_1 wino gemm', 1024, 2048, 13, 3, 1, (1, 1, 12, 12)),
    ('fcn.fc6 split-K gemm', 512, 4096, 13, 7, 0, None),
    ('fcn.score_pool4 1x1 taps', 512, 11, 26, 1, 0, None),
]


@pytest.mark.parametrize('case', FULL_LAYERS, ids=[c[0] for c in FULL_LAYERS])
def test_full_size_layers_are_exactly_homogeneous(ops, case):
    """Size-independent exactness at the real layer sizes (where the oracle takes minutes): a
    bias-free convolution is linear, and scaling by a power of two is exact in binary floating
    point, so conv(4 x) == 4 conv(x) and conv(x) with weights / 2 == conv(x) / 2 BIT FOR BIT, for
    every kernel family (direct, Winograd transforms included, split-K sums); and zero maps to
    zero."""
    name, Cin, Cout, H, k, pad, win = case
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn(4, Cin, H, H, device='cuda', generator=g)
    w = torch.randn(Cout, Cin, k, k, device='cuda', generator=g) / (Cin * k * k) ** 0.5
    kw = dict(window=win) if win else {}
    conv = ops.Conv(w, None, pad=pad, relu=False)
    y = conv(x, **kw)
    assert torch.isfinite(y).all() and float(y.abs().max()) > 0
    assert torch.equal(conv(4.0 * x, **kw), 4.0 * y)
    half = ops.Conv(0.5 * w, None, pad=pad, relu=False)
    assert torch.equal(half(x, **kw), 0.5 * y)
    assert not conv(torch.zeros_like(x), **kw).any()


def test_pool_unpool_round_trip_full_size(ops):
    """Round trip at the largest map of configs[1] (64 ch, 422x422): for a rectified map
    maxpool(DePool2D(up = pooled, pre, pooled)) == pooled exactly, the unpooled map equals `pre` on
    the mask and 0 elsewhere, and the window forms write exactly their window."""
    g = torch.Generator(device='cuda').manual_seed(3)
    pre = torch.relu(torch.randn(8, 64, 422, 422, device='cuda', generator=g))
    pooled = ops.maxpool2x2(pre)
    un = ops.unpool_eqmask(pooled, pre, pooled)
    assert torch.equal(ops.maxpool2x2(un), pooled)
    assert torch.equal(un, torch.where(un != 0, pre, torch.zeros_like(pre)))
    assert int((un != 0).sum()) >= int((pooled != 0).sum())          # >= : ties mark several
    win = (99, 98, 226, 227)
    out = torch.full_like(pre, -1.0)
    ops.unpool_eqmask(pooled, pre, pooled, out=out, window=win)
    y0, x0, h, w = win
    assert torch.equal(out[:, :, y0:y0 + h, x0:x0 + w], un[:, :, y0:y0 + h, x0:x0 + w])
    out[:, :, y0:y0 + h, x0:x0 + w] = -1.0
    assert bool((out == -1.0).all())
    pw = (49, 49, 113, 114)
    pout = torch.full_like(pooled, -1.0)
    ops.maxpool2x2(pre, out=pout, window=pw)
    assert torch.equal(pout[:, :, 49:49 + 113, 49:49 + 114], pooled[:, :, 49:49 + 113, 49:49 + 114])


@pytest.mark.parametrize('seed', range(24))
def test_conv3x3_randomised_against_oracle(ops, seed):
    """Seeded random 3x3 layers through whatever kernel the dispatch picks (halo 32-/16-row,
    Winograd GEMM / fused, static taps): random channel counts, pad, batch, optional two-source
    concat, DePool2D input, skip-add, ReLU, output window, placement into a larger tensor, tile
    anchor -- always against the float64 oracle of the same computation."""
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.integers(1, 4))
    Cin = int(rng.choice([3, 8, 11, 16, 32, 48, 64, 130]))
    Cout = int(rng.choice([5, 11, 16, 24, 64, 96, 130, 260]))
    H, W = int(rng.integers(6, 23)), int(rng.integers(6, 23))
    pad = int(rng.choice([0, 1, 1, 2, 7]))
    relu = bool(rng.integers(0, 2))
    unpool = bool(rng.integers(0, 3) == 0)
    two = (not unpool) and Cin >= 8 and bool(rng.integers(0, 3) == 0)
    w, b = rnd(rng, Cout, Cin, 3, 3) / np.sqrt(9 * Cin), rnd(rng, Cout)
    conv = ops.Conv(w, b, pad=pad, relu=relu)
    conv.wino = Cin % 16 == 0 and bool(rng.integers(0, 2))        # force either family
    kw = {}
    if unpool:
        pre = np.maximum(rnd(rng, B, Cin, H, W), 0)
        pooled = onn.maxpool2(pre)
        up = rnd(rng, *pooled.shape)
        x_ref = onn.depool_eqmask(up.astype(np.float64), pre.astype(np.float64),
                                  pooled.astype(np.float64))
        x_arg, kw = dev(up), dict(pre=dev(pre), pooled=dev(pooled))
    elif two:
        c1 = int(rng.choice([4, Cin // 2])) if Cin // 2 >= 4 else 4
        xa, xb = rnd(rng, B, c1, H, W), rnd(rng, B, Cin - c1, H, W)
        x_ref = np.concatenate([xa, xb], 1).astype(np.float64)
        x_arg, kw = dev(xa), dict(x2=dev(xb))
    else:
        x = rnd(rng, B, Cin, H, W)
        x_ref, x_arg = x.astype(np.float64), dev(x)
    ref = onn.conv2d(x_ref, w.astype(np.float64), b.astype(np.float64), pad=pad)
    OH, OW = ref.shape[2], ref.shape[3]
    y0, x0 = int(rng.integers(0, OH)), int(rng.integers(0, OW))
    h, ww = int(rng.integers(1, OH - y0 + 1)), int(rng.integers(1, OW - x0 + 1))
    if rng.integers(0, 2):
        add = rnd(rng, B, Cout, OH + 2, OW + 1)
        ref = ref + add[:, :, 1:1 + OH, 1:1 + OW]
        kw.update(add=dev(add), add_off=(1 + y0, 1 + x0))
    if relu:
        ref = np.maximum(ref, 0)
    big = torch.full((B, Cout + 1, OH + 3, OW + 2), -9.0, device='cuda')
    conv(x_arg, window=(y0, x0, h, ww), out=big, out_c0=1, place=(y0 + 2, x0 + 1),
         anchor=(int(rng.integers(0, 2)), int(rng.integers(0, 2))), **kw)
    got = host(big)
    sel = got[:, 1:, y0 + 2:y0 + 2 + h, x0 + 1:x0 + 1 + ww]
    assert np.abs(sel - ref[:, :, y0:y0 + h, x0:x0 + ww]).max() <= 4 * conv_tol(ref, 9 * Cin)
    got[:, 1:, y0 + 2:y0 + 2 + h, x0 + 1:x0 + 1 + ww] = -9.0
    assert (got == -9.0).all(), 'wrote outside the placement window'


@pytest.mark.parametrize('shape', [(2, 3, 8, 8), (1, 5, 9, 7), (3, 2, 211, 13), (1, 1, 2, 2)])
def test_maxpool_bit_exact(ops, shape):
    rng = np.random.default_rng(11)
    x = rnd(rng, *shape)
    assert np.array_equal(host(ops.maxpool2x2(dev(x))), onn.maxpool2(x))


@pytest.mark.parametrize('k,s,H,W', [(4, 2, 7, 7), (4, 2, 16, 9), (16, 8, 5, 6)])
def test_deconv_flip_and_crop_add(ops, k, s, H, W):
    """Asymmetric random kernels: checks the P3 flip convention, windows and the fused add."""
    rng = np.random.default_rng(k * 10 + s)
    x, Wt, b = rnd(rng, 2, 11, H, W), rnd(rng, 11, 11, k, k), rnd(rng, 11)
    ref = onn.deconv2d(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), stride=s)
    d = ops.Deconv(Wt, b, s)
    got = host(d(dev(x)))
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-5 * (1 + np.abs(ref).max())
    # cropped window + add of a center-cropped bigger tensor
    fh, fw = ref.shape[2:]
    oh, ow = fh - 3, fw - 2
    other = rnd(rng, 2, 11, fh + 4, fw + 1)
    ref2 = onn.center_crop(ref, oh, ow) + onn.center_crop(other.astype(np.float64), oh, ow)
    got2 = host(d(dev(x), add=dev(other), add_off=((fh + 4 - oh) // 2, (fw + 1 - ow) // 2),
                  window=((fh - oh) // 2, (fw - ow) // 2, oh, ow)))
    assert np.abs(got2 - ref2).max() <= 1e-5 * (1 + np.abs(ref2).max())


@pytest.mark.parametrize('dtype', ['f32', 'f64'])
@pytest.mark.parametrize('case', [  # k, s, B, Cin, Cout, H, W, window (None = full), add
    (16, 8, 3, 11, 11, 9, 11, None, False), (16, 8, 2, 11, 11, 9, 11, (19, 27, 40, 37), False),
    (16, 8, 2, 5, 16, 4, 3, (1, 2, 30, 21), False), (16, 8, 1, 11, 11, 31, 31, (31, 31, 224, 224), False),
    (4, 2, 3, 11, 11, 7, 9, None, False), (4, 2, 2, 11, 11, 7, 9, (3, 1, 11, 14), True),
    (4, 2, 2, 16, 13, 16, 5, (0, 5, 34, 6), True), (4, 2, 70, 11, 11, 3, 3, None, True)])
def test_deconv_output_phase_kernel(ops, case, dtype, monkeypatch):
    """K = 2 stride layers (the three FCN-8 upsamplers, models/fcn8.py:90,100,109) run on the output-phase
    kernel (csrc/deconv_phase.hip): exact on integer data against the oracle, and the same bits as the gather
    kernel on random data -- same sums in the same order -- for full maps, windows that start and end inside a
    phase period, a fused skip tensor (stride 2), float32 and float64."""
    k, s, B, Cin, Cout, H, W, window, with_add = case
    dt = torch.float32 if dtype == 'f32' else torch.float64
    npdt = np.float32 if dtype == 'f32' else np.float64
    rng = np.random.default_rng(k + H)
    fh, fw = (H - 1) * s + k, (W - 1) * s + k
    oy0, ox0, oh, ow = window if window is not None else (0, 0, fh, fw)
    for integer in (True, False):
        if integer:
            x = rng.integers(-3, 4, (B, Cin, H, W)).astype(npdt)
            Wt = rng.integers(-2, 3, (Cin, Cout, k, k)).astype(npdt)
            b = rng.integers(-2, 3, Cout).astype(npdt)
            other = rng.integers(-5, 6, (B, Cout, oh + 3, ow + 2)).astype(npdt)
        else:
            x, Wt, b = rnd(rng, B, Cin, H, W).astype(npdt), rnd(rng, Cin, Cout, k, k).astype(npdt), rnd(rng, Cout).astype(npdt)
            other = rnd(rng, B, Cout, oh + 3, ow + 2).astype(npdt)
        d = ops.Deconv(Wt, b, s, dtype=dt)
        kw = dict(window=(oy0, ox0, oh, ow))
        if with_add:
            kw.update(add=torch.from_numpy(other).cuda(), add_off=(2, 1))
        xd = torch.from_numpy(x).cuda()
        got = host(d(xd, **kw))
        assert d.last_form == 'phase'
        monkeypatch.setattr(ops, 'DECONV_PHASE', False)
        gather = host(d(xd, **kw))
        assert d.last_form == 'gather'
        monkeypatch.setattr(ops, 'DECONV_PHASE', True)
        assert np.array_equal(got, gather), np.abs(got - gather).max()
        if integer:
            ref = onn.deconv2d(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), stride=s)
            ref = ref[:, :, oy0:oy0 + oh, ox0:ox0 + ow]
            if with_add:
                ref = ref + other[:, :, 2:2 + oh, 1:1 + ow]
            assert np.array_equal(got.astype(np.float64), ref)
    # placement: a caller's output buffer is written inside the window only
    out = torch.full((B, Cout, oh, ow), -9.0, dtype=dt, device='cuda')
    d(xd, out=out, **kw)
    assert np.array_equal(host(out), got)


def test_deconv_21_classes(ops):
    """The reference's default n_classes=21 (models/fcn8.py:17): the gather-form transposed conv
    takes up to 32 channels (the 32-channel instantiation), k16 s8 and k4 s2."""
    rng = np.random.default_rng(21)
    for k, s, H, W in [(4, 2, 6, 5), (16, 8, 3, 4)]:
        x, Wt, b = rnd(rng, 2, 21, H, W), rnd(rng, 21, 21, k, k), rnd(rng, 21)
        ref = onn.deconv2d(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), stride=s)
        got = host(ops.Deconv(Wt, b, s)(dev(x)))
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-5 * (1 + np.abs(ref).max())
    with pytest.raises(RuntimeError, match='no kernel variant'):
        ops.Deconv(rnd(rng, 4, 33, 4, 4), rnd(rng, 33), 2)(dev(rnd(rng, 1, 4, 3, 3)))


def test_crop_softmax_and_residual(ops):
    rng = np.random.default_rng(3)
    score = 5 * rnd(rng, 2, 11, 30, 28)
    y = rng.random((2, 11, 24, 20)).astype(np.float32)
    ref = onn.softmax_channels(onn.center_crop(score.astype(np.float64), 24, 20))
    got = host(ops.crop_softmax(dev(score), 24, 20))
    assert np.abs(got - ref).max() <= 1e-6
    assert np.abs(got.sum(1) - 1).max() <= 1e-6
    de = host(ops.crop_softmax(dev(score), 24, 20, minuend=dev(y)))
    assert np.abs(de - (y - ref)).max() <= 1e-6


def test_refine_update_semantics(ops):
    """One fused step == reference arithmetic; frozen images do not move; the early-stop test
    is applied AFTER the update (iterative_inference.py:270-277)."""
    rng = np.random.default_rng(4)
    B, C, H, W = 3, 11, 19, 23
    score = 3 * rnd(rng, B, C, H + 4, W + 2)
    y0 = rng.random((B, C, H, W)).astype(np.float32)
    r = onn.softmax_channels(onn.center_crop(score.astype(np.float64), H, W))
    # image 2: y == r exactly representable? make score so that r ~ y: use y0[2] := r[2]
    y0[2] = r[2].astype(np.float32)
    step = 0.3
    de = y0.astype(np.float64) - r
    y_ref = np.clip(y0 - step * de, 0, 1)
    norms = np.linalg.norm(de, axis=1).mean(axis=(1, 2))
    y = dev(y0)
    st = ops.RefineState(B, H, W, y.device)
    st.active[1] = 0                                   # image 1 frozen beforehand
    ops.refine_update(dev(score), y, st, step)
    ops.refine_finalize(st, 1e-3)
    got = host(y)
    assert np.abs(got[0] - y_ref[0]).max() <= 1e-6
    assert np.array_equal(got[1], y0[1])               # frozen image untouched
    assert np.abs(got[2] - y_ref[2]).max() <= 1e-6     # updated in the iteration it converges
    active, iters, last = host(st.active), host(st.iters), host(st.last_norm)
    assert list(active) == [1, 0, 0] and list(iters) == [1, 0, 1]
    assert abs(last[0] - norms[0]) <= 1e-6 and last[2] < 1e-3


def test_confusion_matches_val_fn(ops):
    rng = np.random.default_rng(6)
    B, C, H, W = 2, 11, 16, 24
    y = rng.random((B, C, H, W)).astype(np.float32)
    y /= y.sum(1, keepdims=True)
    y[0, :, 0, 0] = 1.0 / C                             # an all-tie pixel: argmax -> first index
    labels = rng.integers(0, C + 1, size=(B, H, W))
    t = np.zeros((B, C + 1, H, W), dtype=np.float32)
    np.put_along_axis(t, labels[:, None], 1.0, axis=1)
    acc_ref, jacc_ref, mse_ref = ometrics.val_fn(y.astype(np.float64), t.astype(np.float64), C, [C])
    from iterative_inference_segm_amd.api import Metrics
    m = Metrics(C, 'cuda')
    ops.confusion_accumulate(dev(y), dev(t), m.cm, m.sums)
    acc, jacc, mse = m.result()
    assert np.array_equal(jacc, jacc_ref)               # integer counts: exact
    assert abs(acc - acc_ref) <= 1e-12
    assert abs(mse - mse_ref) <= 1e-6 * mse_ref


def test_abi_rejects_bad_arguments(ops):
    """Error behaviour: negative iiseg_status -> RuntimeError naming the cause."""
    rng = np.random.default_rng(1)
    conv = ops.Conv(rnd(rng, 4, 3, 3, 3), rnd(rng, 4), pad=1, relu=True)
    with pytest.raises(RuntimeError, match='input channels'):
        conv(dev(rnd(rng, 1, 5, 8, 8)))
    with pytest.raises(RuntimeError, match='shape'):
        conv(dev(rnd(rng, 1, 3, 8, 8)), window=(0, 0, 9, 9))      # window outside the output
    with pytest.raises(RuntimeError, match='device tensors'):
        conv(torch.zeros(1, 3, 8, 8))
    # the newer entry points through the raw C ABI: sizing queries answer 0 / "unsupported" for
    # requests their kernels do not take, calls return negative statuses instead of launching
    import ctypes as C
    from iterative_inference_segm_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.B, d.C1, d.C2, d.H, d.W, d.Cout, d.KH, d.KW, d.pad, d.dil = 1, 20, 0, 8, 8, 32, 3, 3, 1, 1
    d.OH, d.OW = 8, 8
    assert lib.iiseg_conv_plan(C.byref(d)) == 0
    assert lib.iiseg_conv_wino_supported(C.byref(d)) == 0          # 20 channels: not a multiple of 16
    assert lib.iiseg_conv_wino_weight_elems(C.byref(d)) == 0
    assert lib.iiseg_conv_wino_f32(None, C.byref(d), None, None, None, None, None, None, None, None,
                                   None, 7) == -5                    # IISEG_ERR_UNSUPPORTED
    d.C1 = 32
    assert lib.iiseg_conv_plan(C.byref(d)) == 0 and lib.iiseg_conv_wino_supported(C.byref(d)) == 1
    assert lib.iiseg_conv_wino_f32(None, C.byref(d), None, None, None, None, None, None, None, None,
                                   None, 7) == -1                    # IISEG_ERR_NULL
    d.tile_y0 = 2
    assert lib.iiseg_conv_wino_supported(C.byref(d)) == 0          # anchor parity must be 0 / 1
    d.tile_y0 = 0
    assert lib.iiseg_conv_gemm_supported(C.byref(d)) == 0          # padded conv: not the GEMM form
    d.oy0, d.OH = 1, 7
    assert lib.iiseg_conv_pool_supported(C.byref(d)) == 0          # odd window origin
    d.oy0, d.OH = 0, 8
    import os
    if os.environ.get('IISEG_CONV_HALO', '1') != '0':
        assert lib.iiseg_conv_pool_supported(C.byref(d)) == 1
    assert lib.iiseg_strerror(-5).decode().startswith('no kernel variant')
    assert lib.iiseg_bn_stats_workspace_elems(16) == 16 * 64 * 2
    assert lib.iiseg_depool_bwd_f32(None, None, None, None, None, 1, 4, 4) == -1
    assert lib.iiseg_add_noise_f32(None, None, None, 0.1, None, 10) == -1


def _eq_bits(pre, pooled):
    """Byte form of the DePool2D mask (include/iiseg.h, iiseg_conv_mask_f32) from host arrays."""
    B, Cc, H, W = pre.shape
    h2, w2 = H // 2, W // 2
    m = np.zeros((B, Cc, h2, w2), np.uint8)
    for dy in (0, 1):
        for dx in (0, 1):
            eq = pre[:, :, dy:2 * h2:2, dx:2 * w2:2] == pooled
            m |= (eq.astype(np.uint8) << (dy * 2 + dx))
    return m


@pytest.mark.parametrize('mma', ['f32', 'bf16'])
@pytest.mark.parametrize('case', [(2, 5, 21, 19, 40, 3), (1, 12, 16, 70, 100, 1), (2, 8, 9, 9, 64, 5)])
def test_conv_pool_mask_bytes_are_exact(ops, case, mma):
    """Encoder side of the byte-mask DePool2D: conv + pool + mask_out with the pre-pool map NOT
    stored gives the same pool as the pool-fused conv and exactly the bytes of pre == pooled; a
    window writes only its pooling windows (full map and LICM-style windows)."""
    B, Cin, H, W, Cout, pad = case
    rng = np.random.default_rng(hash(case) % 2**32)
    x, w, b = rnd(rng, B, Cin, H, W), rnd(rng, Cout, Cin, 3, 3) * 0.3, rnd(rng, Cout)
    conv = ops.Conv(w, b, pad=pad, relu=True, mma=mma)
    if not (conv.pool_fusable() and conv.mask_ok()):
        pytest.skip('halo kernels switched off')
    fh, fw = conv.out_hw(H, W)
    pooled = torch.empty((B, Cout, fh // 2, fw // 2), device='cuda')
    full = conv(dev(x), pool_out=pooled)
    ref_mask = _eq_bits(host(full), host(pooled))
    pool2 = torch.full_like(pooled, -3.0)
    mask = torch.full(pooled.shape, 0xAA, dtype=torch.uint8, device='cuda')
    assert conv(dev(x), pool_out=pool2, mask_out=mask, store_out=False) is None
    assert np.array_equal(host(pool2), host(pooled))
    assert np.array_equal(mask.cpu().numpy(), ref_mask)
    # with the pre-pool map stored as well
    mask.fill_(0x55)
    out = conv(dev(x), pool_out=pool2, mask_out=mask)
    assert np.array_equal(host(out), host(full)) and np.array_equal(mask.cpu().numpy(), ref_mask)
    for region in [(3, 5, 4, 6), (0, 1, fh, 3), (fh - 3, fw - 4, 3, 4), (2, 2, 1, 1)]:
        y0, x0, h, ww = win = conv.pool_window(H, W, region)
        pool2 = torch.full_like(pooled, -3.0)
        mask = torch.full(pooled.shape, 0xAA, dtype=torch.uint8, device='cuda')
        shape_only = torch.empty(full.shape, device='meta')
        conv(dev(x), window=win, out=shape_only, place=(y0, x0), pool_out=pool2, mask_out=mask,
             store_out=False)
        exp_p = np.full(pooled.shape, -3.0, np.float32)
        exp_m = np.full(pooled.shape, 0xAA, np.uint8)
        q = (slice(None), slice(None), slice(y0 // 2, (y0 + h) // 2), slice(x0 // 2, (x0 + ww) // 2))
        exp_p[q], exp_m[q] = host(pooled)[q], ref_mask[q]
        assert np.array_equal(host(pool2), exp_p), region
        assert np.array_equal(mask.cpu().numpy(), exp_m), region


@pytest.mark.parametrize('mma', ['f32', 'bf16'])
@pytest.mark.parametrize('case', [(2, 24, 20, 26, 40), (1, 64, 33, 41, 21), (2, 16, 14, 15, 8),
                                  (1, 40, 64, 70, 64)])
def test_unpool_conv_from_mask_bytes_is_bit_identical(ops, case, mma, monkeypatch):
    """Decoder side: the conv over the DePool2D of `up` takes the mask as bytes (mask_in) and gives
    bit for bit what it gives from pre / pooled -- 32-row, 16-row (<= 16 output channels) and bf16
    halo kernels, full maps, decoder windows, with and without the skip add; ties and all-equal
    windows (several bits set) included.  (bf16 mode, <= 16 output channels: by default the pre /
    pooled form of such a layer runs the fp32 16-row kernel and only the byte form the bf16 one --
    ops.py; here both are put on the bf16 kernel so that one kernel is compared with itself.)"""
    monkeypatch.setattr(ops, 'BF16_UPCONV1', True)
    B, Cc, H, W, Cout = case
    rng = np.random.default_rng(hash(case) % 2**32)
    pre = np.maximum(rnd(rng, B, Cc, H, W), 0)                  # post-ReLU map: many exact ties at 0
    h2, w2 = H // 2, W // 2
    pooled = pre[:, :, :2 * h2, :2 * w2].reshape(B, Cc, h2, 2, w2, 2).max(axis=(3, 5))
    up = rnd(rng, B, Cc, h2, w2)
    w, b = rnd(rng, Cout, Cc, 3, 3) * 0.3, rnd(rng, Cout)
    conv = ops.Conv(w, b, pad=1, relu=False, mma=mma)
    if not conv.mask_ok():
        pytest.skip('halo kernels switched off')
    mask = torch.from_numpy(_eq_bits(pre, pooled)).cuda()
    assert int((mask.cpu().numpy() == 15).sum()) > 0             # all-zero windows: every bit set
    ref = conv(dev(up), pre=dev(pre), pooled=dev(pooled))
    got = conv(dev(up), mask_in=mask, unpool_hw=(H, W))
    assert np.array_equal(host(got), host(ref))
    add = rnd(rng, B, Cout, H + 3, W + 2)
    for (y0, x0, h, ww) in [(1, 2, 9, 11), (0, 0, H, 5), (H - 4, W - 7, 4, 7), (3, 3, 1, 1)]:
        kw = dict(window=(y0, x0, h, ww), add=dev(add), add_off=(y0 + 1, x0))
        ref = conv(dev(up), pre=dev(pre), pooled=dev(pooled), **kw)
        got = conv(dev(up), mask_in=mask, unpool_hw=(H, W), **kw)
        assert np.array_equal(host(got), host(ref)), (y0, x0, h, ww)


def test_mask_bytes_are_refused_off_the_halo_kernels(ops):
    """Layers that do not run on a halo kernel (Winograd, >= 256 output channels) say so instead
    of silently taking another path."""
    rng = np.random.default_rng(3)
    conv = ops.Conv(rnd(rng, 256, 32, 3, 3), rnd(rng, 256), pad=1, relu=False)
    assert not conv.mask_ok()
    up = dev(rnd(rng, 1, 32, 4, 4))
    with pytest.raises(RuntimeError):
        conv(up, mask_in=torch.zeros((1, 32, 4, 4), dtype=torch.uint8, device='cuda'), unpool_hw=(8, 8))


def test_dispatch_timing_of_launches(built_lib):
    """Launch profiling (include/iiseg.h iiseg_profile_begin / _end): every launch of the library between
    the two calls carries start / stop events on its dispatch; the per-launch times are positive, their
    count is the number of launches, and their sum is not larger than a bracket of separately recorded
    events around the same launches (which also holds the gaps between the kernels)."""
    import torch
    from iterative_inference_segm_amd import _lib, ops
    lib = _lib.load()
    assert lib.iiseg_profile_count() == -1
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.rand(4, 64, 60, 60, device='cuda', generator=g)
    W = torch.randn(64, 64, 3, 3, device='cuda', generator=g) * 0.05
    conv = ops.Conv(W, None, pad=1, relu=True)
    conv(x); conv(x)
    torch.cuda.synchronize()
    ops.profile_begin()
    ops.CONV_PROFILE = prof = []
    try:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            conv(x)
        p = ops.maxpool2x2(x)
        e1.record()
        torch.cuda.synchronize()
    finally:
        ops.CONV_PROFILE = None
        n = ops.profile_end()
    assert lib.iiseg_profile_count() == -1
    assert n == 6 and len(ops.PROFILE_MS) == 6 and all(t > 0 for t in ops.PROFILE_MS)
    per_conv = [s.elapsed_time(e) for _, _, s, e in prof]
    assert len(per_conv) == 5 and all(t > 0 for t in per_conv)
    assert abs(sum(per_conv) - sum(ops.PROFILE_MS[:5])) < 1e-6
    assert sum(ops.PROFILE_MS) <= e0.elapsed_time(e1) * 1.05
    assert tuple(p.shape) == (4, 64, 30, 30)


SMALL_CASES = [  # B, Cin, H, W, Cout, k, pad, dil, relu, window, place
    (2, 11, 70, 130, 11, 3, 0, 1, True, None, False),      # context module: 'valid' 11 -> 11
    (1, 11, 90, 100, 11, 3, 0, 16, True, None, False),     # dilation 16
    (2, 11, 40, 75, 11, 3, 0, 8, True, None, False),
    (3, 11, 31, 67, 11, 1, 0, 1, False, None, False),      # the 1x1 output layer, ragged last pixel group
    (1, 16, 20, 66, 16, 3, 0, 2, False, (3, 2, 9, 57), True),    # window + placement, 16 channels
    (1, 3, 19, 19, 5, 3, 0, 4, True, None, False),         # tiny map
    (2, 11, 21, 31, 11, 3, 0, 1, True, None, False),       # ragged last pixel group at the end of the buffer
    (2, 11, 23, 37, 11, 3, 0, 2, False, None, False),
]


@pytest.mark.parametrize('case', SMALL_CASES)
def test_conv_small_channels_on_the_vector_alu(ops, case):
    """Layers between at most 16 channels on either side (csrc/conv_small.hip: the context module,
    models/contextmod_dae.py:74-105) on the vector ALU: equal to the oracle on integer data (every product and
    partial sum exact in fp32), within the fp32 conv tolerance on random data; window and placement."""
    import ctypes as C
    B, Cin, H, W, Cout, k, pad, dil, relu, window, place = case
    rng = np.random.default_rng(sum(c for c in case[:9]))
    conv_i = ops.Conv(rng.integers(-2, 3, size=(Cout, Cin, k, k)).astype(np.float32),
                      rng.integers(-4, 5, size=Cout).astype(np.float32), pad=pad, relu=relu, dil=dil)
    d, _, _ = conv_i._plan(B, Cin, 0, H, W, window, None, False)
    assert conv_i.lib.iiseg_conv_small_supported(C.byref(d)) == 1
    d1 = type(d).from_buffer_copy(d)
    d1.pad = 1                              # (zero-padded layers stay on the halo kernel)
    assert conv_i.lib.iiseg_conv_small_supported(C.byref(d1)) == 0
    xi = rng.integers(-3, 4, size=(B, Cin, H, W)).astype(np.float32)
    ref = onn.conv2d(xi.astype(np.float64), host(conv_i.W).astype(np.float64), host(conv_i.b).astype(np.float64),
                     pad=pad, dilation=dil, relu=relu)
    kw = {}
    if window is not None:
        y0, x0, oh, ow = window
        ref = ref[:, :, y0:y0 + oh, x0:x0 + ow]
        kw['window'] = window
    if place:
        big = torch.full((B, Cout, ref.shape[2] + 3, ref.shape[3] + 2), 7.0, device='cuda')
        conv_i(dev(xi), out=big, place=(2, 1), **kw)
        hb = host(big)
        assert np.array_equal(hb[:, :, 2:2 + ref.shape[2], 1:1 + ref.shape[3]], ref.astype(np.float32))
        hb[:, :, 2:2 + ref.shape[2], 1:1 + ref.shape[3]] = 7.0
        assert np.all(hb == 7.0)
    else:
        got = host(conv_i(dev(xi), **kw))
        assert got.shape == ref.shape and np.array_equal(got, ref.astype(np.float32))
    # random data: fp32 sequential FMAs against float64
    x, Wt, b = rnd(rng, B, Cin, H, W), rnd(rng, Cout, Cin, k, k) / np.sqrt(Cin * k * k), rnd(rng, Cout)
    ref = onn.conv2d(x.astype(np.float64), Wt.astype(np.float64), b.astype(np.float64), pad=pad, dilation=dil,
                     relu=relu)
    if window is not None:
        ref = ref[:, :, y0:y0 + oh, x0:x0 + ow]
    got = host(ops.Conv(Wt, b, pad=pad, relu=relu, dil=dil)(dev(x), **kw))
    assert np.abs(got - ref).max() <= conv_tol(ref, Cin * k * k)
