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
