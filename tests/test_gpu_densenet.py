"""GPU parity of the FC-DenseNet host path: batch-statistics BN, concat-free dense
blocks (conv output slices), the stride-2 transposed 3x3 conv, and the whole network followed by
the standard DAE (padding=0, h = pool4) -- float32 within tolerance, float64 strictly."""
import numpy as np
import pytest
import torch

from oracle import dae as odae, densenet as oden, nn as onn, refine as orefine
from iterative_inference_segm_amd import synthetic as S

pytestmark = pytest.mark.gpu


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def p64(params):
    return [{k: (np.asarray(v, np.float64) if k != 'kind' else v) for k, v in p.items()}
            for p in params]


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-5), (torch.float64, 1e-12)])
def test_bn_ops_and_transposed_conv(built_lib, dtype, tol):
    from iterative_inference_segm_amd import ops
    rng = np.random.default_rng(0)
    npd = np.float64
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=npd)).to(dtype).cuda()
    # statistics + apply on a channel-slice view of a wider stack
    buf = rng.standard_normal((3, 10, 9, 7)) * 2 + 0.5
    C = 6
    mean = torch.zeros(10, dtype=dtype, device='cuda'); inv = torch.zeros(10, dtype=dtype, device='cuda')
    ops.bn_stats(dev(buf), 0, 4, mean, inv)
    ops.bn_stats(dev(buf), 4, 2, mean, inv)                       # incremental slice
    m_ref = buf[:, :C].mean(axis=(0, 2, 3)); v_ref = buf[:, :C].var(axis=(0, 2, 3))
    assert np.abs(host(mean)[:C] - m_ref).max() <= tol * 10
    assert np.abs(host(inv)[:C] - 1 / np.sqrt(v_ref + 1e-4)).max() <= tol * 10
    beta, gamma = rng.standard_normal(C), rng.uniform(0.5, 1.5, C)
    got = host(ops.bn_relu(dev(buf), C, dev(beta), dev(gamma), mean[:C].contiguous(),
                           inv[:C].contiguous()))
    ref = oden._bn_relu(buf[:, :C], beta, gamma)
    assert np.abs(got - ref).max() <= tol * 20
    # transposed 3x3 stride-2 conv (P3 flip) with a center-crop window into an output slice
    x, Wt, b = rng.standard_normal((2, 20, 7, 6)), rng.standard_normal((20, 12, 3, 3)), rng.standard_normal(12)
    full = onn.deconv2d(x, Wt, b, stride=2)                        # (2,12,15,13)
    conv = ops.Conv(Wt, b, pad=0, relu=False, layout='iohw', transposed=True, dtype=dtype)
    assert np.abs(host(conv(dev(x))) - full).max() <= tol * 50
    out = torch.full((2, 20, 14, 12), -7.0, dtype=dtype, device='cuda')
    conv(dev(x), window=(0, 0, 14, 12), out=out, out_c0=5)
    o = host(out)
    assert np.abs(o[:, 5:17] - full[:, :, :14, :12]).max() <= tol * 50
    assert np.all(o[:, :5] == -7.0) and np.all(o[:, 17:] == -7.0)  # neighbours untouched


def _small(dtype):
    from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan
    nl = [2, 3, 2, 2, 2, 2, 2, 2, 2, 3, 2]
    plan = layer_plan(n_layers_per_block=nl, n_first=8, growth=4)
    params = S.make_densenet_params(plan, seed=7)
    net = FCDenseNet(params, 11, layer=['pool4'], n_layers_per_block=nl, growth=4, dtype=dtype)
    return nl, params, net


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-4), (torch.float64, 1e-10)])
def test_densenet_forward_small(built_lib, dtype, tol):
    nl, params, net = _small(dtype)
    X = S.make_images(3, 64, 96, seed=8)
    h, y = net(torch.from_numpy(X).to(dtype).cuda())
    h_ref, y_ref = oden.densenet_forward(p64(params), X.astype(np.float64), layer=['pool4'],
                                         n_layers_per_block=nl, growth=4)
    assert h.shape == h_ref.shape == (3, 8 + 4 * (2 + 3 + 2 + 2), 4, 6)
    assert np.abs(host(h) - h_ref).max() <= tol * (1 + np.abs(h_ref).max())
    assert np.abs(host(y) - y_ref).max() <= tol


def test_densenet_host_with_standard_dae_f64(built_lib):
    """segm_net='densenet' wiring of iterative_inference.py:140-143,151-164: padding=0, h = the
    pool4 stack concatenated after the DAE's pool4; 3 refinement steps, strict (float64)."""
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    dt = torch.float64
    nl, params, net = _small(dt)
    hc = 8 + 4 * 9
    dp = S.make_dae_params(h_channels=(hc,), n_filters=4, additional_pool=1, seed=9)
    dae = StandardDAE(dp, 11, concat_h=['pool4'], padding=0, n_filters=4, additional_pool=1, dtype=dt)
    ii = IterativeInference(net, dae, 11, [11], dtype=dt)
    X = S.make_images(2, 64, 64, seed=10)
    out = ii.pred_fcn_fn(X)
    H, Y = out[:-1], out[-1]
    h_ref, y_ref = oden.densenet_forward(p64(params), X.astype(np.float64), layer=['pool4'],
                                         n_layers_per_block=nl, growth=4)
    dp64 = {k: tuple(np.asarray(a, np.float64) for a in v) for k, v in dp.items()}
    dae_fn = lambda hh, yy: odae.dae_forward(dp64, hh, yy, padding=0, n_filters=4, additional_pool=1)
    yii_ref, it_ref = orefine.refine_batch(dae_fn, [h_ref], y_ref, 0.2, 3)
    Yii, iters, _ = ii.refine(H, Y, 0.2, 3)
    assert list(host(iters)) == list(it_ref)
    assert np.abs(host(Yii) - yii_ref).max() <= 1e-9


def test_fused_bnrelu_conv_equals_two_kernels(built_lib, monkeypatch):
    """The opt-in BN_ReLU_Conv kernel (iiseg_conv_bnrelu_f32, IISEG_BNRELU_FUSE=1) gives bit for bit
    the result of iiseg_bn_relu_f32 followed by the conv: same arithmetic, zero padding applied to
    the normalised map, input = the first n channels of a wider stack, output = a channel slice."""
    from iterative_inference_segm_amd import ops
    monkeypatch.setattr(ops, 'BNRELU_FUSE', True)
    rng = np.random.default_rng(5)
    B, cap, n, H, W, Cout = 3, 40, 22, 19, 37, 16
    stack = torch.from_numpy(rng.standard_normal((B, cap, H, W)).astype(np.float32)).cuda()
    mean = torch.zeros(cap, device='cuda'); inv = torch.zeros(cap, device='cuda')
    ops.bn_stats(stack, 0, n, mean, inv)
    beta = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, n).astype(np.float32)).cuda()
    w = rng.standard_normal((Cout, n, 3, 3)).astype(np.float32) / np.sqrt(9 * n)
    conv = ops.Conv(w, rng.standard_normal(Cout).astype(np.float32), pad=1, relu=False)
    out_f = torch.full((B, cap, H, W), -2.0, device='cuda')
    assert conv.bnrelu_conv(stack, n, (beta, gamma, mean, inv), out_f, 20) is not None
    out_r = torch.full((B, cap, H, W), -2.0, device='cuda')
    conv(ops.bn_relu(stack, n, beta, gamma, mean, inv), out=out_r, out_c0=20)
    assert torch.equal(out_f, out_r)
    assert bool((out_f[:, :20] == -2.0).all()) and bool((out_f[:, 36:] == -2.0).all())
