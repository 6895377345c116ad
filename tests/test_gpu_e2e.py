"""GPU parity, end to end: pred_fcn_fn / pred_dae_fn / de_fn / refine on the HIP path against the
float64 oracle with identical seeded weights and inputs.  north_star tolerance: 1e-4 max-abs on
the refined softmax map (fp32 HIP vs float64 oracle)."""
import numpy as np
import pytest
import torch

from oracle import dae as odae
from oracle import fcn8 as ofcn8
from oracle import refine as orefine
from iterative_inference_segm_amd import synthetic as S

pytestmark = pytest.mark.gpu

TOL = 1e-4   # BASELINE.json north_star: within 1e-4 on the refined softmax map


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def build(built_lib, fcn_params, dae_params, concat_h, n_filters, pad=100, **dae_kw):
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fcn = FCN8(fcn_params, 11, layer=list(concat_h) + ['probs_dimshuffle'], pad=pad)
    dae = StandardDAE(dae_params, 11, concat_h=concat_h, padding=pad, n_filters=n_filters, **dae_kw)
    return IterativeInference(fcn, dae, 11, void_labels=[11])


def to64(p):
    return {k: tuple(np.asarray(a, dtype=np.float64) for a in v) for k, v in p.items()}


def test_small_fcn8_dae_refine(built_lib):
    """Scaled-down FCN-8 (width/16) + standard DAE (n_filters=4), real pad-100 geometry,
    48x40 images, 4 refinement steps, batch of 3 (per-image oracle loop vs batched HIP loop)."""
    concat_h = ['pool4']
    fp = S.make_fcn8_params(width_div=16, fc_channels=64, seed=11)
    X = S.make_images(3, 48, 40, seed=5)
    dp = S.make_dae_params(h_channels=(fp['conv4_3'][0].shape[0],), n_filters=4, seed=12)
    ii = build(built_lib, fp, dp, concat_h, 4)

    h_ref, y_ref = ofcn8.fcn8_forward(to64(fp), X.astype(np.float64),
                                      layer=concat_h + ['probs_dimshuffle'])
    out = ii.pred_fcn_fn(X)
    H, Y = out[:-1], out[-1]
    assert np.abs(host(H[0]) - h_ref).max() <= 1e-4 * (1 + np.abs(h_ref).max())
    assert np.abs(host(Y) - y_ref).max() <= TOL

    dae_fn = lambda hh, yy: odae.dae_forward(to64(dp), hh, yy, concat_h=concat_h, n_filters=4)
    r_ref = dae_fn([h_ref], y_ref)
    R = ii.pred_dae_fn(*(H + [Y]))
    assert np.abs(host(R) - r_ref).max() <= TOL
    assert np.abs(host(ii.de_fn(*(H + [Y]))) - (y_ref - r_ref)).max() <= TOL

    yii_ref, it_ref = orefine.refine_batch(dae_fn, [h_ref], y_ref, 0.1, 4)
    Yii, iters, _ = ii.refine(H, Y, 0.1, 4)
    assert list(host(iters)) == list(it_ref)
    assert np.abs(host(Yii) - yii_ref).max() <= TOL


def test_early_stop_freezes_per_image(built_lib):
    """Per-image early stop (F3): with a large eps every image stops after its first step; the
    batched loop must return exactly the 1-step result although it keeps iterating."""
    concat_h = ['pool3']
    fp = S.make_fcn8_params(width_div=16, fc_channels=32, seed=21)
    dp = S.make_dae_params(h_channels=(fp['conv3_3'][0].shape[0],), concat_h=concat_h,
                           n_filters=4, additional_pool=1, seed=22)
    ii = build(built_lib, fp, dp, concat_h, 4, additional_pool=1)
    X = S.make_images(2, 32, 32, seed=6)
    out = ii.pred_fcn_fn(X)
    H, Y = out[:-1], out[-1]
    y1, it1, _ = ii.refine(H, Y, 0.5, 1, early_stop=False)
    y5, it5, norms = ii.refine(H, Y, 0.5, 5, eps=1e9)
    assert list(host(it5)) == [1, 1]
    assert np.array_equal(host(y1), host(y5))
    # and the oracle agrees on the iteration count semantics
    dae_fn = lambda hh, yy: odae.dae_forward(to64(dp), hh, yy, concat_h=concat_h, n_filters=4,
                                             additional_pool=1)
    _, it_ref = orefine.refine_batch(dae_fn, [host(H[0]).astype(np.float64)],
                                     host(Y).astype(np.float64), 0.5, 5, eps=1e9)
    assert list(it_ref) == [1, 1]


def test_full_size_single_image(built_lib):
    """BASELINE config 1/2 geometry: real FCN-8 + 64-filter DAE at 224x224, 11 classes, one
    image, 3 refinement steps (the oracle needs ~25 s for this on the box's host cores)."""
    concat_h = ['pool4']
    fp, dp = S.make_fcn8_params(), S.make_dae_params()
    ii = build(built_lib, fp, dp, concat_h, 64)
    X = S.make_images(1, 224, 224, seed=1234)
    out = ii.pred_fcn_fn(X)
    H, Y = out[:-1], out[-1]
    h_ref, y_ref = ofcn8.fcn8_forward(to64(fp), X.astype(np.float64),
                                      layer=concat_h + ['probs_dimshuffle'])
    assert np.abs(host(H[0]) - h_ref).max() <= 1e-4 * (1 + np.abs(h_ref).max())
    assert np.abs(host(Y) - y_ref).max() <= TOL
    dp64 = to64(dp)
    dae_fn = lambda hh, yy: odae.dae_forward(dp64, hh, yy)
    yii_ref, it_ref = orefine.refine_batch(dae_fn, [h_ref], y_ref, 0.1, 3)
    Yii, iters, _ = ii.refine(H, Y, 0.1, 3)
    err = np.abs(host(Yii) - yii_ref)
    print('full-size refine max-abs err %.3e, frac>1e-5 %.2e' % (err.max(), (err > 1e-5).mean()))
    assert list(host(iters)) == list(it_ref)
    assert err.max() <= TOL
