"""CPU: the oracle's hand-written backward of the DAE reconstruction error (oracle/dae_grad.py, the
checker of the true-gradient extension mode) is pinned by central finite differences in float64."""
import numpy as np
import pytest

from iterative_inference_segm_amd import synthetic as S
from oracle import dae_grad as G

CASES = [
    dict(concat_h=['pool1'], additional_pool=1, conv_before_pool=1, skip=True),
    dict(concat_h=['pool1', 'pool2'], additional_pool=1, conv_before_pool=2, skip=True),
    dict(concat_h=['input'], additional_pool=2, conv_before_pool=1, skip=False),
]


@pytest.mark.parametrize('cfg', CASES)
def test_sqerr_gradient_matches_finite_differences(cfg):
    rng = np.random.default_rng(3)
    pad, nf, B, Hh, Ww = 3, 2, 1, 12, 10
    chans = {'input': 3, 'pool1': 3, 'pool2': 4}
    hch = tuple(chans[c] for c in cfg['concat_h'])
    dp = S.make_dae_params(h_channels=hch, concat_h=cfg['concat_h'], n_filters=nf,
                           conv_before_pool=cfg['conv_before_pool'],
                           additional_pool=cfg['additional_pool'], seed=5)
    dp = {k: tuple(np.asarray(a, np.float64) for a in v) for k, v in dp.items()}
    y = rng.random((B, 11, Hh, Ww)); y /= y.sum(1, keepdims=True)
    one = len(cfg['concat_h']) == 1 and cfg['concat_h'][-1] != 'input'    # the pad rule applies
    size = (Hh + (2 * pad - 2 if one else 0), Ww + (2 * pad - 2 if one else 0))
    hs = []
    for c in cfg['concat_h']:
        k = 0 if c == 'input' else int(c[-1])
        shp = (Hh, Ww) if c == 'input' else (size[0] >> k, size[1] >> k)
        hs.append(rng.random((B, chans[c]) + shp))
    kw = dict(padding=pad, n_filters=nf, **cfg)
    g, r = G.dae_sqerr_grad(dp, hs, y, **kw)
    assert g.shape == y.shape and np.abs(g).max() > 1e-2
    e = 1e-6
    for _ in range(10):
        idx = tuple(rng.integers(0, s) for s in y.shape)
        yp, ym = y.copy(), y.copy()
        yp[idx] += e
        ym[idx] -= e
        fd = (G.sqerr(dp, hs, yp, **kw) - G.sqerr(dp, hs, ym, **kw)) / (2 * e)
        assert abs(fd - g[idx]) <= 1e-6 * (1 + abs(fd)), (idx, fd, g[idx])
