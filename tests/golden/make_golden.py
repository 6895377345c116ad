"""Generates the committed golden vectors with the float64 oracle (run in the build container:
`python tests/golden/make_golden.py`).  The reference itself cannot run anywhere reachable
(Python 2 + Theano), so these are oracle outputs on seeded synthetic weights and
inputs -- data only: seeds, inputs and expected outputs; no reference source.

  mini_e2e.npz : scaled-down FCN-8 (width/16, fc 64) + standard DAE (n_filters=4), real
                 pad-100 geometry, 3 images 48x40, 4 refinement steps (step 0.1).
  full224.npz  : the BASELINE config-1/2 network (real FCN-8 + 64-filter DAE) on one 224x224
                 image: FCN softmax map and one DAE reconstruction, stored subsampled [::4, ::4]
                 (float32) plus the SHA-256 of the full float32 arrays (oracle regression pin).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import dae as odae, fcn8 as ofcn8, refine as orefine  # noqa: E402
from iterative_inference_segm_amd import synthetic as S  # noqa: E402

to64 = lambda p: {k: tuple(np.asarray(a, np.float64) for a in v) for k, v in p.items()}
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()

MINI = dict(fcn_seed=11, dae_seed=12, img_seed=5, width_div=16, fc_channels=64, n_filters=4,
            n=3, h=48, w=40, step=0.1, num_iter=4)
FULL = dict(fcn_seed=1234, dae_seed=4321, img_seed=1234)


def mini():
    c = MINI
    fp = S.make_fcn8_params(width_div=c['width_div'], fc_channels=c['fc_channels'], seed=c['fcn_seed'])
    dp = S.make_dae_params(h_channels=(fp['conv4_3'][0].shape[0],), n_filters=c['n_filters'],
                           seed=c['dae_seed'])
    X = S.make_images(c['n'], c['h'], c['w'], seed=c['img_seed'])
    h, y = ofcn8.fcn8_forward(to64(fp), X.astype(np.float64), layer=['pool4', 'probs_dimshuffle'])
    dp64 = to64(dp)
    dae_fn = lambda hh, yy: odae.dae_forward(dp64, hh, yy, n_filters=c['n_filters'])
    r = dae_fn([h], y)
    yii, iters = orefine.refine_batch(dae_fn, [h], y, c['step'], c['num_iter'])
    np.savez_compressed(os.path.join(HERE, 'mini_e2e.npz'), h=h.astype(np.float32),
                        y=y.astype(np.float32), r=r.astype(np.float32),
                        yii=yii.astype(np.float32), iters=iters,
                        **{'cfg_' + k: v for k, v in c.items()})


def full():
    c = FULL
    fp, dp = S.make_fcn8_params(seed=c['fcn_seed']), S.make_dae_params(seed=c['dae_seed'])
    X = S.make_images(1, 224, 224, seed=c['img_seed'])
    h, y = ofcn8.fcn8_forward(to64(fp), X.astype(np.float64), layer=['pool4', 'probs_dimshuffle'])
    r = odae.dae_forward(to64(dp), [h], y)
    np.savez_compressed(os.path.join(HERE, 'full224.npz'), y_sub=y[:, :, ::4, ::4].astype(np.float32),
                        r_sub=r[:, :, ::4, ::4].astype(np.float32),
                        h_sub=h[:, ::8].astype(np.float32), y_sha=sha(y), r_sha=sha(r), h_sha=sha(h),
                        **{'cfg_' + k: v for k, v in c.items()})


if __name__ == '__main__':
    mini()
    full()
    for f in ('mini_e2e.npz', 'full224.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')
