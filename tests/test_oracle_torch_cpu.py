"""CPU: the float32 torch-CPU baseline restatement (oracle/torch_cpu.py, bench.py's cpu_baseline
leg) against the float64 numpy oracle on a small FCN-8 + standard DAE, both loop schedules."""
import numpy as np

from oracle import dae as odae, fcn8 as ofcn8, refine as orefine, torch_cpu as tcpu
from iterative_inference_segm_amd import synthetic as S


def test_torch_cpu_baseline_matches_the_numpy_oracle():
    fp = S.make_fcn8_params(width_div=16, fc_channels=32, seed=3)
    dp = S.make_dae_params(h_channels=(fp['conv4_3'][0].shape[0],), n_filters=4, seed=4)
    X = S.make_images(2, 40, 32, seed=5)
    to64 = lambda p: {k: tuple(np.asarray(a, np.float64) for a in v) for k, v in p.items()}
    h_ref, y_ref = ofcn8.fcn8_forward(to64(fp), X.astype(np.float64),
                                      layer=['pool4', 'probs_dimshuffle'])
    dp64 = to64(dp)
    yii_ref, _ = orefine.refine_batch(lambda hh, yy: odae.dae_forward(dp64, hh, yy, n_filters=4),
                                      [h_ref], y_ref, 0.1, 3, eps=-1.0)
    Pf, Pd = tcpu.prepare_params(fp), tcpu.prepare_params(dp)
    for per_image in (True, False):
        got = tcpu.run_batch(Pf, Pd, X, 0.1, 3, per_image).numpy()
        assert got.shape == yii_ref.shape
        assert np.abs(got - yii_ref).max() <= 1e-4, per_image
