"""The drop-in entry point: same function / flag names as reference iterative_inference.py and
the same error behaviour; on the GPU a whole synthetic evaluation runs through it."""
import inspect
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_signature_and_flags_match_the_reference(monkeypatch, capsys):
    import iterative_inference as ii
    params = list(inspect.signature(ii.inference).parameters)
    # reference iterative_inference.py:56-59, in order
    assert params[:13] == ['dataset', 'segm_net', 'learn_step', 'num_iter', 'dae_dict_updates',
                           'training_dict', 'data_augmentation', 'which_set', 'ae_h', 'full_im_ft',
                           'savepath', 'loadpath', 'test_from_0_255']
    sig = inspect.signature(ii.inference)
    assert sig.parameters['learn_step'].default == 0.005 and sig.parameters['num_iter'].default == 500
    assert ii._EPSILON == 1e-3
    monkeypatch.setattr(sys, 'argv', ['iterative_inference.py', '-h'])
    with pytest.raises(SystemExit):
        ii.main()
    helptext = capsys.readouterr().out
    for flag in ['-dataset', '-segmentation_net', '-step', '--num_iter', '-ne', '-which_set',
                 '-dae_dict', '-training_dict', '-full_im_ft', '-ae_h', '-data_augmentation',
                 '-test_from_0_255']:
        assert flag in helptext


def test_error_behaviour_without_gpu(tmp_path):
    import iterative_inference as ii
    with pytest.raises(ValueError, match='saving directory'):
        ii.inference('camvid', 'fcn8', savepath=None)                       # :88-89
    kw = dict(savepath=str(tmp_path / 's'), loadpath=str(tmp_path / 'l'), synthetic=True,
              verbose=False, n_images=2)
    with pytest.raises(ValueError):
        ii.inference('camvid', 'nonsense_net', **kw)                        # :146-147
    with pytest.raises(NotImplementedError):
        ii.inference('camvid', 'fcn_fcresnet', **kw)                        # :144-145
    with pytest.raises(ValueError, match='Unknown dataset'):
        ii.inference('imagenet', 'fcn8', **kw)


def test_extra_dae_dict_key_does_not_reach_the_name_builder(tmp_path):
    """`dae_dict['emulate_noise']` (an extra key of this build) must not be forwarded to
    build_experiment_name, which mirrors the reference's keyword list (helpers.py:118-169): the
    call has to get past the name to the dataset check, in both drivers."""
    import iterative_inference as ii
    import iterative_inference_valid as iv
    kw = dict(savepath=str(tmp_path / 's'), loadpath=str(tmp_path / 'l'), synthetic=True,
              verbose=False, n_images=2, dae_dict_updates={'emulate_noise': True, 'kind': 'standard'})
    for mod in (ii, iv):
        with pytest.raises(ValueError, match='Unknown dataset'):
            mod.inference('imagenet', 'fcn8', **kw)


@pytest.mark.gpu
@pytest.mark.parametrize('default_mma', ['f32', 'bf16x3'])
def test_synthetic_evaluation_end_to_end(built_lib, tmp_path, monkeypatch, default_mma):
    """2 batches of 2 images (64x48), reduced DAE width, 3 steps: summary numbers agree with the
    float64 oracle run of the same evaluation; files of the reference are written.  Also with
    IISEG_MMA=bf16x3 (the DAE loop on bf16 hi / lo pairs): same tolerances."""
    import torch
    import iterative_inference as ii
    from iterative_inference_segm_amd import ops as _ops
    monkeypatch.setattr(_ops, 'DEFAULT_MMA', default_mma)
    from oracle import dae as odae, fcn8 as ofcn8, metrics as ometrics, refine as orefine
    from iterative_inference_segm_amd import synthetic as S
    from iterative_inference_segm_amd.data_loader import load_data
    dd = {'kind': 'standard', 'unpool_type': 'trackind', 'n_filters': 4, 'additional_pool': 2,
          'concat_h': ['pool4'], 'skip': True, 'from_gt': False}
    # small FCN weights on disk in the reference's arr_%d layout, DAE weights synthetic
    from iterative_inference_segm_amd import weights, fcn8 as pfcn8
    fp = S.make_fcn8_params(width_div=16, fc_channels=32, seed=1234)
    wdir = tmp_path / 'w' / 'camvid'
    wdir.mkdir(parents=True)
    weights.save_param_list(str(wdir / 'fcn8_model.npz'), fp, pfcn8.PARAM_ORDER)
    # h_channels table of the entry point assumes real VGG widths; patch the synthetic DAE maker
    dp = S.make_dae_params(h_channels=(fp['conv4_3'][0].shape[0],), n_filters=4, seed=4321)
    exp = ii.build_experiment_name('fcn8', data_aug=False, ae_h=False, **dict(
        {'kind': 'fcn8', 'dropout': 0.0, 'skip': True, 'unpool_type': 'standard', 'n_filters': 64,
         'conv_before_pool': 1, 'additional_pool': 0, 'concat_h': ['input'], 'noise': 0.0,
         'from_gt': True, 'temperature': 1.0, 'layer': 'probs_dimshuffle', 'exp_name': '',
         'bn': 0}, **dd))
    ldir = tmp_path / 'l' / 'camvid' / exp
    ldir.mkdir(parents=True)
    from iterative_inference_segm_amd import dae as pdae
    weights.save_param_list(str(ldir / 'dae_model_best.npz'), dp, pdae.param_order())
    out = ii.inference('camvid', 'fcn8', 0.1, 3, dae_dict_updates=dd, savepath=str(tmp_path / 's'),
                       loadpath=str(tmp_path / 'l'), weights_path=str(tmp_path / 'w'),
                       synthetic=True, n_images=4, image_size=(64, 48), batch_size=2,
                       verbose=False)
    sdir = tmp_path / 's' / 'camvid' / exp / 'img_plots' / 'test'
    assert (sdir / 'config.txt').exists() and (sdir / 'batch0.npz').exists()
    with np.load(str(sdir / 'batch1.npz')) as f:
        assert sorted(f.files) == ['L', 'X', 'Y_fcn', 'Y_ii'] and f['Y_ii'].shape == (2, 11, 64, 48)
    assert (ldir / 'img_plots' / 'test' / 'batch0.npz').exists()           # copy_tree (:324-326)

    # oracle evaluation of the same two batches
    it = load_data('camvid', {}, one_hot=True, batch_size=[2, 5, 2], which_set='test',
                   synthetic=True, n_images=4, image_size=(64, 48))
    to64 = lambda p: {k: tuple(np.asarray(a, np.float64) for a in v) for k, v in p.items()}
    fp64, dp64 = to64(fp), to64(dp)
    rec = acc = 0.0
    jacc = np.zeros((2, 11))
    for i in range(2):
        X, L = it.batch(i)
        h, y = ofcn8.fcn8_forward(fp64, X.astype(np.float64), layer=['pool4', 'probs_dimshuffle'])
        yii, _ = orefine.refine_batch(lambda hh, yy: odae.dae_forward(dp64, hh, yy, n_filters=4),
                                      [h], y, 0.1, 3)
        a, j, m = ometrics.val_fn(yii, L.astype(np.float64), 11, [11])
        rec += m; acc += a; jacc += j
    loss_r, acc_r, miou_r = ometrics.summarize(rec, acc, jacc, 2)
    assert out['ii']['batches'] == 2
    assert abs(out['ii']['jaccard'] - miou_r) <= 0.05            # north_star: mIoU within +-0.05
    assert abs(out['ii']['acc'] - acc_r) <= 1e-3 and abs(out['ii']['loss'] - loss_r) <= 1e-4


@pytest.mark.gpu
def test_valid_driver_per_iteration_jaccard(built_lib, tmp_path):
    """iterative_inference_valid.inference: per-iteration mean Jaccard over images still
    iterating == the oracle's restatement of valid_mat (reference :231,265-288,298), including an
    image that stops early (big eps via a tiny step is not available, so we stop by eps below)."""
    import torch
    import iterative_inference_valid as iv
    from oracle import dae as odae, fcn8 as ofcn8, metrics as ometrics
    from iterative_inference_segm_amd import synthetic as S
    from iterative_inference_segm_amd.data_loader import load_data
    dd = {'kind': 'standard', 'unpool_type': 'trackind', 'n_filters': 4, 'additional_pool': 2,
          'concat_h': ['pool4'], 'skip': True, 'from_gt': False}
    res = iv.inference('camvid', 'fcn8', 0.3, 3, dae_dict_updates=dd, which_set='val',
                       savepath=str(tmp_path / 's'), loadpath=str(tmp_path / 'l'),
                       weights_path=str(tmp_path / 'w'), synthetic=True, n_images=3,
                       image_size=(32, 32), batch_size=2, verbose=False)
    assert res.shape == (3,)
    # oracle restatement with the same synthetic nets (full-width FCN-8 at 32x32 is cheap)
    fp = S.make_fcn8_params(3, 11, seed=1234)
    dp = S.make_dae_params(11, (512,), ['pool4'], 4, 1, 2, 'trackind', seed=4321)
    to64 = lambda p: {k: tuple(np.asarray(a, np.float64) for a in v) for k, v in p.items()}
    fp64, dp64 = to64(fp), to64(dp)
    it = load_data('camvid', {}, one_hot=True, batch_size=[2, 2, 2], which_set='val',
                   synthetic=True, n_images=3, image_size=(32, 32))
    valid_mat = np.zeros((2, 11, 3))
    for i in range(it.nbatches):
        X, L = it.batch(i)
        h, y = ofcn8.fcn8_forward(fp64, X.astype(np.float64), layer=['pool4', 'probs_dimshuffle'])
        for im in range(X.shape[0]):
            y_im, h_im, t_im = y[im:im + 1], [h[im:im + 1]], L[im:im + 1].astype(np.float64)
            for k in range(3):                                    # :265-288
                grad = y_im - odae.dae_forward(dp64, h_im, y_im, n_filters=4)
                y_im = np.clip(y_im - 0.3 * grad, 0, 1)
                if np.linalg.norm(grad, axis=1).mean() < 1e-3:
                    break
                valid_mat[:, :, k] += ometrics.jaccard(y_im, t_im, 11)
    with np.errstate(invalid='ignore', divide='ignore'):
        ref = np.nanmean(valid_mat[0] / valid_mat[1], axis=0)
    assert np.allclose(res, ref, atol=2e-3, equal_nan=True)
    saved = tmp_path / 's' / 'camvid'
    assert any(f.name.startswith('iterations0.3') for f in saved.rglob('*.npz'))


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['fcn8', 'contextmod'])
def test_default_mma_bf16c8_with_the_non_standard_dae_kinds(built_lib, tmp_path, monkeypatch, kind):
    """`--mma bf16c8` / IISEG_MMA=bf16c8 with the entry point's DEFAULT dae kind ('fcn8', reference
    iterative_inference.py:64-77) and with 'contextmod': these DAEs keep fp32 NCHW activations, so
    their layers -- built under a C8 default -- must answer pool / mask fusion questions for the
    form that actually runs.  The evaluation runs and agrees with the f32 default
    run within the 16-bit mode's statistical tolerance (mIoU +-0.05)."""
    import iterative_inference as ii
    from iterative_inference_segm_amd import ops as _ops
    dd = {'kind': kind, 'concat_h': ['input']}
    res = {}
    for mma in ('f32', 'bf16c8'):
        monkeypatch.setattr(_ops, 'DEFAULT_MMA', mma)
        out = ii.inference('camvid', 'fcn8', 0.1, 2, dae_dict_updates=dd,
                           savepath=str(tmp_path / ('s' + mma)), loadpath=str(tmp_path / ('l' + mma)),
                           weights_path=str(tmp_path / 'w'), synthetic=True, n_images=2,
                           image_size=(64, 48), batch_size=2, verbose=False, save_npz=False)
        res[mma] = out['ii']
        assert out['ii']['batches'] == 1 and np.isfinite(out['ii']['loss'])
    assert abs(res['bf16c8']['jaccard'] - res['f32']['jaccard']) <= 0.05


@pytest.mark.gpu
def test_batches_in_flight_do_not_change_what_the_driver_writes(built_lib, tmp_path):
    """inference(in_flight=N): N batches being worked on at a time (api.EnginePool, one engine / HIP stream
    each) -- the summary and every batch%d.npz are those of the one-after-the-other run, bit for bit,
    including the last, smaller batch and the early stop's host reads."""
    import iterative_inference as ii
    dd = {'kind': 'standard', 'unpool_type': 'trackind', 'n_filters': 4, 'additional_pool': 2,
          'concat_h': ['pool4'], 'skip': True, 'from_gt': False}
    outs = {}
    for n in (1, 2, 3):
        outs[n] = ii.inference('camvid', 'fcn8', 0.1, 5, dae_dict_updates=dd, savepath=str(tmp_path / ('s%d' % n)),
                               loadpath=str(tmp_path / ('l%d' % n)), weights_path=str(tmp_path / 'w'),
                               synthetic=True, n_images=9, image_size=(64, 48), batch_size=2, verbose=False,
                               in_flight=n)
        assert outs[n]['ii']['batches'] == 5
    files = {n: sorted((tmp_path / ('s%d' % n)).rglob('batch*.npz')) for n in outs}
    assert len(files[1]) == 5
    for n in (2, 3):
        assert json.dumps(outs[n], sort_keys=True) == json.dumps(outs[1], sort_keys=True)   # (NaN-safe)
        assert [f.name for f in files[n]] == [f.name for f in files[1]]
        for a, b in zip(files[1], files[n]):
            with np.load(str(a)) as fa, np.load(str(b)) as fb:
                for k in ('X', 'L', 'Y_fcn', 'Y_ii'):
                    assert np.array_equal(fa[k], fb[k]), (n, a.name, k)
