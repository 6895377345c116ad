#!/usr/bin/env python3
"""Step / iteration cross-validation of iterative inference on MI355X -- drop-in for the
reference's iterative_inference_valid.py (:56-310 `inference`, :313-390 `main`).

For every refinement iteration the Jaccard numerator / denominator of the images that are still
iterating is accumulated into `valid_mat[2, C, num_iter]` (:231,280-288); `inference` returns
`nanmean(valid_mat[0] / valid_mat[1], axis=0)` (:298) and `main` sweeps the reference's step grid
`[.01, .02, .05, .08, .1, .5, 1.]` (:373) to pick the best (step, num_iter).  The whole loop,
including the per-iteration confusion counts, runs on the device (`api.refine(...,
per_iter_target=...)`); batches shard over ranks with one all-reduce of `valid_mat`.
"""
import argparse
import os

import numpy as np
import torch

import iterative_inference as base
from iterative_inference_segm_amd import dist as iidist
from iterative_inference_segm_amd.api import Metrics
from iterative_inference_segm_amd.helpers import build_experiment_name

STEPS = [.01, .02, .05, .08, .1, .5, 1.]          # iterative_inference_valid.py:373


def inference(dataset, segm_net, learn_step=0.005, num_iter=500, dae_dict_updates={},
              training_dict={}, data_augmentation=False, which_set='test', ae_h=False,
              full_im_ft=False, savepath=None, loadpath=None, test_from_0_255=False,
              weights_path=None, synthetic=False, n_images=20, image_size=(224, 224),
              batch_size=10, verbose=True):
    """Reference signature (iterative_inference_valid.py:56-59) + keyword-only extras.
    Returns the per-iteration mean Jaccard, shape (num_iter,)."""
    dae_dict = {'kind': 'fcn8', 'dropout': 0.0, 'skip': True, 'unpool_type': 'standard',
                'n_filters': 64, 'conv_before_pool': 1, 'additional_pool': 0,
                'concat_h': ['input'], 'noise': 0.0, 'from_gt': True, 'temperature': 1.0,
                'layer': 'probs_dimshuffle', 'exp_name': '', 'bn': 0}
    dae_dict.update(dae_dict_updates)
    name_kw = dict(dae_dict)
    name_kw.update(training_dict)
    name_kw.pop('emulate_noise', None)   # extra key of this build, not part of the reference's name
    exp_name = build_experiment_name(segm_net, data_aug=data_augmentation, ae_h=ae_h, **name_kw)
    exp_name += '_ftsmall' if full_im_ft else ''                         # :86
    if savepath is None:
        raise ValueError('A saving directory must be specified')

    rank, world, device = iidist.init_from_env()
    say = print if (verbose and rank == 0) else (lambda *a, **k: None)
    loadpath = loadpath if loadpath is not None else base.LOADPATH
    weights_path = weights_path if weights_path is not None else base.WEIGHTS_PATH
    savepath = os.path.join(savepath, dataset, exp_name, 'img_plots', str(learn_step), which_set)
    loadpath = os.path.join(loadpath, dataset, exp_name)
    if rank == 0:
        os.makedirs(savepath, exist_ok=True)
    iidist.barrier()

    ii, data_iter = base.build_networks(dataset, segm_net, dae_dict, loadpath, weights_path,
                                        which_set, device, synthetic=synthetic, n_images=n_images,
                                        image_size=image_size, batch_size=batch_size,
                                        val_batch_size=batch_size,        # [10, 10, 10] at :117
                                        test_from_0_255=test_from_0_255, say=say)
    C = data_iter.non_void_nclasses
    counts = torch.zeros((int(num_iter), C * (C + 1)), dtype=torch.int64, device=device)
    for i in iidist.shard_batches(data_iter.nbatches, rank, world):
        X, L = data_iter.batch(i) if hasattr(data_iter, 'batch') else data_iter.next()
        pred = ii.pred_fcn_fn(X)
        Y, H = pred[-1], pred[:-1]
        _, _, _, per_iter = ii.refine(H, Y, learn_step, num_iter, eps=base._EPSILON,
                                      per_iter_target=L)
        counts += per_iter
    if world > 1:
        torch.distributed.all_reduce(counts)
    counts = counts.cpu().numpy()
    valid_mat = np.zeros((2, C, int(num_iter)))                          # :231
    for it in range(int(num_iter)):
        _, jacc, _ = Metrics.reduce_host(counts[it], np.array([0.0, 1.0]), C)
        valid_mat[:, :, it] = jacc
    with np.errstate(invalid='ignore', divide='ignore'):
        res = np.nanmean(valid_mat[0, :, :] / valid_mat[1, :, :], axis=0)  # :298
    if rank == 0:
        say(res.max() if np.isfinite(res).any() else float('nan'))
        say(int(np.nanargmax(res)) if np.isfinite(res).any() else -1)
        say(learn_step)
        np.savez(os.path.join(savepath, 'iterations' + str(learn_step) + '.npz'), valid_mat)  # :304
    iidist.barrier()
    return res


def main():
    parser = argparse.ArgumentParser(description='Iterative inference: step / iteration search.')
    parser.add_argument('-dataset', type=str, default='camvid')
    parser.add_argument('-segmentation_net', type=str, default='fcn8')
    parser.add_argument('-step', type=float, default=0.05)
    parser.add_argument('--num_iter', '-ne', type=int, default=50)
    parser.add_argument('-which_set', type=str, default='val')
    parser.add_argument('-dae_dict', type=base._json_dict,
                        default={'kind': 'standard', 'dropout': 0, 'skip': True,
                                 'unpool_type': 'trackind', 'noise': 0.5, 'concat_h': ['pool4'],
                                 'from_gt': False, 'n_filters': 64, 'conv_before_pool': 1,
                                 'additional_pool': 2, 'path_weights': '',
                                 'layer': 'probs_dimshuffle', 'exp_name': 'flip_final_', 'bn': 0})
    parser.add_argument('-training_dict', type=base._json_dict,
                        default={'training_loss': ['crossentropy', 'squared_error'],
                                 'learning_rate': 0.001, 'lr_anneal': 0.99,
                                 'weight_decay': 0.0001, 'optimizer': 'rmsprop'})
    parser.add_argument('-full_im_ft', type=bool, default=False)
    parser.add_argument('-ae_h', type=bool, default=False)
    parser.add_argument('-data_augmentation', type=bool, default=True)
    parser.add_argument('-test_from_0_255', type=bool, default=False)
    parser.add_argument('--savepath', type=str, default=base.SAVEPATH)
    parser.add_argument('--loadpath', type=str, default=base.LOADPATH)
    parser.add_argument('--weights_path', type=str, default=base.WEIGHTS_PATH)
    parser.add_argument('--synthetic', action='store_true')
    parser.add_argument('--n_images', type=int, default=20)
    parser.add_argument('--image_size', type=int, nargs=2, default=[224, 224])
    parser.add_argument('--batch_size', type=int, default=10)
    args = parser.parse_args()

    all_results = np.zeros((len(STEPS), int(args.num_iter)))             # :374
    for i, s in enumerate(STEPS):                                        # :376-382
        all_results[i, :] = inference(
            args.dataset, args.segmentation_net, s, int(args.num_iter), which_set=args.which_set,
            savepath=args.savepath, loadpath=args.loadpath, test_from_0_255=args.test_from_0_255,
            ae_h=args.ae_h, dae_dict_updates=args.dae_dict,
            data_augmentation=args.data_augmentation, training_dict=args.training_dict,
            full_im_ft=args.full_im_ft, weights_path=args.weights_path, synthetic=args.synthetic,
            n_images=args.n_images, image_size=tuple(args.image_size), batch_size=args.batch_size)
    all_results = np.nan_to_num(all_results, nan=-1.0)
    max_per_step = all_results.max(1)                                    # :383-386
    argmax_per_step = all_results.argmax(1)
    best_step = max_per_step.argmax()
    print('Best step: ' + str(STEPS[best_step]))                         # :388-390
    print('Result: ' + str(max_per_step.max()))
    print('Num iters: ' + str(argmax_per_step[best_step] + 1))


if __name__ == '__main__':
    main()
