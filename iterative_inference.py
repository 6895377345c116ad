#!/usr/bin/env python3
"""Iterative inference on MI355X -- drop-in for the reference's iterative_inference.py.

Same entry points (`inference(...)`, `main()`), same knobs (`-segmentation_net`, `-dae_dict`,
`-step`, `--num_iter`, ...; dict keys of iterative_inference.py:64-77,355-362), same outputs
(stdout summary lines of helpers.py:172-177, `config.txt`, `batch%d.npz` with X, L, Y_ii,
Y_fcn).  Differences, all forced by the environment: lab paths keyed on getuser()
(iterative_inference.py:32-51) become `--savepath/--loadpath/--weights_path`; dict flags are
JSON; `--synthetic` supplies seeded data / weights when no checkpoint or dataset exists; with
WORLD_SIZE > 1 (torchrun) batches shard over the GPUs of the node and the metric accumulators
are all-reduced once (RCCL); on each GPU `--in_flight` batches (default 2) are being worked on at a
time, each by its own engine on its own HIP stream (api.EnginePool), and the host reads a batch's
results -- printing, batch%d.npz -- when the batch after it has been queued.  All arithmetic runs in
the HIP kernels of libiiseg_hip.so.
"""
import argparse
import collections
import json
import os

import numpy as np
import torch

from iterative_inference_segm_amd import dist as iidist
from iterative_inference_segm_amd import synthetic as S
from iterative_inference_segm_amd.api import EPSILON, EnginePool, IterativeInference
from iterative_inference_segm_amd.dae import buildDAE, param_order
from iterative_inference_segm_amd.data_loader import load_data
from iterative_inference_segm_amd.fcn8 import buildFCN8
from iterative_inference_segm_amd.helpers import build_experiment_name, print_results, results_line

SAVEPATH = os.environ.get('IISEG_SAVEPATH', './iiseg_out/save/')
LOADPATH = os.environ.get('IISEG_LOADPATH', './iiseg_out/load/')
WEIGHTS_PATH = os.environ.get('IISEG_WEIGHTS_PATH', './iiseg_out/load/')

_EPSILON = EPSILON  # 1e-3, iterative_inference.py:53


def _copy_tree(src, dst):
    import shutil
    shutil.copytree(src, dst, dirs_exist_ok=True)


def build_networks(dataset, segm_net, dae_dict, loadpath, weights_path, which_set, device,
                   synthetic=False, n_images=20, image_size=(224, 224), batch_size=10,
                   val_batch_size=5, test_from_0_255=False, say=print):
    """Data iterator + segmentation net + DAE as the reference builds them
    (iterative_inference.py:114-179).  Returns (ii, data_iter)."""
    # Build dataset iterator (:117-125)
    data_iter = load_data(dataset, {}, one_hot=True, batch_size=[batch_size, val_batch_size, batch_size],
                          return_0_255=test_from_0_255, which_set=which_set, synthetic=synthetic,
                          n_images=n_images, image_size=image_size)
    n_batches_test = data_iter.nbatches
    n_classes = data_iter.non_void_nclasses
    void_labels = data_iter.void_labels
    nb_in_channels = data_iter.data_shape[0]

    # Build segmentation network (:131-147)
    say('Building segmentation network')
    if segm_net == 'fcn8':
        fcn_weights = os.path.join(weights_path, dataset, 'fcn8_model.npz')
        fcn_params = None
        if not os.path.exists(fcn_weights):
            if not synthetic:
                raise IOError('FCN-8 weights not found: %s (use --synthetic)' % fcn_weights)
            fcn_params = S.make_fcn8_params(nb_in_channels, n_classes, seed=1234)
        fcn = buildFCN8(nb_in_channels, path_weights=fcn_weights, n_classes=n_classes,
                        void_labels=void_labels, trainable=False, load_weights=True,
                        layer=dae_dict['concat_h'] + [dae_dict['layer']], params=fcn_params,
                        device=device)
        padding = 100
        h_channels = {'input': nb_in_channels, 'pool1': 64, 'pool2': 128, 'pool3': 256,
                      'pool4': 512, 'pool5': 512}
    elif segm_net == 'densenet':                                 # :140-143
        from iterative_inference_segm_amd.densenet import build_fcdensenet, layer_plan
        dn_weights = os.path.join(weights_path, dataset, 'FC-DenseNet103_weights.npz')
        dn_params = None
        if not os.path.exists(dn_weights):
            if not synthetic:
                raise IOError('FC-DenseNet weights not found: %s (use --synthetic)' % dn_weights)
            dn_params = S.make_densenet_params(layer_plan(nb_in_channels=nb_in_channels,
                                                          n_classes=n_classes), seed=2024)
        fcn = build_fcdensenet(layer=dae_dict['concat_h'], nb_in_channels=nb_in_channels,
                               n_classes=n_classes, weight_path=dn_weights, params=dn_params,
                               device=device)
        padding = 0
        # pool k stacks of DenseNet103: 48 + 16*(4,9,16,26,38) channels
        h_channels = {'input': nb_in_channels, 'pool1': 112, 'pool2': 192, 'pool3': 304,
                      'pool4': 464, 'pool5': 656}
    elif segm_net == 'fcn_fcresnet':
        raise NotImplementedError                                # :144-145
    else:
        raise ValueError                                         # :146-147

    # Build DAE with pre-trained weights (:149-179)
    say('Building DAE network')
    if dae_dict['kind'] == 'standard':
        dae_weights = os.path.join(loadpath, 'dae_model_best.npz')
        dae_params = None
        if not os.path.exists(dae_weights):
            if not synthetic:
                raise IOError('DAE weights not found: %s (use --synthetic)' % dae_weights)
            dae_params = S.make_dae_params(
                n_classes, tuple(h_channels[c] for c in dae_dict['concat_h']),
                dae_dict['concat_h'], dae_dict['n_filters'], dae_dict['conv_before_pool'],
                dae_dict['additional_pool'], dae_dict['unpool_type'], seed=4321,
                bn=dae_dict['bn'])
        dae = buildDAE(n_classes=n_classes,
                       nb_features_to_concat=h_channels[dae_dict['concat_h'][0]],
                       padding=padding, trainable=True, void_labels=void_labels,
                       load_weights=True, path_weights=loadpath, model_name='dae_model_best.npz',
                       out_nonlin='softmax', concat_h=dae_dict['concat_h'],
                       noise=dae_dict['noise'], n_filters=dae_dict['n_filters'],
                       conv_before_pool=dae_dict['conv_before_pool'],
                       additional_pool=dae_dict['additional_pool'], dropout=dae_dict['dropout'],
                       skip=dae_dict['skip'], unpool_type=dae_dict['unpool_type'],
                       bn=dae_dict['bn'], params=dae_params, device=device,
                       # extra key of this build: the stochastic masks of noise > 0
                       emulate_noise=bool(dae_dict.get('emulate_noise', False)))
    elif dae_dict['kind'] == 'fcn8':                             # :165-170
        from iterative_inference_segm_amd.fcn8 import buildFCN8_DAE
        dae_weights = os.path.join(loadpath, 'dae_model_best.npz')
        dae_params = None
        if not os.path.exists(dae_weights):
            if not synthetic:
                raise IOError('DAE weights not found: %s (use --synthetic)' % dae_weights)
            dae_params = S.make_fcn8_dae_params(
                n_classes, dae_dict['concat_h'],
                tuple(h_channels[c] for c in dae_dict['concat_h']), seed=555)
        dae = buildFCN8_DAE(n_classes=n_classes, nb_in_channels=n_classes, path_weights=loadpath,
                            model_name='dae_model_best.npz', trainable=True, load_weights=True,
                            concat_h=dae_dict['concat_h'], noise=dae_dict['noise'],
                            params=dae_params, device=device)
    elif dae_dict['kind'] == 'contextmod':                       # :171-177
        from iterative_inference_segm_amd.contextmod import buildDAE_contextmod
        dae_weights = os.path.join(loadpath, 'dae_model_best.npz')
        dae_params = None
        if not os.path.exists(dae_weights):
            if not synthetic:
                raise IOError('DAE weights not found: %s (use --synthetic)' % dae_weights)
            dae_params = S.make_contextmod_params(n_classes, nb_in_channels, seed=777)
        dae = buildDAE_contextmod(n_classes=n_classes, path_weights=loadpath,
                                  model_name='dae_model_best.npz', trainable=True,
                                  load_weights=True, out_nonlin='softmax',
                                  noise=dae_dict['noise'], concat_h=dae_dict['concat_h'],
                                  params=dae_params, device=device)
    else:
        raise ValueError('Unknown dae kind')                     # :178-179

    ii = IterativeInference(fcn, dae, n_classes, void_labels, device=device)
    return ii, data_iter


def inference(dataset, segm_net, learn_step=0.005, num_iter=500, dae_dict_updates={},
              training_dict={}, data_augmentation=False, which_set='test', ae_h=False,
              full_im_ft=False, savepath=None, loadpath=None, test_from_0_255=False,
              weights_path=None, synthetic=False, n_images=20, image_size=(224, 224),
              batch_size=10, early_stop=True, save_npz=True, verbose=True, update='residual',
              dry_run=False, in_flight=None):
    """Signature of reference iterative_inference.py:56-59 plus keyword-only extras.
    Returns a dict of the three summary lines (the reference returns None and only prints).

    dry_run: rehearse the multi-rank protocol without any HIP work (CPU, gloo): argument handling,
    rendezvous from the torchrun environment, rank-0 directory / config.txt creation, the sharding
    of the reference batches over ranks, one all-reduce of the three metric accumulators and the
    summary from rank 0.  The per-batch "metrics" are counts of the synthetic labels themselves.

    in_flight (default IISEG_IN_FLIGHT, else 2 -- 3 for batches of 32 images and fewer, 4 for 16 and fewer): whole batches being worked on at a time on this GPU, each by
    its own engine (own nets, sessions, graphs) on its own HIP stream; per-batch outputs and the running
    totals are the same and come in the same order, one batch later."""
    # Update DAE parameters (:64-79)
    dae_dict = {'kind': 'fcn8', 'dropout': 0.0, 'skip': True, 'unpool_type': 'standard',
                'n_filters': 64, 'conv_before_pool': 1, 'additional_pool': 0,
                'concat_h': ['input'], 'noise': 0.0, 'from_gt': True, 'temperature': 1.0,
                'layer': 'probs_dimshuffle', 'exp_name': '', 'bn': 0}
    dae_dict.update(dae_dict_updates)

    # Prepare load/save directories (:84-104)
    name_kw = dict(dae_dict)
    name_kw.update(training_dict)
    name_kw.pop('emulate_noise', None)   # extra key of this build, not part of the reference's name
    exp_name = build_experiment_name(segm_net, data_aug=data_augmentation, ae_h=ae_h, **name_kw)
    if savepath is None:
        raise ValueError('A saving directory must be specified')

    rank, world, device = iidist.init_from_env('cpu' if dry_run else None)
    say = print if (verbose and rank == 0) else (lambda *a, **k: None)
    loadpath = loadpath if loadpath is not None else LOADPATH
    weights_path = weights_path if weights_path is not None else WEIGHTS_PATH
    savepath = os.path.join(savepath, dataset, exp_name, 'img_plots', which_set)
    loadpath = os.path.join(loadpath, dataset, exp_name)
    if rank == 0:
        if not os.path.exists(savepath):
            os.makedirs(savepath)
        else:
            say('\033[93m The following folder already exists {}. '
                'It will be overwritten in a few seconds...\033[0m'.format(savepath))
        say('Saving directory : ' + savepath)
        with open(os.path.join(savepath, 'config.txt'), 'w') as f:
            for key, value in sorted(locals().items()):
                if key not in ('f', 'say'):
                    f.write('{} = {}\n'.format(key, value))
    iidist.barrier()

    if dry_run:
        return _dry_run(dataset, which_set, synthetic, n_images, image_size, batch_size,
                        test_from_0_255, rank, world, device, say)
    if in_flight is None:
        # batches of 32 images and fewer leave more of the chip idle per launch: one more in flight, two more for 16 and fewer
        in_flight = int(os.environ.get('IISEG_IN_FLIGHT', '2' if batch_size > 32 else '3' if batch_size > 16 else '4'))
    in_flight = max(1, int(in_flight)) if device.type == 'cuda' else 1
    engines = []
    for k in range(in_flight):
        ii, it_k = build_networks(dataset, segm_net, dae_dict, loadpath, weights_path, which_set,
                                  device, synthetic=synthetic, n_images=n_images,
                                  image_size=image_size, batch_size=batch_size,
                                  test_from_0_255=test_from_0_255, say=say if k == 0 else (lambda *a, **kw: None))
        engines.append(ii)
        if k == 0:
            data_iter = it_k
    pool = EnginePool(engines, device=device)
    n_batches_test = data_iter.nbatches
    n_classes = data_iter.non_void_nclasses
    void_labels = data_iter.void_labels


    # Infer (:215-294); batches shard over ranks, every reference batch on exactly one rank
    say('Start infering')
    say('Inference step: ' + str(learn_step) + 'num iter ' + str(num_iter))
    tot = {k: iidist.EvalAccumulator(n_classes) for k in ('fcn', 'dae', 'ii')}
    pending = collections.deque()

    def retire():
        # the host side of the oldest batch in flight: metrics, running totals, batch%d.npz
        i, X_b, L_b, Y_fcn, Y_ii, ms, done = pending.popleft()
        if done is not None:
            done.synchronize()
        for key, m in zip(('fcn', 'dae', 'ii'), ms):
            acc, _, mse = m.result()
            tot[key].add_batch(m.cm.cpu().numpy(), acc, mse)
        if world == 1:
            for label, key in (('>>>>> FCN:', 'fcn'), ('>>>>> FCN+DAE:', 'dae'),
                               ('>>>>> ITERATIVE INFERENCE:', 'ii')):
                loss, acc, miou, _, nb = tot[key].results()
                say(label + '\n    Loss: %s\n    Acc: %s\n    Jaccard: %s' % (loss, acc, miou))
        if save_npz:                                             # :293 (with the path separator)
            np.savez(os.path.join(savepath, 'batch' + str(i) + '.npz'), X=X_b,
                     L=L_b, Y_ii=Y_ii.cpu().numpy(), Y_fcn=Y_fcn.cpu().numpy())

    for i in iidist.shard_batches(n_batches_test, rank, world):
        say('-' * 30 + '\n' + '*' * 5 + 'Batch %d out of %d' % (i + 1, n_batches_test) + '*' * 5
            + '\n' + '-' * 30)
        X_test_batch, L_test_batch = data_iter.batch(i) if hasattr(data_iter, 'batch') \
            else data_iter.next()
        with pool.lane() as ii:     # the next engine; everything below is queued on its stream
            L_dev = torch.from_numpy(np.ascontiguousarray(L_test_batch, dtype=np.float32)).to(device)
            pred = ii.pred_fcn_fn(X_test_batch)                      # :237-239
            Y_test_batch, H_test_batch = pred[-1], pred[:-1]
            m_fcn = ii.val_device(Y_test_batch, L_dev)               # :242
            Y_dae = ii.pred_dae_fn(*(H_test_batch + [Y_test_batch]))  # :250
            m_dae = ii.val_device(Y_dae, L_dev)                      # :251
            Y_ii, iters, _ = ii.refine(H_test_batch, Y_test_batch, learn_step, num_iter,
                                       eps=_EPSILON, early_stop=early_stop, mode=update)   # :257-284
            m_ii = ii.val_device(Y_ii, L_dev)                        # :287
            # (results keep their own buffers: the engine's next batch does not overwrite them)
            done = torch.cuda.Event() if device.type == 'cuda' else None
            if done is not None:
                done.record()
        pending.append((i, X_test_batch, L_test_batch, Y_test_batch, Y_ii, (m_fcn, m_dae, m_ii), done))
        if len(pending) >= len(pool):
            retire()
    while pending:
        retire()

    # one all-reduce of the metric accumulators (RCCL over xGMI; no-op on one GPU)
    for acc in tot.values():
        acc.all_reduce(device)

    # Print summary of how things went (:303-321)
    say('-' * 67 + '\n' + '-' * 30 + 'SUMMARY' + '-' * 30 + '\n' + '-' * 67)
    summary = {}
    for label, key in (('>>>>> FCN:', 'fcn'), ('>>>>> FCN+DAE:', 'dae'),
                       ('>>>>> ITERATIVE INFERENCE:', 'ii')):
        loss, acc, miou, iou, nb = tot[key].results()
        summary[key] = {'loss': loss, 'acc': acc, 'jaccard': miou, 'per_class': iou.tolist(),
                        'batches': nb}
        say(label + '\n    Loss: %s\n    Acc: %s\n    Jaccard: %s' % (loss, acc, miou))
    say('>>>>> Per class jaccard:')
    labs = data_iter.mask_labels
    for c in range(len(labs) - len(void_labels)):
        say('    ' + labs[c] + ' : fcn ->  %f, ii ->  %f'
            % (summary['fcn']['per_class'][c], summary['ii']['per_class'][c]))

    # Move segmentations (:323-326)
    if rank == 0 and savepath != loadpath and save_npz:
        say('Copying images to {}'.format(loadpath))
        _copy_tree(savepath, os.path.join(loadpath, 'img_plots', which_set))
    iidist.barrier()
    return summary


def _dry_run(dataset, which_set, synthetic, n_images, image_size, batch_size, test_from_0_255,
             rank, world, device, say):
    """The data-parallel protocol of `inference` on CPU: same data iterator, same sharding, same
    accumulators and all-reduce; a batch's confusion counts are those of its labels against
    themselves (diagonal), so the reduced summary is checkable: Jaccard 1, every batch counted once."""
    data_iter = load_data(dataset, {}, one_hot=True, batch_size=[batch_size, 5, batch_size],
                          return_0_255=test_from_0_255, which_set=which_set, synthetic=synthetic,
                          n_images=n_images, image_size=image_size)
    n_classes = data_iter.non_void_nclasses
    tot = {k: iidist.EvalAccumulator(n_classes) for k in ('fcn', 'dae', 'ii')}
    mine = iidist.shard_batches(data_iter.nbatches, rank, world)
    for i in mine:
        _, L = data_iter.batch(i) if hasattr(data_iter, 'batch') else data_iter.next()
        counts = np.asarray(L)[:, :n_classes].sum(axis=(0, 2, 3))
        cm = np.zeros((n_classes, n_classes + 1))
        cm[np.arange(n_classes), np.arange(n_classes)] = counts
        for acc in tot.values():
            acc.add_batch(cm, 1.0, 0.0)
    for acc in tot.values():
        acc.all_reduce(device)
    summary = {}
    for label, key in (('>>>>> FCN:', 'fcn'), ('>>>>> FCN+DAE:', 'dae'),
                       ('>>>>> ITERATIVE INFERENCE:', 'ii')):
        loss, acc, miou, iou, nb = tot[key].results()
        summary[key] = {'loss': loss, 'acc': acc, 'jaccard': miou, 'per_class': iou.tolist(),
                        'batches': nb}
        say(label + '\n    Loss: %s\n    Acc: %s\n    Jaccard: %s' % (loss, acc, miou))
    say('DRY RUN: %d ranks, %d batches reduced, rank 0 owned %s' % (world, summary['ii']['batches'], mine))
    iidist.barrier()
    return summary


def _json_dict(s):
    return json.loads(s) if isinstance(s, str) else s


def main():
    parser = argparse.ArgumentParser(description='Iterative inference.')
    parser.add_argument('-dataset', type=str, default='camvid', help='Dataset.')
    parser.add_argument('-segmentation_net', type=str, default='fcn8', help='Segmentation network.')
    parser.add_argument('-step', type=float, default=1.0, help='step')
    parser.add_argument('--num_iter', '-ne', type=int, default=1, help='Max number of iterations')
    parser.add_argument('-which_set', type=str, default='test', help='Inference set')
    parser.add_argument('-dae_dict', type=_json_dict,
                        default={'kind': 'contextmod', 'dropout': 0, 'skip': True,
                                 'unpool_type': 'trackind', 'noise': 0, 'concat_h': ['input'],
                                 'from_gt': False, 'n_filters': 64, 'conv_before_pool': 1,
                                 'additional_pool': 2, 'path_weights': '',
                                 'layer': 'probs_dimshuffle', 'exp_name': 'flip_final_', 'bn': 0},
                        help='DAE kind and parameters (JSON)')
    parser.add_argument('-training_dict', type=_json_dict,
                        default={'training_loss': ['crossentropy'], 'learning_rate': 0.0001,
                                 'lr_anneal': 0.99, 'weight_decay': 0.0001,
                                 'optimizer': 'rmsprop'},
                        help='Training parameters (JSON)')
    parser.add_argument('-full_im_ft', type=bool, default=False)
    parser.add_argument('-ae_h', type=bool, default=False)
    parser.add_argument('-data_augmentation', type=bool, default=True)
    parser.add_argument('-test_from_0_255', type=bool, default=False)
    # replacements for the getuser() path table, and synthetic mode
    parser.add_argument('--savepath', type=str, default=SAVEPATH)
    parser.add_argument('--loadpath', type=str, default=LOADPATH)
    parser.add_argument('--weights_path', type=str, default=WEIGHTS_PATH)
    parser.add_argument('--synthetic', action='store_true',
                        help='seeded synthetic data and weights (no dataset / checkpoints here)')
    parser.add_argument('--n_images', type=int, default=20)
    parser.add_argument('--image_size', type=int, nargs=2, default=[224, 224])
    parser.add_argument('--batch_size', type=int, default=10)
    parser.add_argument('--no_early_stop', action='store_true')
    parser.add_argument('--dry_run', action='store_true',
                        help='CPU rehearsal of the multi-rank protocol (gloo), no HIP work')
    parser.add_argument('--in_flight', type=int, default=None,
                        help='whole batches worked on at a time on each GPU (engines / HIP streams; default '
                             'IISEG_IN_FLIGHT, else 2, 3 for batches <= 32, 4 for <= 16; 1 = one batch after the other)')
    parser.add_argument('--update', choices=['residual', 'gradient'], default='residual',
                        help="'residual': the reference's y += step*(r - y) (default); 'gradient': "
                             "descend the true gradient of ||r(y|h) - y||^2 (extension)")
    parser.add_argument('--mma', choices=['f32', 'bf16', 'bf16c8', 'bf16x3'], default=None,
                        help="matrix-pipe operand mode of the float32 path (same as IISEG_MMA): 'f32' "
                             "(default, the path with the 1e-4 claims), 'bf16' / 'bf16c8' (16-bit "
                             "operands, statistical parity), 'bf16x3' (the DAE loop on bf16 hi / lo "
                             "pairs: fp32-class, DESIGN 3.8)")
    args = parser.parse_args()
    if args.mma is not None and not args.dry_run:
        from iterative_inference_segm_amd import ops as _ops
        _ops.DEFAULT_MMA = args.mma

    inference(args.dataset, args.segmentation_net, float(args.step), int(args.num_iter),
              which_set=args.which_set, savepath=args.savepath, loadpath=args.loadpath,
              full_im_ft=args.full_im_ft, test_from_0_255=args.test_from_0_255, ae_h=args.ae_h,
              dae_dict_updates=args.dae_dict, data_augmentation=args.data_augmentation,
              training_dict=args.training_dict, weights_path=args.weights_path,
              synthetic=args.synthetic, n_images=args.n_images,
              image_size=tuple(args.image_size), batch_size=args.batch_size,
              early_stop=not args.no_early_stop, update=args.update, dry_run=args.dry_run,
              in_flight=args.in_flight)


if __name__ == '__main__':
    main()
