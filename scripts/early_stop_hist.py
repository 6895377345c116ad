#!/usr/bin/env python3
"""The loop's stop test as the reference runs it (iterative_inference.py:265-277: an image stops the
moment mean_px ||de||_2 < 1e-3): iteration histogram and images/s on the DAMPED set, where the loop is
contractive and images do converge.  VERDICT round 4, "Next round" item 4 (measure first).

    python scripts/early_stop_hist.py [--batch 64] [--num_iter 50] [--steps 0.1,0.5] [--mma f32,bf16c8]
"""
import argparse
import collections
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_inference_segm_amd import synthetic as S  # noqa: E402


def engine(mma, dtype=torch.float32):
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp, dp, temp = S.make_damped_set()
    m = None if mma == 'f32' else mma
    return IterativeInference(
        FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'], temperature=temp, dtype=dtype, mma=m),
        StandardDAE(dp, 11, dtype=dtype, mma=m), 11, [11], dtype=dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--num_iter', type=int, default=50)
    ap.add_argument('--steps', default='0.1,0.5')
    ap.add_argument('--mma', default='f32,bf16c8')
    ap.add_argument('--eps', type=float, default=1e-3)
    ap.add_argument('--reps', type=int, default=3)
    args = ap.parse_args()
    B = args.batch
    Xs = [torch.from_numpy(S.make_images(B, 224, 224, seed=9000 + i)).cuda() for i in range(2)]
    for mma in args.mma.split(','):
        ii = engine(mma)
        ii.prepare(B, 224, 224)
        for step in [float(v) for v in args.steps.split(',')]:
            res = {}
            for early in (True, False):
                out = ii.pred_fcn_fn(Xs[0])
                ii.refine(out[:-1], out[-1], step, args.num_iter, eps=args.eps, early_stop=early)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for r in range(args.reps):
                    out = ii.pred_fcn_fn(Xs[(r + 1) % 2])
                    Yii, iters, norm = ii.refine(out[:-1], out[-1], step, args.num_iter, eps=args.eps,
                                                 early_stop=early)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / args.reps
                it = iters.cpu().tolist()
                res['early' if early else 'fixed'] = {
                    'images_per_s': round(B / dt, 1), 'ms_per_batch': round(dt * 1e3, 2),
                    'iters_hist': dict(sorted(collections.Counter(it).items())),
                    'sum_iters': sum(it), 'B_max_iters': B * max(it),
                    'norm_min_max': [float(norm.min()), float(norm.max())]}
            print(json.dumps({'mma': mma, 'step': step, 'num_iter': args.num_iter, 'eps': args.eps,
                              'batch': B, **res}), flush=True)
        del ii
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
