"""End-to-end parity report on the damped synthetic set (iterative_inference_segm_amd/synthetic.py,
DAMPED): configs[1] (FCN-8 + 64-filter DAE, 224x224, 10 steps of 0.1, early stop off), N images in
batches of 8, every fast mode against the float64 HIP path (itself pinned to the CPU oracle by
tests/test_gpu_damped.py / tests/test_gpu_f64.py).  Prints a markdown table.
Usage: python scripts/parity_report.py [n_images=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from iterative_inference_segm_amd import synthetic as S
from iterative_inference_segm_amd.api import IterativeInference
from iterative_inference_segm_amd.dae import StandardDAE
from iterative_inference_segm_amd.fcn8 import FCN8

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fp, dp, temp = S.make_damped_set()


def engine(dtype, mma=None):
    return IterativeInference(
        FCN8(fp, 11, layer=['pool4', 'probs_dimshuffle'], temperature=temp, dtype=dtype,
             mma=None if mma == 'bf16x3' else mma),
        StandardDAE(dp, 11, dtype=dtype, mma=mma), 11, [11], dtype=dtype)


modes = [('fp32 MFMA', torch.float32, None), ('bf16x3 (DAE loop on hi/lo pairs)', torch.float32, 'bf16x3'),
         ('bf16 operands, fp32 activations', torch.float32, 'bf16'),
         ('bf16 operands, bf16 C8 activations', torch.float32, 'bf16c8')]
ii64 = engine(torch.float64)
engines = [(n, engine(dt, m)) for n, dt, m in modes]
st = {n: dict(within=0, px=0, mx=0.0, sm=0.0, agree=0, cm=np.zeros((11, 12))) for n, _, _ in modes}
for b in range(N // 8):
    X = S.make_images(8, 224, 224, seed=5000 + b)
    o = ii64.pred_fcn_fn(X)
    ref = ii64.refine(o[:-1], o[-1], 0.1, 10, early_stop=False)[0]
    T = S.labels_from_map(ref.cpu().numpy(), seed=6000 + b)
    for n, ii in engines:
        o = ii.pred_fcn_fn(X)
        y = ii.refine(o[:-1], o[-1], 0.1, 10, early_stop=False)[0]
        e = (y.double() - ref).abs()
        s = st[n]
        s['within'] += int((e.amax(1) <= 1e-4).sum()); s['px'] += e.shape[0] * e.shape[2] * e.shape[3]
        s['mx'] = max(s['mx'], float(e.max())); s['sm'] += float(e.sum())
        s['agree'] += int((y.argmax(1) == ref.argmax(1)).sum())
        s['cm'] += ii.val_device(y, T).cm.cpu().numpy().reshape(11, 12)
print('| mode | pixels within 1e-4 of float64 | max error | mean error | refined argmax agreement | mIoU '
      '(labels = argmax of the float64 result) |\n|---|---:|---:|---:|---:|---:|')
for n, _, _ in modes:
    s = st[n]
    c = s['cm'][:, :11]; tp = np.diag(c)
    with np.errstate(invalid='ignore', divide='ignore'):
        miou = float(np.nanmean(tp / (c.sum(1) + c.sum(0) - tp)))
    print('| %s | %.5f | %.2e | %.2e | %.6f | %.5f |' % (n, s['within'] / s['px'], s['mx'],
                                                         s['sm'] / (s['px'] * 11), s['agree'] / s['px'], miou))
print('\n%d images (batches of 8, seeds 5000...), 10 steps of 0.1, early stop off, product path '
      '(`refine()`, own DePool2D masks, HIP-graph replay).' % N)
