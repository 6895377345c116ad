"""End-to-end parity report of BASELINE configs[2] AS WRITTEN (FC-DenseNet103 + standard DAE, 224x224, batch 32,
10 steps of 0.1, bf16 operands + bf16 C8 activations, fp32 accumulate) on the damped synthetic set of that
config (synthetic.make_damped_densenet_set): every mode against the float64 HIP path (pinned to the CPU oracle by
tests/test_gpu_configs.py::test_config3_densenet103_vs_oracle_batch2) -- labels = argmax of the float64-refined
map, so the float64 path's mIoU is 1 by construction and north_star's "mIoU within +-0.05" reads
|mIoU(mode) - 1| <= 0.05.  Batch-statistics BatchNorm couples the images of a batch: whole batches of 32.
Prints a markdown table.  Usage: python scripts/parity_c3.py [n_batches=2] [default|damped]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from iterative_inference_segm_amd import synthetic as S
from iterative_inference_segm_amd.api import IterativeInference
from iterative_inference_segm_amd.dae import StandardDAE
from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 2
which = sys.argv[2] if len(sys.argv) > 2 else 'damped'
B = 32
plan = layer_plan()
if which == 'damped':
    params, dp = S.make_damped_densenet_set(plan)
else:
    params, dp = S.make_densenet_params(plan), S.make_dae_params(h_channels=(464,))


def engine(dtype, mma=None):
    return IterativeInference(FCDenseNet(params, 11, layer=['pool4'], dtype=dtype, mma=mma),
                              StandardDAE(dp, 11, padding=0, dtype=dtype, mma=mma), 11, [11], dtype=dtype)


modes = [('fp32 MFMA', torch.float32, None), ('bf16 operands, fp32 activations', torch.float32, 'bf16'),
         ('bf16 operands, bf16 C8 activations (configs[2] as written)', torch.float32, 'bf16c8')]
ii64 = engine(torch.float64)
engines = [(n, engine(dt, m)) for n, dt, m in modes]
st = {n: dict(within=0, px=0, mx=0.0, sm=0.0, agree=0, agree0=0, hrms=[], cm=np.zeros((11, 12))) for n, _, _ in modes}
conf = []
for b in range(NB):
    X = S.make_images(B, 224, 224, seed=7000 + b)
    o = ii64.pred_fcn_fn(X)
    ref = ii64.refine(o[:-1], o[-1], 0.1, 10, early_stop=False)[0]
    y0_ref, h_ref = o[-1], o[0]
    conf.append(float(y0_ref.amax(1).mean()))
    T = S.labels_from_map(ref.cpu().numpy(), seed=8000 + b)
    for n, ii in engines:
        o = ii.pred_fcn_fn(X)
        y = ii.refine(o[:-1], o[-1], 0.1, 10, early_stop=False)[0]
        e = (y.double() - ref).abs()
        s = st[n]
        s['within'] += int((e.amax(1) <= 1e-4).sum()); s['px'] += e.shape[0] * e.shape[2] * e.shape[3]
        s['mx'] = max(s['mx'], float(e.max())); s['sm'] += float(e.sum())
        s['agree'] += int((y.argmax(1) == ref.argmax(1)).sum())
        s['agree0'] += int((o[-1].argmax(1) == y0_ref.argmax(1)).sum())
        s['hrms'].append(float(((o[0].double() - h_ref) ** 2).mean().sqrt() / (h_ref ** 2).mean().sqrt()))
        s['cm'] += ii.val_device(y, T).cm.cpu().numpy().reshape(11, 12)
print('| mode | pixels within 1e-4 of float64 | max error | mean error | refined argmax agreement | mIoU '
      '(labels = argmax of the float64 result) | y0 argmax agreement | h rel rms |\n|---|---:|---:|---:|---:|---:|---:|---:|')
for n, _, _ in modes:
    s = st[n]
    c = s['cm'][:, :11]; tp = np.diag(c)
    with np.errstate(invalid='ignore', divide='ignore'):
        miou = float(np.nanmean(tp / (c.sum(1) + c.sum(0) - tp)))
    print('| %s | %.5f | %.2e | %.2e | %.6f | %.5f | %.6f | %.2e |'
          % (n, s['within'] / s['px'], s['mx'], s['sm'] / (s['px'] * 11), s['agree'] / s['px'], miou,
             s['agree0'] / s['px'], float(np.mean(s['hrms']))))
print('\n%s set, %d batches of %d images (seeds 7000...), FC-DenseNet103 (batch-statistics BatchNorm) + standard '
      'DAE (padding 0, h = pool4), 10 steps of 0.1, early stop off, product path (`refine()`, own DePool2D '
      'masks, HIP-graph replay); mean max-probability of the float64 y0: %.3f.' % (which, NB, B, float(np.mean(conf))))
