"""How fast the free-running refinement loop amplifies a perturbation of y0 (float64 HIP path =
the reference's numerics), and how far the fp32 path drifts from it, per number of steps.
Usage: python scripts/sensitivity.py [c2|c2d|c3|c5]   (c2d: configs[1] on the DAMPED synthetic set,
synthetic.DAMPED -- there the perturbation decays) """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from iterative_inference_segm_amd import synthetic as S
from iterative_inference_segm_amd.api import IterativeInference
from iterative_inference_segm_amd.dae import StandardDAE

which = sys.argv[1] if len(sys.argv) > 1 else 'c2'
F32, F64 = torch.float32, torch.float64
if which == 'c3':
    from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan
    params = S.make_densenet_params(layer_plan()); dp = S.make_dae_params(h_channels=(464,))
    mk = lambda dt: IterativeInference(FCDenseNet(params, 11, layer=['pool4'], dtype=dt),
                                       StandardDAE(dp, 11, padding=0, dtype=dt), 11, [11], dtype=dt)
    B, steps = 4, [1, 2, 5, 10]
else:
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp = S.make_fcn8_params()
    if which == 'c5':
        ch = ['pool3', 'pool4']; dp = S.make_dae_params(h_channels=(256, 512), concat_h=ch)
        kw = dict(concat_h=ch, pad_multi_concat=True); B, steps = 2, [1, 2, 5, 10, 20, 50]
    else:
        ch = ['pool4']; dp = S.make_dae_params(); kw = {}; B, steps = 2, [1, 2, 5, 10]
    temp = 1.0
    if which == 'c2d':
        fp, dp, temp = S.make_damped_set()
    mk = lambda dt: IterativeInference(FCN8(fp, 11, layer=ch + ['probs_dimshuffle'], dtype=dt,
                                            temperature=temp),
                                       StandardDAE(dp, 11, dtype=dt, **kw), 11, [11], dtype=dt)
X = S.make_images(B, 224, 224, seed=1234)
ii64, ii32 = mk(F64), mk(F32)
o64, o32 = ii64.pred_fcn_fn(X), ii32.pred_fcn_fn(X)
H64, Y64 = o64[:-1], o64[-1]
g = torch.Generator(device='cuda').manual_seed(0)
u = torch.rand(Y64.shape, generator=g, device='cuda', dtype=F64) * 2 - 1
def stat(a, b):
    e = (a - b).abs()
    return 'max %.2e mean %.2e within1e-4 %.4f argmax %.5f' % (
        float(e.max()), float(e.mean()), float((e.amax(1) <= 1e-4).double().mean()),
        float((a.argmax(1) == b.argmax(1)).double().mean()))
for n in steps:
    base = ii64.refine(H64, Y64, 0.1, n, early_stop=False)[0]
    print('steps %d' % n, flush=True)
    for eps in (1e-7, 1e-6, 1e-5):
        pert = ii64.refine(H64, (Y64 * (1 + eps * u)).clamp(0, 1), 0.1, n, early_stop=False)[0]
        print('   f64 vs f64(y0*(1+%g u)): %s' % (eps, stat(base, pert)), flush=True)
    got = ii32.refine(o32[:-1], o32[-1], 0.1, n, early_stop=False)[0].to(F64)
    print('   f64 vs fp32 path        : %s' % stat(base, got), flush=True)
