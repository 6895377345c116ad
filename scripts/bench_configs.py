"""Times the non-headline BASELINE configs on one MI355X (parity-test cases, not bench lines):
  c3: FC-DenseNet103 + standard DAE (padding 0, h = pool4 464 ch @14^2), 224x224, batch 32, 10 steps
  c2_f64: configs[1] network in the float64 strict-parity mode, batch 16
  c4: FCN-8 + standard DAE at 360x480, batch 32 per GPU, 10 steps
  c5: FCN-8 + generalised standard DAE, concat_h=[pool3, pool4] (256 + 512 ch), pad-100, 50 steps,
      224x224, batch 64 (SURVEY A9' variant (ii), build-defined)
  c5ctx: FCN-8 + contextmod DAE with concat_h=[input], 50 steps (A9' variant (i))
  c2_grad: configs[1] in the true-gradient mode (extension): forward + hand-written backward per step
  fcn8dae: FCN-8 + the 'fcn8'-kind DAE (models/fcn8_dae.py, the default kind of inference()), concat_h =
      [input, pool3, pool4], real widths, 224x224, batch 32, 10 steps (~119 + 10 x ~125 GFLOP per image)
An optional third argument selects the matrix-operand mode of the fp32 rows: f32 (default) | bf16 | bf16c8 |
bf16x3 (the standard DAE's loop on bf16 hi / lo pairs, the segmentation net in fp32).
Usage: python scripts/bench_configs.py [c3|c2_f64|c4|c5|c5ctx|c2_grad|fcn8dae] [reps] [mma]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from iterative_inference_segm_amd import synthetic as S
from iterative_inference_segm_amd.api import IterativeInference
from iterative_inference_segm_amd.dae import StandardDAE

which = sys.argv[1] if len(sys.argv) > 1 else 'c3'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
size, steps, mode = (224, 224), 10, 'residual'
mma = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != 'f32' else None
if which in ('c4', 'c5', 'c5ctx', 'c2_grad', 'fcn8dae'):
    from iterative_inference_segm_amd.fcn8 import FCN8
    dt = torch.float32
    if which == 'c4':
        B, gflop, size, concat_h = 32, 1867.8, (360, 480), ['pool4']
        dae = StandardDAE(S.make_dae_params(), 11, mma=mma)
    elif which == 'fcn8dae':
        from iterative_inference_segm_amd.fcn8 import FCN8DAE
        B, gflop, concat_h = 32, 119.24 + 10 * 125.0, ['input', 'pool3', 'pool4']
        dae = FCN8DAE(S.make_fcn8_dae_params(concat_h=concat_h, h_channels=(3, 256, 512)), 11,
                      concat_h=concat_h, mma=mma)
    elif which == 'c2_grad':
        B, gflop, concat_h, mode = 64, float('nan'), ['pool4'], 'gradient'
        dae = StandardDAE(S.make_dae_params(), 11)
    elif which == 'c5':
        B, gflop, steps, concat_h = 64, float('nan'), 50, ['pool3', 'pool4']
        dae = StandardDAE(S.make_dae_params(h_channels=(256, 512), concat_h=concat_h), 11,
                          concat_h=concat_h, pad_multi_concat=True, mma=mma)
    else:
        from iterative_inference_segm_amd.contextmod import ContextModDAE
        B, gflop, steps, concat_h = 64, float('nan'), 50, ['input']
        dae = ContextModDAE(S.make_contextmod_params(), 11)
    net = FCN8(S.make_fcn8_params(), 11, layer=concat_h + ['probs_dimshuffle'],
               mma=None if mma == 'bf16x3' else mma)
elif which == 'c3':
    from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan
    B, dt, gflop = 32, torch.float32, 254.6
    # (IISEG_C3_NET_MMA: the dense net's mode on its own -- 'bf16' = the round-2 form next to a C8 DAE)
    net = FCDenseNet(S.make_densenet_params(layer_plan()), 11, layer=['pool4'],
                     mma=os.environ.get('IISEG_C3_NET_MMA') or {'bf16x3': None}.get(mma, mma))
    dae = StandardDAE(S.make_dae_params(h_channels=(464,)), 11, padding=0, mma=mma)
else:
    from iterative_inference_segm_amd.fcn8 import FCN8
    B, dt, gflop = 16, torch.float64, 872.3
    net = FCN8(S.make_fcn8_params(), 11, layer=['pool4', 'probs_dimshuffle'], dtype=dt)
    dae = StandardDAE(S.make_dae_params(), 11, dtype=dt)
ii = IterativeInference(net, dae, 11, [11], dtype=dt)
Xs = [torch.from_numpy(S.make_images(B, size[0], size[1], seed=7 + i)).to(dt).cuda() for i in range(3)]
state = {'i': 0}
def step():
    X = Xs[state['i'] % 3]; state['i'] += 1      # a different image batch every step
    out = ii.pred_fcn_fn(X)
    ii.refine(out[:-1], out[-1], 0.1 if mode == 'residual' else 0.01, steps, early_stop=False,
              mode=mode)
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    step()
torch.cuda.synchronize()
dt_s = (time.perf_counter() - t0) / reps
print('%s%s: %dx%d, %d steps, batch %d, %.1f ms/batch, %.2f images/s, %.1f TFLOP/s nominal' %
      (which, '' if mma is None else ' [' + mma + ']', size[0], size[1], steps, B, dt_s * 1e3, B / dt_s, B / dt_s * gflop / 1e3), flush=True)
