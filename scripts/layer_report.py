"""Per-layer conv timing of one configs[1] batch in steady state (B=64): for every conv call the
kernels it launched (HIP events from ops.CONV_PROFILE), issued GFLOP, TFLOP/s, and the gap to
an MFMA-bound launch at 131 TFLOP/s (the measured ceiling of the GEMM kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from iterative_inference_segm_amd import ops, synthetic as S

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
MMA = sys.argv[2] if len(sys.argv) > 2 else None          # 'bf16': the 16-bit MFMA path
ii, fp, dp = bench.build_model('cuda', ['pool4'], mma=MMA)
NIT = 2
Xs = [torch.from_numpy(S.make_images(B, 224, 224, seed=s)).cuda() for s in (1, 2)]
T = torch.from_numpy(S.make_labels(B, 224, 224)).cuda()
bench.one_step(ii, Xs[0], T, NIT, 0.1)          # folds the borders
torch.cuda.synchronize()
names = {}
for prefix, net in (('fcn.', ii.fcn), ('dae.', ii.dae)):
    for k, c in net.conv_layers().items():
        names[id(c)] = prefix + k
    for k, (ch, cy) in getattr(net, 'hsplit', {}).items():
        names[id(ch)], names[id(cy)] = prefix + k + '(h)', prefix + k + '(y)'
orig = ops.Conv.__call__
calls = []
ops.CONV_PROFILE = prof = []
def wrapped(self, *a, **kw):
    n0 = len(prof)
    out = orig(self, *a, **kw)
    w = kw.get('window')
    calls.append((names.get(id(self), '?') + ('' if w is None else ' [%dx%d@%d,%d]' % (w[2], w[3], w[0], w[1])),
                  n0, len(prof)))
    return out
ops.Conv.__call__ = wrapped
torch.cuda._sleep(int(8e8))     # the host enqueues the whole step behind this: no launch gaps in the brackets
bench.one_step(ii, Xs[1], T, NIT, 0.1, graph=False)
torch.cuda.synchronize()
tot = gap_tot = 0
for name, n0, n1 in calls:
    ent = prof[n0:n1]
    ms = sum(s.elapsed_time(e) for _, _, s, e in ent)
    fl = sum(f for _, f, _, _ in ent)
    kern = ' '.join('%s=%.3f' % (k.replace('_kernel', '').replace('_f32', '').replace('wino_', 'w_'), s.elapsed_time(e)) for k, _, s, e in ent)
    gap = ms - fl / 131e9
    tot += ms; gap_tot += gap
    print('%-34s %7.3f ms %7.1f GF %6.1f TF/s  gap %6.3f ms  %s' % (name, ms, fl / 1e9, fl / ms / 1e9, gap, kern))
print('conv total %.1f ms, gap to 131 TF/s on issued flops %.1f ms (FCN + %d DAE steps)' % (tot, gap_tot, NIT))
