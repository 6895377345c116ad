"""Per-layer conv timing of one configs[1] step (B=64): HIP events around each conv launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from iterative_inference_segm_amd import ops, synthetic as S

ii, fp, dp = bench.build_model('cuda', ['pool4'])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
X = torch.from_numpy(S.make_images(B, 224, 224)).cuda()
T = torch.from_numpy(S.make_labels(B, 224, 224)).cuda()
bench.one_step(ii, X, T, 2, 0.1)
torch.cuda.synchronize()
names = {}
for prefix, net in (('fcn.', ii.fcn), ('dae.', ii.dae)):
    for k, c in net.conv_layers().items():
        names[id(c)] = prefix + k
orig = ops.Conv.__call__
log = []
def wrapped(self, *a, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(self, *a, **kw); e1.record()
    w = kw.get('window')
    oh, ow = (w[2], w[3]) if w is not None else (out.shape[2], out.shape[3])
    log.append((names.get(id(self), '?') + ('' if w is None else ' [%dx%d]' % (oh, ow)), self.flops(out.shape[0], oh, ow), e0, e1))
    return out
ops.Conv.__call__ = wrapped
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record(); bench.one_step(ii, X, T, 2, 0.1); t1.record()
torch.cuda.synchronize()
tot = 0
for n, f, a, b in log:
    ms = a.elapsed_time(b); tot += ms
    print('%-28s %8.3f ms %7.1f TF/s' % (n, ms, f / ms / 1e9))
print('conv total %.1f ms; step (FCN + 1 DAE iter + metrics) %.1f ms' % (tot, t0.elapsed_time(t1)))
