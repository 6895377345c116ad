"""Per-launch table of the FC-DenseNet103 forward on C8 stacks (configs[2], batch 32): HIP events around every
conv launch behind a blocked stream, merged by (kernel, Cin, map).  Usage: python scripts/c3_net_profile.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get('AB_LIB'):          # A/B of two builds of the library in one gpurun call
    from iterative_inference_segm_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
from iterative_inference_segm_amd import ops, synthetic as S
from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = FCDenseNet(S.make_densenet_params(layer_plan()), 11, layer=['pool4'], mma='bf16c8')
X = torch.from_numpy(S.make_images(B, 224, 224, seed=7)).cuda()
for _ in range(2):
    net(X)
rows, order = {}, []
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); net(X); e1.record(); torch.cuda.synchronize()
print('forward wall %.3f ms' % e0.elapsed_time(e1))
torch.cuda._sleep(int(6e8))
ops.CONV_PROFILE = prof = []
ops.CONV_PROFILE_INFO = info = []
net(X)
torch.cuda.synchronize()
ops.CONV_PROFILE = ops.CONV_PROFILE_INFO = None
it = iter(info)
tot = 0.0
for k, f, s, e in prof:
    ms = s.elapsed_time(e)
    g = next(it) if k.startswith('conv_c8_') else None
    key = (k, g['Cin'] if g else int(f), g['OH'] if g else 0)
    if key not in rows:
        rows[key] = [0.0, 0.0, 0]; order.append(key)
    r = rows[key]; r[0] += f; r[1] += ms; r[2] += 1
    tot += ms
for key in order:
    f, ms, n = rows[key]
    print("%-22s %s  map %4d  n %2d  %.4f ms  %7.1f TF/s" % (key[0], ("Cin %5d" % key[1]) if key[0].startswith("conv_c8_") else ("GFLOP %5.1f" % (key[1] / 1e9)), key[2], n, ms / n, f / ms / 1e9))
print('conv launches total %.3f ms' % tot)
