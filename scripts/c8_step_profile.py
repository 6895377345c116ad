"""Per-launch table of conv_c8_kernel over one bench step of the bf16c8 leg (configs[1], batch 64):
HIP events around every launch behind a blocked stream (bench.conv_roofline's protocol), launches of
the same geometry merged.  Usage: python scripts/c8_step_profile.py [batch] [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get('AB_LIB'):
    from iterative_inference_segm_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
import bench
from iterative_inference_segm_amd import ops, synthetic as S

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else 'bf16c8'
cfg = sys.argv[3] if len(sys.argv) > 3 else 'c1'      # c1: configs[1] (FCN-8 + DAE); c3: configs[2] (FC-DenseNet103 + DAE)
if cfg == 'c3':
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan
    ii = IterativeInference(FCDenseNet(S.make_densenet_params(layer_plan()), 11, layer=['pool4'], mma=mode),
                            StandardDAE(S.make_dae_params(h_channels=(464,)), 11, padding=0, mma=mode), 11, [11])
else:
    ii, _, _ = bench.build_model('cuda', ['pool4'], mma=mode)
Xs = [torch.from_numpy(S.make_images(B, 224, 224, seed=1234 + 1000 * i)).cuda() for i in range(2)]
Ts = [torch.from_numpy(S.make_labels(B, 224, 224, seed=99 + 1000 * i)).cuda() for i in range(2)]
if cfg != 'c3':
    ii.prepare(B, 224, 224)
for i in range(3):
    bench.one_step(ii, Xs[i % 2], Ts[i % 2], 10, 0.1)
rows = {}
order = []
other = {}
REP = 3
for rep in range(REP):
    torch.cuda.synchronize()
    torch.cuda._sleep(int(6e8))
    ops.CONV_PROFILE = prof = []
    ops.CONV_PROFILE_INFO = info = []
    bench.one_step(ii, Xs[rep % 2], Ts[rep % 2], 10, 0.1, graph=False)
    torch.cuda.synchronize()
    ops.CONV_PROFILE = ops.CONV_PROFILE_INFO = None
    it = iter(info)
    for k, f, s, e in prof:
        ms = s.elapsed_time(e)
        if not k.startswith('conv_c8_'):
            o = other.setdefault(k, [0.0, 0])
            o[0] += ms; o[1] += 1
            continue
        g = next(it)
        if k.startswith('conv_c8_m16'):
            o = other.setdefault(k, [0.0, 0])
            o[0] += ms; o[1] += 1
            mf = other.setdefault('(m16 GFLOP)', [0.0, 0]); mf[0] += f / 1e9
            continue
        key = (g['Cin'], g['Cout'], g['OH'], g['OW'], g['unpool'], g['pool'], g['add'], g['kind'], g['flat'])
        if key not in rows:
            rows[key] = [0.0, 0.0, 0]
            order.append(key)
        r = rows[key]
        r[0] += f; r[1] += ms; r[2] += 1
tot = sum(r[1] for r in rows.values()) / REP
totf = sum(r[0] for r in rows.values()) / REP
print('%5s %5s %9s %s  n  ms/launch  ms/step  TF/s   share' % ('Cin', 'Cout', 'window', 'U P A K F'))
for key in order:
    f, ms, n = rows[key]
    print('%5d %5d %4dx%-4d %d %d %d %d %-16s %3d  %.4f  %.3f  %6.0f  %4.1f%%'
          % (key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7], str(key[8]), n // REP, ms / n,
             ms / REP, f / ms / 1e9, 100.0 * ms / REP / tot))
print('conv_c8 total %.2f ms/step, %.1f GFLOP, %.1f TF/s = %.4f of 2500' % (tot, totf / 1e9, totf / tot / 1e9, totf / tot / 1e9 / 2500))
for k, (ms, n) in other.items():
    print('other %-28s n %3d  %.3f ms/step' % (k, n // REP, ms / REP))
