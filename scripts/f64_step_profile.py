"""Per-layer table of the strict_f64 leg over one bench step (configs[1], float64, batch 32): HIP events
around every conv launch behind a blocked stream (bench.conv_roofline's protocol), launches of the same
layer geometry merged.  Usage: python scripts/f64_step_profile.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get('AB_LIB'):
    from iterative_inference_segm_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
import bench
from iterative_inference_segm_amd import ops, synthetic as S

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ii, _, _ = bench.build_model('cuda', ['pool4'], dtype=torch.float64)
Xs = [torch.from_numpy(S.make_images(B, 224, 224, seed=1234 + 1000 * i)).cuda().double() for i in range(2)]
Ts = [torch.from_numpy(S.make_labels(B, 224, 224, seed=99 + 1000 * i)).cuda().double() for i in range(2)]
ii.prepare(B, 224, 224)
for i in range(2):
    bench.one_step(ii, Xs[i % 2], Ts[i % 2], 10, 0.1)

marks = []
_orig = ops.Conv.__call__


def _wrapped(self, x1, *a, **kw):
    n0 = len(ops.CONV_PROFILE) if ops.CONV_PROFILE is not None else 0
    r = _orig(self, x1, *a, **kw)
    if ops.CONV_PROFILE is not None:
        win = kw.get('window')
        marks.append((n0, len(ops.CONV_PROFILE), self.Cin, self.Cout, self.KH,
                      tuple(win[2:]) if win is not None else tuple(r.shape[2:]) if r is not None else (0, 0),
                      kw.get('pre') is not None))
    return r


ops.Conv.__call__ = _wrapped
rows, order = {}, []
REP = 2
for rep in range(REP):
    torch.cuda.synchronize()
    torch.cuda._sleep(int(6e8))
    ops.CONV_PROFILE = prof = []
    marks.clear()
    bench.one_step(ii, Xs[rep % 2], Ts[rep % 2], 10, 0.1, graph=False)
    torch.cuda.synchronize()
    ops.CONV_PROFILE = None
    for n0, n1, cin, cout, k, win, unpool in marks:
        ms = {}
        fl = 0.0
        for kern, f, s, e in prof[n0:n1]:
            ms[kern] = ms.get(kern, 0.0) + s.elapsed_time(e)
            fl = max(fl, f)
        key = (cin, cout, k, win, unpool, tuple(sorted(ms)))
        if key not in rows:
            rows[key] = [0.0, 0.0, 0, {}]
            order.append(key)
        r = rows[key]
        r[0] += fl; r[1] += sum(ms.values()); r[2] += 1
        for kk, v in ms.items():
            r[3][kk] = r[3].get(kk, 0.0) + v
tot = sum(r[1] for r in rows.values()) / REP
print('%5s %5s k %9s U   n  ms/launch  ms/step  TF/s(nominal)  share  kernels' % ('Cin', 'Cout', 'window'))
for key in order:
    f, ms, n, per = rows[key]
    print('%5d %5d %d %4dx%-4d %d %3d  %8.4f  %7.3f  %6.1f  %4.1f%%  %s'
          % (key[0], key[1], key[2], key[3][0], key[3][1], key[4], n // REP, ms / n, ms / REP, f / ms / 1e9,
             100.0 * ms / REP / tot, ' '.join('%s=%.3f' % (k.replace('_kernel', ''), v / n) for k, v in per.items())))
print('conv total %.2f ms/step' % tot)
