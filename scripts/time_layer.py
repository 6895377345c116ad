"""Times one conv layer (HIP events, reps launches).  Usage: time_layer.py cin cout H mma [batch] [window] [origin]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iterative_inference_segm_amd import ops
cin, cout, H = (int(a) for a in sys.argv[1:4]); mma = sys.argv[4]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
win = int(sys.argv[6]) if len(sys.argv) > 6 else 0
g = torch.Generator(device='cuda').manual_seed(0)
W = torch.randn(cout, cin, 3, 3, device='cuda', generator=g) * (2.0 / (cin * 9)) ** 0.5
b = torch.randn(cout, device='cuda', generator=g) * 0.1
x = torch.rand(B, cin, H, H, device='cuda', generator=g)
conv = ops.Conv(W, b, pad=1, relu=True, mma=mma)
org = int(sys.argv[7]) if len(sys.argv) > 7 else (H - win) // 2
kw = dict(window=(org, org, win, win)) if win else {}
out = conv(x, **kw)
torch.cuda._sleep(int(3e8))
ops.CONV_PROFILE = prof = []
for _ in range(5):
    conv(x, out=out, **kw)
torch.cuda.synchronize()
ops.CONV_PROFILE = None
per = {}
for k, f, s, e in prof:
    per.setdefault(k, []).append(s.elapsed_time(e))
print(os.environ.get('IISEG_BF16_DEBUG', '0'), ' '.join('%s=%.3f' % (k, sorted(v)[len(v) // 2]) for k, v in per.items()), flush=True)
