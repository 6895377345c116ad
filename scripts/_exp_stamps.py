import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iterative_inference_segm_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_exp_lib.so')
import torch
from iterative_inference_segm_amd import ops
cin, cout, H, win = (int(a) for a in sys.argv[1:5])
mma = sys.argv[5] if len(sys.argv) > 5 else 'bf16'
pool = len(sys.argv) > 6
g = torch.Generator(device='cuda').manual_seed(0)
W = torch.randn(cout, cin, 3, 3, device='cuda', generator=g) * (2.0 / (cin * 9)) ** 0.5
b = torch.randn(cout, device='cuda', generator=g) * 0.1
x = torch.rand(64, cin, H, H, device='cuda', generator=g)
ops.BF16_WINO_MIN_CIN = 4096
conv = ops.Conv(W, b, pad=1, relu=True, mma=mma)
org = (H - win) // 2
kw = dict(window=(org, org, win, win))
if pool:
    kw['pool_out'] = torch.empty(64, cout, H // 2, H // 2, device='cuda')
    kw['mask_out'] = torch.empty(64, cout, H // 2, H // 2, device='cuda', dtype=torch.uint8)
out = conv(x, **kw)
for _ in range(3):
    conv(x, out=out, **kw)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 64)()
print('rc', lib.iiseg_debug_stamps(buf))
s = list(buf)
print('kernel total %d  prologue %d  loop %d  epilogue %d' % (s[3] - s[0], s[1] - s[0], s[2] - s[1], s[3] - s[2]))
if mma == 'bf16':
    print('k-tile 3: issue loads %d  mfma phase %d  vmcnt(0) %d  barrier1 %d  cvt+store %d  barrier2 %d  (sum %d)' % (
        s[11] - s[10], s[12] - s[11], s[13] - s[12], s[14] - s[13], s[15] - s[14], s[16] - s[15], s[16] - s[10]))
else:
    print('k-tile 3 (4 channels): mfma+issue %d  store+vmcnt %d  barrier %d (sum %d)' % (s[11] - s[10], s[12] - s[11], s[13] - s[12], s[13] - s[10]))
