"""Runs one Winograd layer a few times (for rocprofv3 --pmc): fused kernel vs GEMM + output.
Usage: python scripts/pmc_wino.py <cin> <cout> <H> <fused 0|1>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iterative_inference_segm_amd import ops
cin, cout, H, fused = (int(a) for a in sys.argv[1:5])
ops.WINO_FUSED_MAX_CIN = 4096 if fused else 0
g = torch.Generator(device='cuda').manual_seed(0)
W = torch.randn(cout, cin, 3, 3, device='cuda', generator=g) * (2.0 / (cin * 9)) ** 0.5
b = torch.randn(cout, device='cuda', generator=g) * 0.1
x = torch.rand(64, cin, H, H, device='cuda', generator=g)
conv = ops.Conv(W, b, pad=1, relu=True); conv.wino = True
out = conv(x)
for _ in range(3):
    conv(x, out=out)
torch.cuda.synchronize()
print('done')
