"""fc6 / fc7 of the FCN-8 (7x7 'valid' on pool5, 1x1) on the bf16 GEMM kernel, whole call (im2col + GEMM + output) by HIP
events.  Usage: [IISEG_BF16_GEMM_VAR=1] python scripts/fc_gemm_time.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iterative_inference_segm_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device='cuda').manual_seed(0)
for name, cin, cout, k, hw in (('fc6', 512, 4096, 7, 13), ('fc7', 4096, 4096, 1, 7)):
    W = torch.randn(cout, cin, k, k, device='cuda', generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.zeros(cout, device='cuda')
    conv = ops.Conv(W, b, pad=0, relu=True, mma='bf16')
    x = torch.rand(B, cin, hw, hw, device='cuda', generator=g)
    for _ in range(3):
        y = conv(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = conv(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * B * y.shape[2] * y.shape[3] * cout * cin * k * k
    print('%s var %s: %.4f ms/call  %.0f TFLOP/s (whole call)' % (name, os.environ.get('IISEG_BF16_GEMM_VAR', '0'), ms, fl / ms / 1e9), flush=True)
