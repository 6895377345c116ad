"""Per-layer timing of the implicit-GEMM conv kernel on representative shapes of configs[1]
(B=64).  Usage: python scripts/bench_conv.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from iterative_inference_segm_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = 64
# name, Cin, Cout, H, k, pad, mode
LAYERS = [
    ('conv1_2   64->64   422^2', 64, 64, 422, 3, 1, 'plain'),
    ('conv2_2  128->128  211^2', 128, 128, 211, 3, 1, 'plain'),
    ('conv3_2  256->256  105^2', 256, 256, 105, 3, 1, 'plain'),
    ('conv4_2  512->512   52^2', 512, 512, 52, 3, 1, 'plain'),
    ('dconv5_1 1024->1024 26^2', 1024, 1024, 26, 3, 1, 'concat'),
    ('dconv6_1 1024->2048 13^2', 1024, 2048, 13, 3, 1, 'plain'),
    ('up_conv6 2048->1024 13^2', 2048, 1024, 13, 3, 1, 'unpool'),
    ('up_conv4  512->256  52^2', 512, 256, 52, 3, 1, 'unpool'),
    ('up_conv2  128->64  211^2', 128, 64, 211, 3, 1, 'unpool'),
    ('up_conv1   64->11  422^2', 64, 11, 422, 3, 1, 'unpool_crop'),
    ('dconv1_1  11->64  224p100', 11, 64, 224, 3, 100, 'plain'),
    ('fc6      512->4096 7x7 13^2', 512, 4096, 13, 7, 0, 'plain'),
]
g = torch.Generator(device='cuda').manual_seed(0)
tot_f = tot_t = 0
for name, cin, cout, H, k, pad, mode in LAYERS:
    W = torch.randn(cout, cin, k, k, device='cuda', generator=g) * 0.05
    b = torch.randn(cout, device='cuda', generator=g)
    conv = ops.Conv(W, b, pad=pad, relu=True)
    kw = {}
    if mode == 'concat':
        x = torch.rand(B, cin // 2, H, H, device='cuda', generator=g)
        kw['x2'] = torch.rand(B, cin // 2, H, H, device='cuda', generator=g)
    elif mode.startswith('unpool'):
        pre = torch.relu(torch.randn(B, cin, H, H, device='cuda', generator=g))
        pooled = ops.maxpool2x2(pre)
        x = torch.randn(B, cin, H // 2, H // 2, device='cuda', generator=g)
        kw.update(pre=pre, pooled=pooled)
        if mode == 'unpool_crop':
            kw['window'] = (99, 99, 224, 224)
    else:
        x = torch.rand(B, cin, H, H, device='cuda', generator=g)
    out = conv(x, **kw)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        conv(x, out=out, **kw)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    fl = conv.flops(B, out.shape[2], out.shape[3])
    tot_f += fl; tot_t += ms
    print('%-28s %8.3f ms  %7.1f TF/s' % (name, ms, fl / ms / 1e9), flush=True)
    del conv, x, out, kw
    torch.cuda.empty_cache()
print('TOTAL %.1f ms  %.1f TF/s' % (tot_t, tot_f / tot_t / 1e9))
