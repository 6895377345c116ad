"""One C8 layer of configs[1] on its own, N launches (for rocprofv3 --pmc / timing A/B of kernel variants).
Usage: python3 scripts/c8_layer.py <name from scripts/bench_c8.py LAYERS> [launches] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get('AB_LIB'):
    from iterative_inference_segm_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
from iterative_inference_segm_amd import ops

LAYERS = {
    'fcn.conv1_2': (64, 64, 422, 1, (96, 230), 'pool'),
    'fcn.conv2_1': (64, 128, 211, 1, (47, 117), 'plain'),
    'fcn.conv2_2': (128, 128, 211, 1, (46, 120), 'pool'),
    'fcn.conv3_1': (128, 256, 105, 1, (22, 62), 'plain'),
    'fcn.conv3_2': (256, 256, 105, 1, (21, 64), 'plain'),
    'fcn.conv3_3': (256, 256, 105, 1, (20, 66), 'pool'),
    'fcn.conv4_1': (256, 512, 52, 1, (9, 35), 'plain'),
    'fcn.conv4_2': (512, 512, 52, 1, (8, 37), 'plain'),
    'fcn.conv4_3': (512, 512, 52, 1, (6, 40), 'pool'),
    'fcn.conv5_1': (512, 512, 26, 1, (2, 22), 'plain'),
    'fcn.conv5_2': (512, 512, 26, 1, (1, 24), 'plain'),
    'fcn.conv5_3': (512, 512, 26, 1, (0, 26), 'pool'),
    'dae.conv5_1y': (512, 1024, 26, 1, (3, 20), 'pool'),
    'dae.up_conv4': (512, 256, 52, 1, (11, 31), 'unpool'),
    'dae.conv1_1': (16, 64, 224, 100, (98, 226), 'pool'),
    'dae.conv2_1': (64, 128, 211, 1, (48, 116), 'pool'),
    'dae.conv3_1': (128, 256, 105, 1, (22, 62), 'pool'),
    'dae.conv4_1': (256, 512, 52, 1, (10, 34), 'pool'),
    'dae.conv6_1': (1024, 2048, 13, 1, (0, 13), 'pool'),
    'dae.up_conv6': (2048, 1024, 13, 1, (2, 10), 'unpool'),
    'dae.up_conv5': (1024, 512, 26, 1, (5, 17), 'unpool'),
    'dae.up_conv3': (256, 128, 105, 1, (24, 58), 'unpool'),
    'dae.up_conv2': (128, 64, 211, 1, (49, 113), 'unpool'),
    'dae.up_conv1': (64, 11, 422, 1, (99, 224), 'unpool'),
    'dae.up_conv6p': (2048, 1024, 13, 1, (2, 10), 'plain'),     # the same levels as PLAIN layers (DePool2D materialised)
    'dae.up_conv5p': (1024, 512, 26, 1, (5, 17), 'plain'),
}
name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cin, cout, H, pad, (org, win), kind = LAYERS[name]
g = torch.Generator(device='cuda').manual_seed(0)
W = torch.randn(cout, cin, 3, 3, device='cuda', generator=g) * (2.0 / (cin * 9)) ** 0.5
b = torch.randn(cout, device='cuda', generator=g) * 0.1
conv = ops.Conv(W, b, pad=pad, relu=kind != 'unpool' or cout > 16, mma='bf16c8')
window = (org, org, win, win)
if kind == 'unpool':
    h2 = H // 2
    up8 = ops.nchw_to_c8(torch.rand(B, cin, h2, h2, device='cuda', generator=g))
    mask = torch.randint(0, 16, (B, ops.c8_chunks(cin), h2, h2, 8), device='cuda', generator=g, dtype=torch.uint8)
    # (the decoder layers of the bench config carry a skip tensor: the store + bf16 C8 addend epilogue)
    fh = H + 2 * pad - 2
    skip8 = ops.nchw_to_c8(torch.rand(B, cout, fh, fh, device='cuda', generator=g)) if cout > 16 else None
    f = (lambda: conv(up8, mask_in=mask, unpool_hw=(H, H), window=window, add=skip8, add_off=(org, org))) \
        if skip8 is not None else (lambda: conv(up8, mask_in=mask, unpool_hw=(H, H), window=window))
else:
    x8 = ops.nchw_to_c8(torch.rand(B, cin, H, H, device='cuda', generator=g))
    fh = H + 2 * pad - 2
    if kind == 'pool':
        pw = conv.pool_window(H, H, window)
        po8 = ops.empty_c8(B, cout, fh // 2, fh // 2, 'cuda')
        mo8 = torch.empty(po8.shape, dtype=torch.uint8, device='cuda')
        f = lambda: conv(x8, window=pw, pool_out=po8, mask_out=mo8, store_out=False)
    else:
        f = lambda: conv(x8, window=window)
f(); f()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    f()
e1.record(); e1.synchronize()
print('%s: %.4f ms/launch over %d' % (name, e0.elapsed_time(e1) / n, n))
