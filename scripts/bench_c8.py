"""Layer-by-layer A/B: the bf16 direct kernel on fp32 NCHW activations (conv_halo_bf16 / Winograd,
mma='bf16') against the same layer on bf16 C8 activations (conv_c8_bf16, mma='bf16c8') at the
configs[1] launch geometries (batch 64, the loop's steady-state windows).  Median of 7 launches, HIP
events.  Usage: python scripts/bench_c8.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iterative_inference_segm_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
# name, Cin, Cout, map H (=W) of the conv INPUT, pad, window (origin, size) of the output, kind
LAYERS = [
    ('fcn.conv1_2', 64, 64, 422, 1, (96, 230), 'plain'),
    ('fcn.conv2_2', 128, 128, 211, 1, (46, 120), 'plain'),
    ('fcn.conv3_2', 256, 256, 105, 1, (21, 64), 'plain'),
    ('fcn.conv4_2', 512, 512, 52, 1, (8, 37), 'plain'),
    ('fcn.conv5_2', 512, 512, 26, 1, (1, 24), 'plain'),
    ('dae.conv1_1', 16, 64, 224, 100, (98, 226), 'pool'),
    ('dae.conv2_1', 64, 128, 211, 1, (48, 116), 'pool'),
    ('dae.conv3_1', 128, 256, 105, 1, (22, 62), 'pool'),
    ('dae.conv4_1', 256, 512, 52, 1, (10, 33), 'pool'),
    ('dae.conv5_1y', 512, 1024, 26, 1, (4, 19), 'plain'),
    ('dae.conv6_1', 1024, 2048, 13, 1, (1, 12), 'plain'),
    ('dae.up_conv6', 2048, 1024, 13, 1, (2, 10), 'unpool'),
    ('dae.up_conv5', 1024, 512, 26, 1, (5, 17), 'unpool'),
    ('dae.up_conv4', 512, 256, 52, 1, (11, 31), 'unpool'),
    ('dae.up_conv3', 256, 128, 105, 1, (24, 58), 'unpool'),
    ('dae.up_conv2', 128, 64, 211, 1, (49, 113), 'unpool'),
    ('dae.up_conv1', 64, 11, 422, 1, (99, 224), 'unpool'),
]
only = os.environ.get('ONLY')


def med(fn, n=7):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[n // 2]


g = torch.Generator(device='cuda').manual_seed(0)
tot_old = tot_new = 0.0
for name, cin, cout, H, pad, (org, win), kind in LAYERS:
    if only and only not in name:
        continue
    W = torch.randn(cout, cin, 3, 3, device='cuda', generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, device='cuda', generator=g) * 0.1
    c_old = ops.Conv(W, b, pad=pad, relu=True, mma='bf16')
    c_new = ops.Conv(W, b, pad=pad, relu=True, mma='bf16c8')
    window = (org, org, win, win)
    if kind == 'unpool':
        h2 = H // 2
        up = torch.rand(B, cin, h2, h2, device='cuda', generator=g)
        pre = torch.rand(B, cin, H, H, device='cuda', generator=g)
        pooled = torch.nn.functional.max_pool2d(pre, 2)
        mask = torch.randint(0, 16, (B, ops.c8_chunks(cin), h2, h2, 8), device='cuda', generator=g,
                             dtype=torch.uint8)
        maskb = torch.randint(0, 16, (B, cin, h2, h2), device='cuda', generator=g, dtype=torch.uint8)
        up8 = ops.nchw_to_c8(up)
        if c_old.mask_ok():
            f_old = lambda: c_old(up, mask_in=maskb, unpool_hw=(H, H), window=window)
        else:
            f_old = lambda: c_old(up, pre=pre, pooled=pooled, window=window)
        f_new = lambda: c_new(up8, mask_in=mask, unpool_hw=(H, H), window=window)
    else:
        x = torch.rand(B, cin, H, H, device='cuda', generator=g)
        x8 = ops.nchw_to_c8(x)
        if kind == 'pool':
            pw = c_new.pool_window(H, H, window)
            fh = H + 2 * pad - 2
            po = torch.empty(B, cout, fh // 2, fh // 2, device='cuda')
            mo = torch.empty(B, cout, fh // 2, fh // 2, device='cuda', dtype=torch.uint8)
            po8 = ops.empty_c8(B, cout, fh // 2, fh // 2, 'cuda')
            mo8 = torch.empty(po8.shape, dtype=torch.uint8, device='cuda')
            if c_old.mask_ok() and c_old.pool_fusable():
                f_old = lambda: c_old(x, window=pw, pool_out=po, mask_out=mo, store_out=False)
            else:
                f_old = lambda: c_old(x, window=window)
            f_new = lambda: c_new(x8, window=pw, pool_out=po8, mask_out=mo8, store_out=False)
        else:
            f_old = lambda: c_old(x, window=window)
            f_new = lambda: c_new(x8, window=window)
    t_old, t_new = med(f_old), med(f_new)
    gf = 2.0 * cin * cout * 9 * win * win * B / 1e9
    tot_old += t_old; tot_new += t_new
    print('%-14s %4d->%4d  %3d^2  old %.3f ms (%5.0f TF/s)   c8 %.3f ms (%5.0f TF/s)  x%.2f'
          % (name, cin, cout, win, t_old, gf / t_old, t_new, gf / t_new, t_old / t_new), flush=True)
print('sum old %.3f ms  c8 %.3f ms' % (tot_old, tot_new))
