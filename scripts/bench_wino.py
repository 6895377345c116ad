"""Winograd F(2x2,3x3) vs direct static-tap conv on the wide 3x3 layers of configs[1] (B=64).
Per layer: direct ms, Winograd ms (input / GEMM / output split), speed-up, max-abs difference.
Usage: python scripts/bench_wino.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iterative_inference_segm_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = 64
# name, Cin, Cout, H (input), pad, window or None
LAYERS = [
    ('conv1_2    64->64   win228', 64, 64, 422, 1, (97, 97, 228, 228)),
    ('conv2_2   128->128  win119', 128, 128, 211, 1, (46, 46, 119, 119)),
    ('conv2_2   128->128  full211', 128, 128, 211, 1, None),
    ('conv3_2   256->256  win64', 256, 256, 105, 1, (20, 20, 64, 64)),
    ('conv3_2   256->256  full105', 256, 256, 105, 1, None),
    ('conv4_2   512->512  win37', 512, 512, 52, 1, (7, 7, 37, 37)),
    ('conv4_2   512->512  full52', 512, 512, 52, 1, None),
    ('conv5_2   512->512  26^2', 512, 512, 26, 1, None),
    ('dconv5_1y 512->1024 win19', 512, 1024, 26, 1, (3, 3, 19, 19)),
    ('dconv6_1 1024->2048 13^2', 1024, 2048, 13, 1, None),
    ('up_conv6 2048->1024 win10', 2048, 1024, 13, 1, (1, 1, 10, 10)),
    ('up_conv5 1024->512  win17', 1024, 512, 26, 1, (4, 4, 17, 17)),
    ('up_conv4  512->256  win31', 512, 256, 52, 1, (10, 10, 31, 31)),
    ('up_conv3  256->128  win58', 256, 128, 105, 1, (23, 23, 58, 58)),
    ('up_conv2  128->64   win113', 128, 64, 211, 1, (49, 49, 113, 113)),
]
g = torch.Generator(device='cuda').manual_seed(0)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


for name, cin, cout, H, pad, win in LAYERS:
    W = torch.randn(cout, cin, 3, 3, device='cuda', generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, device='cuda', generator=g) * 0.1
    x = torch.rand(B, cin, H, H, device='cuda', generator=g)
    direct = ops.Conv(W, b, pad=pad, relu=True); direct.wino = False
    wino = ops.Conv(W, b, pad=pad, relu=True); wino.wino = True
    kw = dict(window=win) if win else {}
    od, ow = direct(x, **kw), wino(x, **kw)
    err = float((od - ow).abs().max()); scale = float(od.abs().max())
    md = timed(lambda: direct(x, out=od, **kw))
    mw = timed(lambda: wino(x, out=ow, **kw))
    ops.CONV_PROFILE = prof = []
    wino(x, out=ow, **kw)
    torch.cuda.synchronize()
    ops.CONV_PROFILE = None
    st = [s.elapsed_time(e) for _, _, s, e in prof] + [0.0]
    gf = prof[1][1]
    fl = direct.flops(B, od.shape[2], od.shape[3])
    print('%-28s direct %7.3f ms %6.1f TF/s | wino %7.3f ms (in %.3f gemm %.3f [%5.1f TF/s] out %.3f) '
          'x%.2f | maxdiff %.2e / %.1f' % (name, md, fl / md / 1e9, mw, st[0], st[1], gf / st[1] / 1e9,
                                          st[2], md / mw, err, scale), flush=True)
    del direct, wino, x, od, ow
    torch.cuda.empty_cache()
