"""The roofline pass of the bf16 leg (bench.conv_roofline) with both timing methods on the same box:
HIP events on each kernel's dispatch vs events recorded on the stream around each launch.
Usage: python scripts/event_overhead.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from iterative_inference_segm_amd import synthetic as S
B = 64
for mode in ('bf16c8', None):
    ii, _, _ = bench.build_model('cuda', ['pool4'], mma=mode)
    X = torch.from_numpy(S.make_images(B, 224, 224, seed=1234)).cuda()
    T = torch.from_numpy(S.make_labels(B, 224, 224, seed=99)).cuda()
    ii.prepare(B, 224, 224)
    for _ in range(3):
        bench.one_step(ii, X, T, 10, 0.1)
    for how in ('dispatch', 'stream', 'dispatch', 'stream'):
        os.environ['IISEG_BENCH_EVENTS'] = how
        rl = bench.conv_roofline(ii, X, T, 10, 0.1, 1.0, 2500.0 if mode else 157.3)
        print('%-7s %-8s %-22s n %3d  avg %.4f ms  kernel ms/step %.2f  achieved %.1f TF/s  frac %.4f  all conv %.2f ms'
              % (mode or 'f32', how, rl['kernel'], rl['launches_per_step'], rl['avg_launch_ms'], rl['kernel_ms_per_step'],
                 rl['achieved'], rl['frac'], rl['all_conv_ms_per_step']), flush=True)
    del ii
    torch.cuda.empty_cache()
