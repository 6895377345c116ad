"""Wall-clock of one bench leg on its own: configs[1], batch 64, 3 warm-up + 8 timed batches through
bench.one_step.  Usage: time_leg.py f32|bf16|bf16c8|bf16x3 [label...]  (A/B runs of kernel variants:
the same process set-up as bench.py without its other legs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get('AB_LIB'):          # A/B of two builds of the library in one gpurun call
    from iterative_inference_segm_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
import bench
from iterative_inference_segm_amd import synthetic as S

mode = sys.argv[1]
ii, _, _ = bench.build_model('cuda', ['pool4'], mma=None if mode == 'f32' else mode)
B = 64
Xs = [torch.from_numpy(S.make_images(B, 224, 224, seed=1234 + 1000 * i)).cuda() for i in range(2)]
Ts = [torch.from_numpy(S.make_labels(B, 224, 224, seed=99 + 1000 * i)).cuda() for i in range(2)]
ii.prepare(B, 224, 224)
for i in range(3):
    bench.one_step(ii, Xs[i % 2], Ts[i % 2], 10, 0.1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(8):
    bench.one_step(ii, Xs[i % 2], Ts[i % 2], 10, 0.1)
torch.cuda.synchronize()
d = (time.perf_counter() - t0) / 8
print(mode, sys.argv[2:], '%.2f ms/batch  %.1f img/s' % (d * 1e3, B / d), flush=True)
