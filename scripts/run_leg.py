"""One leg of the bench workload on its own (for rocprofv3): FCN-8 + standard DAE, configs[1], N
batches of B images through bench.one_step.  Usage: run_leg.py f32|bf16|bf16c8|f64 [B] [batches]
(two warm-up batches first: border stores primed, steady-state windows met)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from iterative_inference_segm_amd import synthetic as S

mode = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dtype = torch.float64 if mode == 'f64' else torch.float32
ii, _, _ = bench.build_model('cuda', ['pool4'], dtype=dtype, mma=None if mode in ('f32', 'f64') else mode)
Xs = [torch.from_numpy(S.make_images(B, 224, 224, seed=1234 + 1000 * i)).to('cuda', dtype) for i in range(2)]
Ts = [torch.from_numpy(S.make_labels(B, 224, 224, seed=99 + 1000 * i)).to('cuda', dtype) for i in range(2)]
ii.prepare(B, 224, 224)
for i in range(2 + n):
    bench.one_step(ii, Xs[i % 2], Ts[i % 2], 10, 0.1)
torch.cuda.synchronize()
print('done', mode, B, n)
