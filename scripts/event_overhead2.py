"""Is the eager roofline pass slower than the graph-replayed timed steps?  bf16 leg, batch 64: wall of one step
(a) as timed (graph replay), (b) eager (graph=False) behind a blocked stream, (c) eager with dispatch timing on;
and the sum of all dispatch-timed kernels of (c)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from iterative_inference_segm_amd import ops, synthetic as S
B = 64
ii, _, _ = bench.build_model('cuda', ['pool4'], mma='bf16c8')
X = torch.from_numpy(S.make_images(B, 224, 224, seed=1234)).cuda()
T = torch.from_numpy(S.make_labels(B, 224, 224, seed=99)).cuda()
ii.prepare(B, 224, 224)
for _ in range(3):
    bench.one_step(ii, X, T, 10, 0.1)
def wall(graph, prof=False, n=3):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        torch.cuda._sleep(int(6e8))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        if prof:
            ops.profile_begin()
        e0.record()
        bench.one_step(ii, X, T, 10, 0.1, graph=graph)
        e1.record()
        torch.cuda.synchronize()
        s = None
        if prof:
            ops.profile_end(); s = sum(ops.PROFILE_MS); nk = len(ops.PROFILE_MS)
        out.append((e0.elapsed_time(e1), s))
    return out
print('graph replay  :', wall(None))
print('eager         :', wall(False))
print('eager+dispatch:', wall(False, True))
print('graph replay  :', wall(None))
