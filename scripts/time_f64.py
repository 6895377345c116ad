"""Wall clock of the strict_f64 leg alone (configs[1], float64, batch 32): 1 warm-up + N timed batches through
bench.one_step, and the same with the refinement loop launched eagerly (graph=False).
Usage: python scripts/time_f64.py [batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from iterative_inference_segm_amd import synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ii, _, _ = bench.build_model('cuda', ['pool4'], dtype=torch.float64)
Xs = [torch.from_numpy(S.make_images(B, 224, 224, seed=1234 + 1000 * i)).cuda().double() for i in range(2)]
Ts = [torch.from_numpy(S.make_labels(B, 224, 224, seed=99 + 1000 * i)).cuda().double() for i in range(2)]
ii.prepare(B, 224, 224)
for graph in (None, False):
    for i in range(2):
        bench.one_step(ii, Xs[i % 2], Ts[i % 2], 10, 0.1, graph=graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        bench.one_step(ii, Xs[i % 2], Ts[i % 2], 10, 0.1, graph=graph)
    torch.cuda.synchronize()
    d = (time.perf_counter() - t0) / n
    print('f64 graph=%s: %.2f ms/batch  %.1f img/s' % (graph, d * 1e3, B / d), flush=True)
