"""configs[2] (FC-DenseNet103 + DAE, bf16 C8, batch 32) on an EnginePool of N engines: images/s.
Usage: [IISEG_DENSENET_GRAPH=0] python scripts/c3_pool_probe.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from iterative_inference_segm_amd import synthetic as S
from iterative_inference_segm_amd.api import IterativeInference
from iterative_inference_segm_amd.dae import StandardDAE
from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan

B = 32
Xs = [torch.from_numpy(S.make_images(B, 224, 224, seed=4000 + i)).cuda() for i in range(2)]
Ts = [torch.from_numpy(S.make_labels(B, 224, 224, seed=4100 + i)).cuda() for i in range(2)]


def build():
    net = FCDenseNet(S.make_densenet_params(layer_plan()), 11, layer=['pool4'], mma='bf16c8')
    dae = StandardDAE(S.make_dae_params(h_channels=(464,)), 11, padding=0, mma='bf16c8')
    return IterativeInference(net, dae, 11, [11])


for n in [int(a) for a in sys.argv[1:]] or [1, 3]:
    pool = bench.make_pool(n, build, Xs[0], Ts[0], 10, 0.1)
    for i in range(2 * n):
        bench.one_step(pool, Xs[i % 2], Ts[i % 2], 10, 0.1)
    torch.cuda.synchronize()
    t = time.perf_counter()
    K = 6 * n
    for i in range(K):
        bench.one_step(pool, Xs[i % 2], Ts[i % 2], 10, 0.1)
    tq = time.perf_counter() - t
    torch.cuda.synchronize()
    d = time.perf_counter() - t
    print('c3 bf16c8 in flight %d, forward graph %s: %.1f images/s, %.2f ms/batch (host queued the %d batches in %.2f ms each)'
          % (n, os.environ.get('IISEG_DENSENET_GRAPH', '1'), B * K / d, d / K * 1e3, K, tq / K * 1e3), flush=True)
    del pool
    torch.cuda.empty_cache()
