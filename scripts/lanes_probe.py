"""Concurrency probes, N engines launched from one thread on N HIP streams: 'split' = one batch as N concurrent
sub-batches; 'pipe' = whole batches round-robin over N engines (batch i on engine i % N).  images/s per
(mode, batch, N).  Usage: python scripts/lanes_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from iterative_inference_segm_amd import synthetic as S

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cases = [('bf16c8', 64, 'pipe', (1, 2, 3)), ('bf16c8', 10, 'pipe', (1, 2, 3)), (None, 64, 'pipe', (1, 2, 3)),
         ('bf16c8', 64, 'split', (2,)), (None, 64, 'split', (2,))]
if os.environ.get('PROBE') == 'small':
    cases = [('bf16c8', 10, 'pipe', (1, 3, 4, 6)), ('bf16c8', 64, 'pipe', (1, 2))]
for mode, B, kind, lanes in cases:
    Xs = [torch.from_numpy(S.make_images(B, 224, 224, seed=1234 + 1000 * i)).cuda() for i in range(3)]
    Ts = [torch.from_numpy(S.make_labels(B, 224, 224, seed=99 + 1000 * i)).cuda() for i in range(3)]
    for n in lanes:
        part = B // n if kind == 'split' else B
        engines = [bench.build_model('cuda', ['pool4'], mma=mode)[0] for _ in range(n)]
        streams = [torch.cuda.Stream() for _ in range(n)] if n > 1 else [torch.cuda.current_stream()]
        for e, s in zip(engines, streams):
            with torch.cuda.stream(s):
                e.prepare(part, 224, 224)
        torch.cuda.synchronize()

        def stepn(i):
            if kind == 'pipe':
                k = i % n
                with torch.cuda.stream(streams[k]):
                    bench.one_step(engines[k], Xs[i % 3], Ts[i % 3], 10, 0.1)
                return
            for k, (e, s) in enumerate(zip(engines, streams)):
                with torch.cuda.stream(s):
                    bench.one_step(e, Xs[i % 3][k * part:(k + 1) * part], Ts[i % 3][k * part:(k + 1) * part], 10, 0.1)
        for i in range(3 * n):
            stepn(i)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(steps):
            stepn(i)
        tq = time.perf_counter() - t
        torch.cuda.synchronize()
        d = time.perf_counter() - t
        print('%-7s batch %3d %-5s lanes %d: %8.1f images/s  %.2f ms/batch (host queued a batch in %.2f ms)'
              % (mode or 'fp32', B, kind, n, B * steps / d, d / steps * 1e3, tq / steps * 1e3), flush=True)
        del engines, streams
        torch.cuda.empty_cache()
