import os, time, torch
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for p in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(p, open(p).read().strip())
    except OSError as e: print(p, 'n/a')
print('torch threads default', torch.get_num_threads())
import sys; sys.path.insert(0, '.')
from oracle import torch_cpu as tcpu
from iterative_inference_segm_amd import synthetic as S
Pf, Pd = tcpu.prepare_params(S.make_fcn8_params()), tcpu.prepare_params(S.make_dae_params())
X = S.make_images(1, 224, 224, seed=7)
for th in (16, 32, 64):
    torch.set_num_threads(th)
    t = time.time(); tcpu.run_batch(Pf, Pd, X, 0.1, 2, True); print('threads', th, '1 img 2 steps', round(time.time() - t, 2), flush=True)
