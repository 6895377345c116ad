#!/usr/bin/env python3
"""Headline benchmark: refined images/s of the iterative-inference hot path on MI355X.

Workload (BASELINE.json configs[1]): FCN-8 + standard DAE (n_filters=64, additional_pool=2,
concat_h=['pool4'], trackind unpool, skip), 11 classes, synthetic random 224x224x3 images,
batch 64 per GPU, 10 refinement steps (step 0.1, early stop disabled so the work is fixed),
fp32 HIP kernels.  One "step" = one batch through the per-batch path of reference
iterative_inference.py:226-294: pred_fcn_fn -> val_fn (FCN metrics, :242) -> pred_dae_fn + val_fn
(one-shot DAE metrics, :250-251; the same DAE forward as the loop's first step, shared) ->
refine x10 (:258-284) -> val_fn (:287).

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment: this process only LAUNCHES
`python -m torch.distributed.run --nproc-per-node N ... bench.py ...` as a child (before anything
here touches the GPU), forwards rank 0's JSON line and exits with the child's code.  Under
torch.distributed.run (the driver's own launch line) every rank runs `worker()`.

Work accounting.  `value` is measured with the exact work eliminations on:
decoder levels computed only on the window that reaches the final center crop; inside the 10-step
loop only the y-dependent part of the DAE encoder maps recomputed; the weights-only pad-100 borders
folded once per input geometry at load time (`ii.prepare`, from an all-zero image) -- every timed
step runs on a DIFFERENT image batch.  All are tested to give BIT-IDENTICAL refined maps.  The JSON
also carries the run with the cross-batch border stores off (`per_batch_only`) and with every
elimination off (`full_recompute`, all 872.3 nominal GFLOP/image executed).

Scheduling.  Every leg runs with whole batches in flight (api.EnginePool, `--in-flight N`, default 2 -- one more for
batches of 32 images and fewer, two more for 16 and fewer): batch i on engine i % N, each engine on its own HIP stream with its own sessions /
graphs / scratch.  Every batch goes through exactly the launches of the single-engine path (bit-identical results,
tests/test_gpu_e2e.py); `ms_per_step` is timed seconds / K.  `one_in_flight` next to a value is the same leg on one
engine; the roofline passes time one engine's launches one at a time.

Extra objects on the ONE JSON line rank 0 prints:
  roofline     the dominant kernel from HIP events around every conv launch (extra, untimed pass
               behind a queued-up stream so that no bracket contains host launch gaps), plus
               `whole_path` = executed conv FLOPs / ms_per_step
  strict_f64   the float64 path (the reference's CPU numerics) on the same config: the number
               that carries the 1e-4 parity claim end to end
  cpu_baseline float32 torch-CPU restatement of the same loop on the host cores (oracle/, "port"),
               reference-faithful per-image schedule and a batched one; rank 0, N = 1 only
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from iterative_inference_segm_amd import dist as iidist  # noqa: E402
from iterative_inference_segm_amd.api import EnginePool  # noqa: E402
from iterative_inference_segm_amd import synthetic as S  # noqa: E402

N_CLASSES = 11
GFLOP_PER_IMAGE = 872.3          # 119.24 (FCN-8) + 10 x 75.31 (DAE), nominal
PEAK_TFLOPS_F32_MFMA = 157.3     # MI355X_MICROARCH.md, fp32 matrix peak
PEAK_TFLOPS_F64_MFMA = 78.6      # v_mfma_f64_16x16x4_f64: half the fp32 matrix rate
PEAK_TFLOPS_BF16_MFMA = 2500.0   # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 / fp16 (no sparsity)


def build_model(device, concat_h, dtype=torch.float32, mma=None, fcn_mma='same'):
    # (mma='bf16x3': the DAE loop on hi / lo pairs, the FCN-8 -- once per batch -- on its fp32 kernels:
    # tests/test_gpu_x3.py::test_x3_engine_free_running_fixed_tolerance)
    if fcn_mma == 'same':
        fcn_mma = None if mma == 'bf16x3' else mma
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.fcn8 import FCN8
    fp = S.make_fcn8_params(seed=1234)      # same seed on every rank: replicated weights
    dp = S.make_dae_params(seed=4321)
    fcn = FCN8(fp, N_CLASSES, layer=concat_h + ['probs_dimshuffle'], device=device, dtype=dtype,
               mma=fcn_mma)
    dae = StandardDAE(dp, N_CLASSES, concat_h=concat_h, padding=100, n_filters=64,
                      additional_pool=2, skip=True, unpool_type='trackind', device=device,
                      dtype=dtype, mma=mma)
    return IterativeInference(fcn, dae, N_CLASSES, [N_CLASSES], device=device, dtype=dtype), fp, dp


def make_pool(n, build, X, T, num_iter, step_size, prepare=True):
    """`n` engines from `build()` behind an api.EnginePool, load-time work done: the weights-only borders
    of this geometry (`prepare`) and one batch per engine, which brings the lazily built state into being
    (packed weights, workspaces, the captured HIP graph of the refinement step).  Untimed set-up, like
    `prepare`; the W warm-up steps of the contract come after it."""
    pool = EnginePool([build() for _ in range(max(1, n))])
    if prepare:
        pool.prepare(X.shape[0], X.shape[2], X.shape[3])
    if len(pool) > 1:
        for _ in range(len(pool)):
            one_step(pool, X, T, num_iter, step_size)
        pool.synchronize()
    return pool


def one_step(ii, X, T, num_iter, step_size, graph=None):
    """One batch of the per-batch path (iterative_inference.py:237-287); returns the device-side
    metric accumulators (refined, FCN, one-shot DAE).  graph=False: every step launched from
    Python (the roofline pass needs its per-launch events)."""
    if isinstance(ii, EnginePool):
        # whole batches in flight: this batch on the pool's next engine / stream (api.EnginePool)
        with ii.lane(X, T) as engine:
            return one_step(engine, X, T, num_iter, step_size, graph)
    out = ii.pred_fcn_fn(X)                                            # :237-239
    H, Y = out[:-1], out[-1]
    m_fcn = ii.val_device(Y, T)                                        # :242
    Yii, _, _, R0 = ii.refine(H, Y, step_size, num_iter, early_stop=False,
                              first_reconstruction=True, graph=graph)  # :258-284
    m_dae = ii.val_device(R0, T)                                       # :250-251 (shared forward)
    return ii.val_device(Yii, T), m_fcn, m_dae                         # :287


def effective_cores():
    """Host cores this process may actually use: the scheduler affinity capped by the cgroup CPU
    quota (a GPU box exposes all 256 hardware threads to a container limited to 16 CPUs; running
    128 BLAS threads on those is several times slower than 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(fp, dp, num_iter, step_size, budget_s=75.0):
    """SURVEY 8(d) / BASELINE.md section 4: the float32 torch-CPU restatement of the identical loop
    (oracle/torch_cpu.py) on all host cores, 1 warm-up batch, then timed batches of 10 images -- 3 of
    them when that fits `budget_s` per schedule, fewer / smaller otherwise (the sample is stated).
    `value` is the reference-faithful per-image (B = 1) schedule of iterative_inference.py:258-284;
    `batched` runs the DAE on the whole batch per step."""
    from oracle import torch_cpu as tcpu
    cores = effective_cores()
    torch.set_num_threads(max(1, cores))
    Pf, Pd = tcpu.prepare_params(fp), tcpu.prepare_params(dp)
    res = {}
    for name, per_image in (('per_image', True), ('batched', False)):
        t0 = time.perf_counter()
        tcpu.run_batch(Pf, Pd, S.make_images(2, 224, 224, seed=700), step_size, num_iter, per_image)
        t_img = (time.perf_counter() - t0) / 2            # warm-up batch (2 images), not reported
        if t_img > budget_s:                              # a very slow host: the warm-up is the sample
            res[name] = {'value': round(1.0 / t_img, 4), 'images': 2, 'batches': [2],
                         'seconds': round(2 * t_img, 2)}
            continue
        # configs[0] / iterative_inference.py:117: batches of 10; as many of them (up to 3) as the
        # budget holds, else one smaller batch -- the shortfall is on the line (`batches`)
        n_img = int(max(2, min(30, budget_s / max(t_img, 1e-3))))
        sizes = [10] * (n_img // 10) if n_img >= 10 else [n_img]
        t0 = time.perf_counter()
        for i, b in enumerate(sizes):
            tcpu.run_batch(Pf, Pd, S.make_images(b, 224, 224, seed=701 + i), step_size, num_iter,
                           per_image)
        dt = time.perf_counter() - t0
        res[name] = {'value': round(sum(sizes) / dt, 4), 'images': sum(sizes), 'batches': sizes,
                     'seconds': round(dt, 2)}
    v = res['per_image']['value']
    return {'value': v, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'cpu_model': _cpu_model(), 'torch_threads': torch.get_num_threads(),
            'host_hw_threads': os.cpu_count(),
            'gflops_nominal': round(v * GFLOP_PER_IMAGE, 1),
            'batched': res['batched'], 'per_image': res['per_image'],
            'sample': 'float32 torch-CPU restatement (oracle/torch_cpu.py; CPU restatement, baseline '
                      'only) of FCN-8 + %d DAE steps at 224x224: 1 warm-up batch of 2 images, then '
                      'batches of %s images (reference-faithful per-image B=1 schedule, %.1f s) and '
                      '%s images (batched schedule, %.1f s)'
                      % (num_iter, res['per_image']['batches'], res['per_image']['seconds'],
                         res['batched']['batches'], res['batched']['seconds'])}


def _parity_report():
    """Rows of the newest committed end-to-end parity report (profiles/rNN_parity_damped_64.md, written
    by scripts/parity_report.py on an MI355X): {mode: {...}} + the file name, or ({}, None)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_parity_damped_64.md')))
    if not files:
        return {}, None
    rows = {}
    for line in open(files[-1]):
        m = re.match(r'\| ([^|]+?) \| ([0-9.]+) \| ([0-9.e+-]+) \| ([0-9.e+-]+) \| ([0-9.]+) \| ([0-9.]+) \|', line)
        if m:
            rows[m.group(1)] = {'pixels_within_1e-4': float(m.group(2)), 'max_err': float(m.group(3)),
                                'mean_err': float(m.group(4)), 'argmax_agreement': float(m.group(5)),
                                'miou': float(m.group(6))}
    return rows, os.path.relpath(files[-1], ROOT)


def _traffic(kernel):
    """HBM GB per launch of `kernel` from the committed rocprofv3 PMC artefact (two --pmc passes,
    FETCH_SIZE x2 correction + WRITE_SIZE; scripts/make_profiles.py pmc), with its provenance; None
    when the artefact has no row for this kernel."""
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic_latest.json')
    try:
        art = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    gb = art.get('gb_per_launch', {}).get(kernel)
    return gb, {'file': 'profiles/hbm_traffic_latest.json', 'commit': art.get('commit'),
                'source': art.get('source')}


def conv_roofline(ii, X, T, num_iter, step_size, ms_per_step, peak, mode='f32'):
    """Extra (untimed) pass with HIP events immediately around every conv launch, on the launch
    stream.  The stream is first blocked by a spin kernel (torch.cuda._sleep) long enough for the
    host to enqueue the whole step behind it, so the GPU then runs launch after launch and a
    bracket never contains a host-side launch gap: all_conv_ms_per_step <= ms_per_step on any host."""
    from iterative_inference_segm_amd import ops
    torch.cuda.synchronize()
    torch.cuda._sleep(int(6e8))          # ~0.25-0.3 s of GPU time: the host runs ahead meanwhile
    # Every kernel launch of the pass carries its own start / stop HIP events on its dispatch
    # (include/iiseg.h iiseg_profile_begin): a launch's time is the kernel's execution time on the launch
    # stream, as rocprofv3 reports it.  (Events recorded separately around a launch put two barrier
    # packets between consecutive kernels: +7 % on a 0.18 ms kernel, profiles/r04_event_overhead.md.)
    dispatch = os.environ.get('IISEG_BENCH_EVENTS', 'dispatch') == 'dispatch'
    if dispatch:
        ops.profile_begin()
    ops.CONV_PROFILE = prof = []
    ops.KERNEL_BYTES.clear()
    one_step(ii, X, T, num_iter, step_size, graph=False)
    torch.cuda.synchronize()
    ops.CONV_PROFILE = None
    if dispatch:
        ops.profile_end()
    per = {}
    for k, f, s, e in prof:
        ent = per.setdefault(k, [0.0, 0.0, 0])
        ent[0] += f; ent[1] += s.elapsed_time(e); ent[2] += 1
    # dominant kernel = largest total time among the matrix-core kernels
    kern = max((k for k in per if per[k][0] > 0), key=lambda k: per[k][1])
    flops, ms, n = per[kern]
    all_ms = sum(v[1] for v in per.values())
    all_gflop = sum(f for _, f, _, _ in prof) / 1e9
    achieved = flops / (ms * 1e-3) / 1e12
    gb, prov = _traffic(kern)
    whole = all_gflop / ms_per_step          # GFLOP / ms = TFLOP/s
    if ops.KERNEL_BYTES.get(kern):
        # a vector-ALU kernel bound by HBM traffic (csrc/conv_small.hip: the context module's 11 -> 11 layers):
        # algorithmic bytes = every input plane read once + every output plane written once
        gbs = ops.KERNEL_BYTES[kern] / (ms * 1e-3) / 1e9
        return {'bound': 'hbm', 'kernel': kern, 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': None,
                'timing': 'HIP events on each kernel dispatch' if dispatch else 'HIP events on the stream',
                'launches_per_step': n, 'avg_launch_ms': round(ms / n, 4),
                'gb_per_launch': round(ops.KERNEL_BYTES[kern] / n / 1e9, 4), 'kernel_ms_per_step': round(ms, 2),
                'per_kernel_ms_per_step': {k: round(v[1], 2) for k, v in per.items()},
                'per_kernel_tflops': {k: round(v[0] / v[1] / 1e9, 1) for k, v in per.items() if v[0]},
                'all_conv_ms_per_step': round(all_ms, 2),
                'whole_path': {'executed_tflops': round(whole, 2), 'frac': round(whole / peak, 4)}}
    # every 3x3 launch of the step on the 16-bit pipe (north_star: ">= 40 % of fp16 MFMA peak for the 3x3
    # convs"): conv_c8_kernel (64-row tiles) + conv_c8_m16_kernel (the class-score layer / dense-block layers)
    k33 = [k for k in per if k.startswith('conv_c8_kernel') or k == 'conv_c8_m16_kernel']
    all33 = None
    if k33 and sum(per[k][1] for k in k33) > 0:
        f33, m33 = sum(per[k][0] for k in k33), sum(per[k][1] for k in k33)
        all33 = {'tflops': round(f33 / m33 / 1e9, 1), 'frac': round(f33 / m33 / 1e9 / peak, 4),
                 'ms_per_step': round(m33, 2), 'gflop_per_step': round(f33 / 1e9, 1),
                 'launches_per_step': sum(per[k][2] for k in k33), 'kernels': sorted(k33)}
    return {'bound': 'mfma', 'kernel': kern, 'achieved': round(achieved, 2), 'peak': peak,
            'all_3x3': all33,
            'timing': 'HIP events on each kernel dispatch (hipExtLaunchKernelGGL start / stop)' if dispatch
                      else 'HIP events recorded on the stream around each launch',
            'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': gb,
            'traffic_unit': 'GB of HBM traffic per launch (rocprofv3 PMC passes)',
            'traffic_provenance': prov,
            'launches_per_step': n, 'avg_launch_ms': round(ms / n, 4),
            'gflop_per_launch': round(flops / n / 1e9, 3), 'kernel_ms_per_step': round(ms, 2),
            'per_kernel_ms_per_step': {k: round(v[1], 2) for k, v in per.items()},
            'per_kernel_tflops': {k: round(v[0] / v[1] / 1e9, 1) for k, v in per.items() if v[0]},
            'all_conv_ms_per_step': round(all_ms, 2),
            'all_conv_gflop_per_step': round(all_gflop, 1),
            'whole_path': {'executed_tflops': round(whole, 2), 'frac': round(whole / peak, 4),
                           'note': 'FLOPs the conv kernels actually issue per step (Winograd layers '
                                   'count their 4/9) / ms_per_step of the timed run / MFMA peak'}}


HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E ~8 TB/s
GFLOP_C3, GFLOP_C4 = 254.6, 1867.8      # nominal GFLOP per refined image, configs[2] / [3]


def other_configs(device, step_size, no_roofline, n_fly=1):
    """BASELINE configs[2], [3], [4] (parity-test cases, not the headline): one warm-up + 3 timed
    batches each through the same per-batch path as the headline (`one_step`), and the conv roofline
    of the dominant kernel from HIP events (the same untimed extra pass).  Single GPU only."""
    from iterative_inference_segm_amd.api import IterativeInference
    from iterative_inference_segm_amd.contextmod import ContextModDAE
    from iterative_inference_segm_amd.dae import StandardDAE
    from iterative_inference_segm_amd.densenet import FCDenseNet, layer_plan
    from iterative_inference_segm_amd.fcn8 import FCN8

    def c3(mma):
        net = FCDenseNet(S.make_densenet_params(layer_plan()), N_CLASSES, layer=['pool4'], device=device, mma=mma)
        dae = StandardDAE(S.make_dae_params(h_channels=(464,)), N_CLASSES, padding=0, device=device, mma=mma)
        return IterativeInference(net, dae, N_CLASSES, [N_CLASSES], device=device)

    def fcn(concat_h, mma):
        return FCN8(S.make_fcn8_params(seed=1234), N_CLASSES, layer=concat_h + ['probs_dimshuffle'],
                    device=device, mma=mma)

    def c4(mma):
        dae = StandardDAE(S.make_dae_params(seed=4321), N_CLASSES, device=device, mma=mma)
        return IterativeInference(fcn(['pool4'], mma), dae, N_CLASSES, [N_CLASSES], device=device)

    def c5i(mma):
        # (the context module itself is fp32 on the vector ALU; `mma` picks the FCN-8 host's path)
        dae = ContextModDAE(S.make_contextmod_params(), N_CLASSES, device=device)
        return IterativeInference(fcn(['input'], mma), dae, N_CLASSES, [N_CLASSES], device=device)

    def c5ii(mma):
        ch = ['pool3', 'pool4']
        dae = StandardDAE(S.make_dae_params(h_channels=(256, 512), concat_h=ch), N_CLASSES, concat_h=ch,
                          pad_multi_concat=True, device=device, mma=mma)
        return IterativeInference(fcn(ch, mma), dae, N_CLASSES, [N_CLASSES], device=device)

    def c1(mma):
        dae = StandardDAE(S.make_dae_params(seed=4321), N_CLASSES, device=device, mma=mma)
        return IterativeInference(fcn(['pool4'], mma), dae, N_CLASSES, [N_CLASSES], device=device)

    cases = [  # key, builder, mma, batch, (H, W), steps, nominal GFLOP / image, what
        ('c1_batch10_f32', c1, None, 10, (224, 224), 10, 872.3,
         'configs[0] on the GPU: the reference\'s own batch of 10 (the batch the CPU baseline is timed on), fp32'),
        ('c1_batch10_bf16c8', c1, 'bf16c8', 10, (224, 224), 10, 872.3, 'the same on bf16 C8'),
        ('c3_f32', c3, None, 32, (224, 224), 10, GFLOP_C3,
         'configs[2]: FC-DenseNet103 + standard DAE (padding 0, h = pool4 464 ch), fp32'),
        ('c3_bf16c8', c3, 'bf16c8', 32, (224, 224), 10, GFLOP_C3,
         'configs[2] as written (bf16, fp32 accumulate): dense-block stacks and DAE on bf16 C8'),
        ('c4_f32', c4, None, 32, (360, 480), 10, GFLOP_C4,
         'configs[3] per GPU: FCN-8 + standard DAE, 360x480 CamVid frames, batch 32 (= 256 / 8), fp32'),
        ('c4_bf16c8', c4, 'bf16c8', 32, (360, 480), 10, GFLOP_C4, 'the same on bf16 C8'),
        ('c5_i_contextmod', c5i, None, 64, (224, 224), 50, None,
         "configs[4] variant (i), SURVEY A9': contextmod DAE, concat_h=['input'], 50 steps (reference-exact)"),
        ('c5_i_bf16c8_host', c5i, 'bf16c8', 64, (224, 224), 50, None,
         'the same DAE (fp32) behind an FCN-8 host on bf16 C8: the host forward is a quarter of the fp32 batch'),
        ('c5_ii_f32', c5ii, None, 64, (224, 224), 50, None,
         "configs[4] variant (ii), build-defined: standard DAE, concat_h=['pool3','pool4'], pad-100, 50 steps"),
        ('c5_ii_bf16c8', c5ii, 'bf16c8', 64, (224, 224), 50, None, 'the same on bf16 C8'),
    ]
    res = {}
    for key, build, mma, B, (H, W), steps, gflop, what in cases:
        Xs = [torch.from_numpy(S.make_images(B, H, W, seed=4000 + i)).to(device) for i in range(2)]
        Ts = [torch.from_numpy(S.make_labels(B, H, W, seed=4100 + i)).to(device) for i in range(2)]
        # batches of 32 and fewer leave more of the chip idle per launch: one more batch in flight, two more for
        # 16 and fewer (measured, configs[2] 2 -> 3: 2418 -> 2719; batch 10, 3 -> 4: 1943 -> 2100, 1677 -> 2350
        # images/s; at batch 64 no difference)
        nf = n_fly + (2 if B <= 16 else 1 if B <= 32 else 0) if n_fly > 1 else n_fly
        pool = make_pool(nf, lambda: build(mma), Xs[0], Ts[0], steps, step_size)
        ii = pool.engines[0]
        nt = 3 * nf
        one_step(pool, Xs[0], Ts[0], steps, step_size)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nt):
            one_step(pool, Xs[(i + 1) % 2], Ts[(i + 1) % 2], steps, step_size)
        torch.cuda.synchronize()
        d = (time.perf_counter() - t0) / nt
        ent = {'what': what, 'value': round(B / d, 2), 'unit': 'images/s', 'ms_per_step': round(d * 1e3, 2),
               'batch': B, 'size': [H, W], 'num_iter': steps, 'timed_batches': nt, 'in_flight': nf,
               'dtype': 'f32' if mma is None else 'bf16 operands, f32 accumulate, bf16 C8 activations'}
        if key == 'c5_i_bf16c8_host':
            ent['dtype'] = 'FCN-8 host: bf16 operands, f32 accumulate, bf16 C8 activations; context module: f32'
        if gflop is not None:
            ent['nominal_equivalent_tflops'] = round(B / d * gflop / 1e3, 1)
        if not no_roofline:
            rl = conv_roofline(ii, Xs[0], Ts[0], steps, step_size, d * 1e3,
                               PEAK_TFLOPS_F32_MFMA if mma is None else PEAK_TFLOPS_BF16_MFMA)
            ent['roofline'] = {k: rl[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac',
                                                  'launches_per_step', 'avg_launch_ms', 'kernel_ms_per_step',
                                                  'all_conv_ms_per_step', 'per_kernel_ms_per_step',
                                                  'per_kernel_tflops', 'all_3x3') if k in rl}
            if 'gb_per_launch' in rl:
                ent['roofline']['gb_per_launch'] = rl['gb_per_launch']
            if key != 'c5_i_bf16c8_host':       # (two pipes in one path: no single peak to price the whole against)
                ent['roofline']['whole_path_frac'] = rl['whole_path']['frac']
        if nf > 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(3):
                one_step(ii, Xs[(i + 1) % 2], Ts[(i + 1) % 2], steps, step_size)
            torch.cuda.synchronize()
            ent['one_in_flight'] = round(3 * B / (time.perf_counter() - t0), 2)
        res[key] = ent
        del ii, pool, Xs, Ts
        torch.cuda.empty_cache()
    return res


def _write_detail(line):
    """Everything measured (per-kernel tables, every config's roofline, notes) goes to a side file -- and to
    stderr -- so that the ONE line on stdout stays short enough for the legs a reader needs to survive in a
    truncated log tail (the bf16 leg was cut off the driver's record)."""
    text = json.dumps(line)
    sys.stderr.write('bench.py detail: ' + text + '\n')
    sys.stderr.flush()
    for d in (os.path.join(ROOT, 'gpurun_out'), ROOT):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, 'bench_detail.json')
            with open(path, 'w') as f:
                f.write(text + '\n')
            return os.path.relpath(path, ROOT)
        except OSError:
            continue
    return None


def _rl_short(rl, extra=()):
    if not rl:
        return None
    keys = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'launches_per_step',
            'avg_launch_ms') + tuple(extra)
    out = {k: rl[k] for k in keys if k in rl}
    if isinstance(rl.get('whole_path'), dict):
        out['whole_path_frac'] = rl['whole_path']['frac']
    elif 'whole_path_frac' in rl:
        out['whole_path_frac'] = rl['whole_path_frac']
    if rl.get('all_3x3'):
        out['all_3x3_frac'] = rl['all_3x3']['frac']
        out['all_3x3_tflops'] = rl['all_3x3']['tflops']
    return out


def compact_line(line, detail_file):
    """The line rank 0 prints: the contract's keys first, short summaries of the side legs, and LAST --
    where a truncated tail keeps them -- `strict_f64`, `cpu_baseline` and the 16-bit MFMA leg `bf16` with
    its dominant-kernel fraction and the fraction over every 3x3 launch (`roofline.all_3x3_frac`)."""
    out = {k: line[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
                                'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config')
           if k in line}
    if 'roofline' in line:
        out['roofline'] = _rl_short(line['roofline'], ('gflop_per_launch', 'kernel_ms_per_step',
                                                       'all_conv_ms_per_step'))
    if 'in_flight' in line:
        out['in_flight'] = line['in_flight']['engines']
        if 'one_in_flight' in line['in_flight']:
            out['one_in_flight'] = line['in_flight']['one_in_flight']['value']
    out['detail'] = detail_file
    for k in ('per_rank_images_per_s', 'metric_all_reduce_ms', 'executed_gflop_per_image'):
        if k in line:
            out[k] = line[k]
    if 'miou' in line:
        out['miou'] = {k: line['miou'][k] for k in ('iterative_inference', 'fcn', 'dae_one_shot')}
    if 'parity' in line:
        d = line['parity'].get('damped_set_64_images', {})
        out['parity_fp32_damped64'] = {k: d[k] for k in ('pixels_within_1e-4', 'max_err', 'miou') if k in d}
    for k in ('per_batch_only', 'full_recompute'):
        if k in line:
            out[k] = line[k]['value']
    if 'concurrent_streams' in line:
        out['concurrent_streams'] = {k: v['value'] for k, v in line['concurrent_streams'].items()
                                     if isinstance(v, dict)}
    if 'early_stop' in line:
        out['early_stop'] = line['early_stop']
    if 'configs' in line:
        cf = {}
        for k, e in line['configs'].items():
            c = {'value': e['value'], 'ms': e['ms_per_step']}
            if 'one_in_flight' in e:
                c['in_flight'], c['one_in_flight'] = e['in_flight'], e['one_in_flight']
            rl = e.get('roofline')
            if rl:
                c['kernel'], c['frac'] = rl['kernel'], rl['frac']
                if 'whole_path_frac' in rl:
                    c['whole'] = rl['whole_path_frac']
                if rl.get('all_3x3'):
                    c['all_3x3'] = rl['all_3x3']['frac']
            for kk in ('parity', 'forward_ms'):
                if kk in e:
                    c[kk] = e[kk]
            cf[k] = c
        out['configs'] = cf
    if 'bf16x3' in line:
        e = line['bf16x3']
        out['bf16x3'] = {'value': e['value'], 'ms_per_step': e['ms_per_step'],
                         'roofline': _rl_short(e.get('roofline')),
                         'fcn_on_pairs_too': e.get('fcn_on_pairs_too', {}).get('value')}
    # ---- the tail ----
    if 'strict_f64' in line:
        e = line['strict_f64']
        out['strict_f64'] = {k: e[k] for k in ('value', 'unit', 'ms_per_step', 'dtype', 'batch', 'in_flight')
                             if k in e}
    if 'cpu_baseline' in line:
        e = line['cpu_baseline']
        out['cpu_baseline'] = {'value': e['value'], 'unit': e['unit'], 'cores': e['cores'], 'kind': e['kind'],
                               'batched': e['batched']['value'],
                               'sample': 'fp32 torch-CPU restatement (oracle/torch_cpu.py), per-image B=1 '
                                         'schedule, %s images in %.0f s on %d cores'
                                         % (e['per_image']['images'], e['per_image']['seconds'], e['cores'])}
    if 'bf16' in line:
        e = line['bf16']
        b = {'value': e['value'], 'unit': e['unit'], 'ms_per_step': e['ms_per_step'], 'mode': e['mode'],
             'in_flight': e.get('in_flight', 1), 'one_in_flight': e.get('one_in_flight', {}).get('value'),
             'miou_iterative_inference': e['miou_iterative_inference'],
             'delta_miou_vs_f32': e['delta_miou_vs_f32']}
        d = e.get('damped_set_64_images', {})
        b['parity_damped64'] = {k: d[k] for k in ('argmax_agreement', 'miou') if k in d}
        if 'concurrent_streams' in e:
            b['concurrent_streams'] = {k: v['value'] for k, v in e['concurrent_streams'].items()
                                       if isinstance(v, dict)}
        b['roofline'] = _rl_short(e.get('roofline'), ('gflop_per_launch', 'kernel_ms_per_step'))
        out['bf16'] = b
    return out


def timed_steps(ii, Xs, Ts, steps, warmup, num_iter, step_size, world, device, start=0):
    """W untimed + exactly K timed steps, barrier + synchronize on both sides, max over ranks.
    Returns (seconds, per-step metric accumulators)."""
    it, n = start, len(Xs)
    for _ in range(warmup):
        one_step(ii, Xs[it % n], Ts[it % n], num_iter, step_size)
        it += 1
    torch.cuda.synchronize()
    iidist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    results = []
    for _ in range(steps):
        results.append(one_step(ii, Xs[it % n], Ts[it % n], num_iter, step_size))
        it += 1
    return t0, results


def launch_workers(args, argv):
    """Parent of an N-GPU run: never touches the GPU, starts torch.distributed.run as a CHILD
    process (one rank per GPU), forwards its output and exits with its code."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC only on this pool (RCCL)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    line = None
    for out in proc.stdout:
        # stdout carries ONE line, the result; whatever else the workers or their libraries print
        # there (e.g. gloo's connection notes) goes to stderr
        if out.lstrip().startswith('{"metric"'):
            line = out
            sys.stdout.write(out)
            sys.stdout.flush()
        else:
            sys.stderr.write(out)
            sys.stderr.flush()
    rc = proc.wait()
    if rc == 0 and line is None:
        sys.stderr.write('bench.py: workers exited without a result line\n')
        rc = 1
    return rc


def dry_run(args):
    """CPU rehearsal of the multi-rank protocol (no HIP work): rendezvous, barrier-bracketed
    timing with max over ranks, the metric all-reduce, rank 0's JSON line.  `value` is null."""
    rank, world, device = iidist.init_from_env('cpu')
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    iidist.barrier()
    t0 = time.perf_counter()
    acc = iidist.EvalAccumulator(N_CLASSES)
    for i in range(args.steps):
        cm = np.zeros((N_CLASSES, N_CLASSES + 1))
        cm[rank % N_CLASSES, rank % N_CLASSES] = 1 + i
        acc.add_batch(cm, 1.0, 0.0)
    acc.all_reduce(device)
    iidist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        print(json.dumps({'metric': 'refined images/s (dry run: protocol only, no GPU work)',
                          'value': None, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
                          'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                          'dtype': 'f32', 'data': 'none', 'dry_run': True,
                          'batches_reduced': acc.results()[4],
                          'config': {'workload': 'dry run', 'parallelism': 'dp%d' % world}}),
              flush=True)
    iidist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU per step')
    ap.add_argument('--num_iter', type=int, default=10)
    ap.add_argument('--step_size', type=float, default=0.1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-full-recompute', action='store_true',
                    help='skip the extra timed runs with the work eliminations off')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-bf16', action='store_true', help='skip the 16-bit MFMA (bf16) leg')
    ap.add_argument('--no-bf16x3', action='store_true',
                    help='skip the split-operand (bf16x3, fp32-class on the 16-bit pipe) leg')
    ap.add_argument('--bf16-mode', default='bf16c8', choices=['bf16c8', 'bf16'],
                    help="activations of the bf16 leg: 'bf16c8' = bf16 C8 chunks between the layers "
                         "(conv_c8_bf16.hip), 'bf16' = fp32 NCHW (round-2 form)")
    ap.add_argument('--in-flight', type=int, default=2,
                    help='whole batches in flight: engines of the api.EnginePool every leg runs on (each on '
                         'its own HIP stream, batches round-robin; 1 = the single-stream path)')
    ap.add_argument('--split-streams', action='store_true',
                    help='also run the legs that split ONE batch into concurrent sub-batches on several HIP '
                         'streams (the round-3/4 scheduling probe; whole batches in flight supersede it)')
    ap.add_argument('--no-two-streams', action='store_true', help='(accepted and ignored: see --split-streams)')
    ap.add_argument('--streams', default='2', help='comma-separated stream counts of those legs')
    ap.add_argument('--all-legs', action='store_true',
                    help='N > 1: also run the per_batch_only / full_recompute / bf16 / strict_f64 legs '
                         '(by default a multi-GPU run times the headline leg only)')
    ap.add_argument('--cpu-budget', type=float, default=75.0,
                    help='seconds per CPU-baseline schedule (3 batches of 10 images need ~55 s on 16 cores)')
    ap.add_argument('--no-strict-f64', action='store_true',
                    help='skip the float64 (strict parity) leg')
    ap.add_argument('--no-early-stop', action='store_true',
                    help='skip the leg with the stop test on (damped set, num_iter 50)')
    ap.add_argument('--no-configs', action='store_true',
                    help='skip the other BASELINE configs (configs[2], [3], [4]: 3 timed batches each)')
    ap.add_argument('--dry-run', action='store_true',
                    help='CPU rehearsal of the launch / reduction protocol, no GPU work')
    args = ap.parse_args()

    # N > 1 outside a torch.distributed environment: fan out FIRST, before any GPU call
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_workers(args, sys.argv[1:]))
    if args.dry_run:
        return dry_run(args)
    worker(args)


def worker(args):
    rank, world, device = iidist.init_from_env('cuda')
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    concat_h = ['pool4']
    if world > 1 and not args.all_legs:
        # a scaling run measures the headline leg; the ablation legs are single-GPU diagnostics
        args.no_full_recompute = args.no_bf16 = args.no_strict_f64 = True
        args.no_bf16x3 = args.no_configs = args.no_early_stop = True
    args.no_two_streams = not args.split_streams
    n_fly = max(1, args.in_flight)
    built = []

    def build_main():
        built.append(build_model(device, concat_h))
        return built[-1][0]
    B = args.batch
    # weak scaling: every rank refines its own shard of `B` synthetic images per step;
    # distinct image batches per step (up to 4, then rotating): nothing image-dependent can be
    # carried from one step to the next
    n_distinct = max(1, min(args.steps + args.warmup, 4))
    Xs = [torch.from_numpy(S.make_images(B, 224, 224, seed=1234 + rank + 1000 * i)).to(device)
          for i in range(n_distinct)]
    Ts = [torch.from_numpy(S.make_labels(B, 224, 224, seed=99 + rank + 1000 * i)).to(device)
          for i in range(n_distinct)]
    # load-time constant folding of the weights-only borders for this geometry (from a zero image)
    pool = make_pool(n_fly, build_main, Xs[0], Ts[0], args.num_iter, args.step_size)
    ii, fp, dp = built[0]

    t0, results = timed_steps(pool, Xs, Ts, args.steps, args.warmup, args.num_iter, args.step_size,
                              world, device)
    # the path's only collective: one all-reduce of the metric accumulator (RCCL over xGMI)
    accs = [iidist.EvalAccumulator(N_CLASSES) for _ in range(3)]
    for ms in results:
        for acc, m in zip(accs, ms):
            a, j, mse = m.result()
            acc.add_batch(m.cm.cpu().numpy(), a, mse)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0           # this rank's own K steps (before the collective)
    t_ar = time.perf_counter()
    for acc in accs:
        acc.all_reduce(device)
    torch.cuda.synchronize()
    t_ar = time.perf_counter() - t_ar
    iidist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_rank = [B * args.steps / t_local]
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
        mine = torch.tensor([per_rank[0]], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        per_rank = [float(t.item()) for t in allr]

    def leg(model, xs, ts, steps, warmup):
        t1, _ = timed_steps(model, xs, ts, steps, warmup, args.num_iter, args.step_size, world, device)
        torch.cuda.synchronize()
        iidist.barrier()
        d = time.perf_counter() - t1
        if world > 1:
            tmax = torch.tensor([d], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            d = float(tmax.item())
        v = world * xs[0].shape[0] * steps / d
        return {'value': round(v, 3), 'unit': 'images/s', 'ms_per_step': round(d / steps * 1e3, 2)}

    images = world * B * args.steps
    value = images / dt
    ms_per_step = dt / args.steps * 1e3
    line = {
        'metric': 'refined images/s (FCN-8 + standard DAE, 10-step iterative inference, 224x224, '
                  '11 classes)',
        'value': round(value, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 2),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic (seeded uniform images, blob labels, He-uniform random weights)',
        'config': {'workload': 'BASELINE configs[1]: FCN-8 + standard DAE (f64c1p2, pool4, '
                               'trackind, skip), 224x224x3, batch %d/GPU, %d steps, step %.2g, '
                               'early stop off; per batch: pred_fcn_fn, val_fn, pred_dae_fn+val_fn, '
                               'refine, val_fn' % (B, args.num_iter, args.step_size),
                   'global_batch': world * B, 'parallelism': 'dp%d' % world},
        'nominal_equivalent_tflops': {
            'value': round(value * GFLOP_PER_IMAGE / 1e3, 2),
            'note': 'images/s x 872.3 nominal GFLOP/image (SURVEY 6.2).  NOT a hardware rate: the '
                    'exact work eliminations and Winograd issue far fewer FLOPs; the hardware rate '
                    'is roofline.whole_path'},
        'parity': {'fp32_value': 'damped synthetic set, all 10 steps free-running: <= 1e-4 with the float64 '
                                 'mask decisions, >= 0.999 of pixels within 1e-4 with its own masks (the '
                                 'rest: verified DePool2D near-tie flips; measured fraction / max / mean in '
                                 '`damped_set_64_images`) (tests/test_gpu_damped.py); chaotic default set '
                                 '(the weights this line is timed on): teacher-forced 1e-4, free-running '
                                 'only statistically (DESIGN.md section 4)',
                   'strict_1e-4_end_to_end': 'strict_f64 leg (float64 = the reference CPU numerics)'},
    }
    prows, pfile = _parity_report()
    line['parity']['damped_set_64_images'] = dict(prows.get('fp32 MFMA', {}), source=pfile,
                                                  reference='float64 HIP path, pinned to the CPU oracle')
    line['in_flight'] = {
        'engines': n_fly,
        'note': 'whole batches in flight (api.EnginePool): batch i runs on engine i %% %d, each engine on its '
                'own HIP stream with its own sessions / graphs / scratch; every batch goes through exactly '
                'the launches of the single-engine path (same kernels, same batch size, bit-identical '
                'results: tests/test_gpu_e2e.py); ms_per_step = timed seconds / K, i.e. the rate, not the '
                'latency of one batch' % n_fly}
    if n_fly > 1:
        line['in_flight']['one_in_flight'] = leg(ii, Xs, Ts, args.steps, 1)
    line['per_rank_images_per_s'] = [round(v, 1) for v in per_rank]
    line['metric_all_reduce_ms'] = round(t_ar * 1e3, 3)
    line['distinct_image_batches'] = n_distinct   # rotated through the timed steps (bit-identity of the
    #                                              work eliminations: tests/test_gpu_e2e.py)
    _, acc, miou, _, nb = accs[0].results()
    _, acc_f, miou_f, _, _ = accs[1].results()
    _, acc_d, miou_d, _, _ = accs[2].results()
    line['miou'] = {'iterative_inference': round(miou, 5), 'fcn': round(miou_f, 5),
                    'dae_one_shot': round(miou_d, 5), 'acc_ii': round(acc, 5),
                    'acc_fcn': round(acc_f, 5), 'acc_dae': round(acc_d, 5), 'batches': nb,
                    'note': 'consistency metric (random weights), reduced over ranks'}
    X, T = Xs[0], Ts[0]
    if not args.no_roofline:
        from iterative_inference_segm_amd import ops as _ops
        # (IISEG_MMA=bf16 forces the 16-bit operand path onto this leg too -- the profiling runs of
        # scripts/profile_r02.sh do: the leg then says so and is priced against the bf16 peak)
        main_bf16 = _ops.DEFAULT_MMA == 'bf16'
        if main_bf16:
            line['dtype'] = 'bf16 operands, f32 accumulate (IISEG_MMA=bf16)'
        rl = conv_roofline(ii, X, T, args.num_iter, args.step_size, ms_per_step,
                           PEAK_TFLOPS_BF16_MFMA if main_bf16 else PEAK_TFLOPS_F32_MFMA)
        line['roofline'] = rl
        line['executed_gflop_per_image'] = round(rl['all_conv_gflop_per_step'] / B, 1)

    if not args.no_full_recompute:
        # same timing protocol with the exact work eliminations switched off, in two stages:
        #   per_batch_only : nothing is kept from one batch to the next (no weights-only border
        #                    stores); decoder DCE / in-loop invariants / h-half stay on
        #   full_recompute : every layer of every step and batch recomputed in full
        def knobs(border, loop):
            for e in pool.engines:
                e.fcn.fold_border = e.dae.fold_border = border
                e.dae.dce = e.dae.licm = loop
        knobs(False, True)
        line['per_batch_only'] = dict(leg(pool, Xs, Ts, args.steps, n_fly), note='IISEG_FCN_BORDER_FOLD=0 '
                                      'IISEG_DAE_BORDER_FOLD=0: no state carried between batches')
        knobs(False, False)
        line['full_recompute'] = dict(
            leg(pool, Xs, Ts, args.steps, n_fly),
            note='IISEG_DECODER_DCE=0 IISEG_ENCODER_LICM=0 IISEG_FCN_BORDER_FOLD=0 '
                 'IISEG_DAE_BORDER_FOLD=0: all 872.3 nominal GFLOP/image recomputed in full every '
                 'step and batch (same kernels)')
        knobs(True, True)
    def concurrent_leg(mma):
        """The same batch as N concurrent sub-batches: N engines (own nets, sessions, graphs and
        scratch), each on its own HIP stream, so that the tail of one engine's kernels (the last,
        partly filled round of workgroups of every launch) overlaps the others'.  Same kernels, same
        per-image results (an image's result does not depend on its batch:
        test_full_config_batch_properties); reported next to a leg's value, never as it."""
        res = {}
        for ns in [int(v) for v in args.streams.split(',') if v]:
            if ns < 2 or B % ns:
                continue
            part = B // ns
            engines = [build_model(device, concat_h, mma=mma)[0] for _ in range(ns)]
            streams = [torch.cuda.Stream(device=device) for _ in range(ns)]
            for e, st_ in zip(engines, streams):
                with torch.cuda.stream(st_):
                    e.prepare(part, 224, 224)
            torch.cuda.synchronize()

            def stepn(x, t):
                for k, (e, st_) in enumerate(zip(engines, streams)):
                    with torch.cuda.stream(st_):
                        one_step(e, x[k * part:(k + 1) * part], t[k * part:(k + 1) * part],
                                 args.num_iter, args.step_size)

            it2 = 0
            for _ in range(max(args.warmup, 2)):
                stepn(Xs[it2 % n_distinct], Ts[it2 % n_distinct]); it2 += 1
            torch.cuda.synchronize()
            iidist.barrier()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                stepn(Xs[it2 % n_distinct], Ts[it2 % n_distinct]); it2 += 1
            torch.cuda.synchronize()
            iidist.barrier()
            d2 = time.perf_counter() - t2
            res[str(ns)] = {'value': round(world * B * args.steps / d2, 3), 'unit': 'images/s',
                            'ms_per_step': round(d2 / args.steps * 1e3, 2), 'sub_batch': part}
            del engines, streams
            torch.cuda.empty_cache()
        res['note'] = ('the batch of %d as N concurrent sub-batches on N HIP streams (N engines); a '
                       'scheduling variant of the same work, not the leg\'s value' % B)
        return res

    if not args.no_two_streams:
        line['concurrent_streams'] = concurrent_leg(None)
    if not args.no_bf16:
        # 16-bit MFMA leg (VERDICT row N1; north_star: ">= 1000 images/s at >= 40 % of fp16 MFMA
        # peak"): bf16 operands + fp32 accumulation on the wide 3x3 layers, everything else as in
        # the fp32 run.  Statistical parity only; the headline `value` stays the fp32 line.
        pool16 = make_pool(n_fly, lambda: build_model(device, concat_h, mma=args.bf16_mode)[0], Xs[0], Ts[0],
                           args.num_iter, args.step_size)
        ii16 = pool16.engines[0]
        t1, res16 = timed_steps(pool16, Xs, Ts, args.steps, args.warmup, args.num_iter, args.step_size,
                                world, device)
        acc16 = iidist.EvalAccumulator(N_CLASSES)
        for ms in res16:
            a, j, mse = ms[0].result()
            acc16.add_batch(ms[0].cm.cpu().numpy(), a, mse)
        acc16.all_reduce(device)
        torch.cuda.synchronize()
        iidist.barrier()
        d16 = time.perf_counter() - t1
        if world > 1:
            tmax = torch.tensor([d16], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            d16 = float(tmax.item())
        v16 = world * B * args.steps / d16
        _, a16, miou16, _, _ = acc16.results()
        leg16 = {'value': round(v16, 3), 'unit': 'images/s',
                 'dtype': 'bf16 operands, f32 accumulate' +
                          (', bf16 C8 activations between the 3x3 layers' if args.bf16_mode == 'bf16c8'
                           else ', fp32 NCHW activations'),
                 'mode': args.bf16_mode, 'in_flight': n_fly,
                 'ms_per_step': round(d16 / args.steps * 1e3, 2),
                 'miou_iterative_inference': round(miou16, 5),
                 'delta_miou_vs_f32': round(miou16 - miou, 5),
                 'parity': 'statistical (north_star: mIoU within +-0.05): on the damped synthetic set '
                           'refined argmax agreement with float64 >= 0.99 and mIoU within 0.05 of 1 '
                           '(tests/test_gpu_damped.py; measured: `damped_set_64_images`); per-layer '
                           'relative RMS error ~3e-3 (tests/test_gpu_bf16.py, tests/test_gpu_c8.py)',
                 'damped_set_64_images': dict(prows.get(
                     'bf16 operands, bf16 C8 activations' if args.bf16_mode == 'bf16c8'
                     else 'bf16 operands, fp32 activations', {}), source=pfile)}
        if not args.no_roofline:
            leg16['roofline'] = conv_roofline(ii16, X, T, args.num_iter, args.step_size,
                                              d16 / args.steps * 1e3, PEAK_TFLOPS_BF16_MFMA)
        if n_fly > 1:
            leg16['one_in_flight'] = leg(ii16, Xs, Ts, args.steps, 1)
        if not args.no_two_streams:
            leg16['concurrent_streams'] = concurrent_leg(args.bf16_mode)
        line['bf16'] = leg16
        del ii16, pool16
        torch.cuda.empty_cache()
    if not args.no_bf16x3:
        # the fp32-CLASS mode of the 16-bit matrix pipe (VERDICT row N1, mode (ii)): the DAE loop on
        # bf16 hi / lo pairs (16 significant bits per operand, x_lo W_hi + x_hi W_lo + x_hi W_hi in one
        # fp32 accumulation, csrc/conv_c8_bf16.hip X3), the FCN-8 on its fp32 MFMA kernels.  A leg:
        # the headline `value` stays the fp32-MFMA line.
        poolx3 = make_pool(n_fly, lambda: build_model(device, concat_h, mma='bf16x3')[0], Xs[0], Ts[0],
                           args.num_iter, args.step_size)
        iix3 = poolx3.engines[0]
        t1, resx3 = timed_steps(poolx3, Xs, Ts, args.steps, args.warmup, args.num_iter, args.step_size,
                                world, device)
        accx3 = iidist.EvalAccumulator(N_CLASSES)
        for ms in resx3:
            a, j, mse = ms[0].result()
            accx3.add_batch(ms[0].cm.cpu().numpy(), a, mse)
        accx3.all_reduce(device)
        torch.cuda.synchronize()
        iidist.barrier()
        dx3 = time.perf_counter() - t1
        _, _, mioux3, _, _ = accx3.results()
        legx3 = {'value': round(world * B * args.steps / dx3, 3), 'unit': 'images/s',
                 'dtype': 'bf16 hi/lo pairs (16 significant bits per operand), three bf16 MFMA products '
                          'per term, f32 accumulate; FCN-8 on the f32 MFMA kernels',
                 'mode': 'bf16x3', 'in_flight': n_fly,
                 'ms_per_step': round(dx3 / args.steps * 1e3, 2),
                 'miou_iterative_inference': round(mioux3, 5),
                 'delta_miou_vs_f32': round(mioux3 - miou, 5),
                 'parity': 'fp32-CLASS, not fp32: per-layer relative RMS error 5e-6 (fp32 MFMA 4e-7, one '
                           'bf16 operand 3e-3); fixed criterion on the damped synthetic set, 10 steps '
                           'free-running, own masks: >= 0.998 of the pixels within 1e-4 of the float64 '
                           'path (the fp32 path is held to >= 0.999), at 224x224 (64 images: '
                           '`damped_set_64_images`) and at 360x480 (tests/test_gpu_x3.py; bit-exact vs '
                           'the oracle on 16-bit integer data)',
                 'damped_set_64_images': dict(prows.get('bf16x3 (DAE loop on hi/lo pairs)', {}), source=pfile)}
        if not args.no_roofline:
            # priced against a third of the bf16 peak: three MFMA products per algorithmic term
            legx3['roofline'] = conv_roofline(iix3, X, T, args.num_iter, args.step_size,
                                              dx3 / args.steps * 1e3, PEAK_TFLOPS_BF16_MFMA / 3)
            legx3['roofline']['peak_note'] = ('2500 / 3: dense bf16 MFMA peak over the three products '
                                              'of a split-operand term (FLOPs counted once)')
        if not args.no_two_streams:
            legx3['concurrent_streams'] = concurrent_leg('bf16x3')
        del iix3, poolx3
        torch.cuda.empty_cache()
        # the opt-in form with the FCN-8's 3x3 layers on pairs too (FCN8(mma='bf16x3')): faster, and
        # 0.993 instead of 0.999 of the damped set's pixels within 1e-4 (tests/test_gpu_x3.py)
        poolx3 = make_pool(n_fly, lambda: build_model(device, concat_h, mma='bf16x3', fcn_mma='bf16x3')[0],
                           Xs[0], Ts[0], args.num_iter, args.step_size)
        legx3['fcn_on_pairs_too'] = dict(
            leg(poolx3, Xs, Ts, args.steps, args.warmup),
            note="FCN8(mma='bf16x3') as well: opt-in, lower parity (>= 0.99 of the damped set's pixels "
                 'within 1e-4 of float64 asserted, tests/test_gpu_x3.py)')
        line['bf16x3'] = legx3
        del poolx3
        torch.cuda.empty_cache()
    if not args.no_early_stop and world == 1:
        # The loop with the reference's stop test ON (iterative_inference.py:265-277, eps 1e-3, num_iter 50) on the
        # DAMPED set, where the loop contracts: iteration histogram and images/s ("early-stop eps
        # 1e-3 active" next to the fixed-work mode).  profiles/r05_early_stop.md has the reading: on synthetic
        # images the per-image norms agree to 0.4 %, so sum(iters) = B x max(iters) and there is no active set
        # to compact.
        import collections
        from iterative_inference_segm_amd.api import IterativeInference
        from iterative_inference_segm_amd.dae import StandardDAE
        from iterative_inference_segm_amd.fcn8 import FCN8
        fpd, dpd, temp = S.make_damped_set()
        iid = IterativeInference(
            FCN8(fpd, N_CLASSES, layer=concat_h + ['probs_dimshuffle'], temperature=temp, device=device,
                 mma=args.bf16_mode),
            StandardDAE(dpd, N_CLASSES, device=device, mma=args.bf16_mode), N_CLASSES, [N_CLASSES], device=device)
        iid.prepare(B, 224, 224)
        es = {}
        for early in (True, False):
            o = iid.pred_fcn_fn(Xs[0])
            iid.refine(o[:-1], o[-1], args.step_size, 50, early_stop=early)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for r in range(2):
                o = iid.pred_fcn_fn(Xs[(r + 1) % n_distinct])
                _, its, nrm = iid.refine(o[:-1], o[-1], args.step_size, 50, early_stop=early)[:3]
            torch.cuda.synchronize()
            d = (time.perf_counter() - t1) / 2
            itl = its.cpu().tolist()
            es['on' if early else 'off'] = {'value': round(B / d, 1), 'ms': round(d * 1e3, 1),
                                            'iters': dict(sorted(collections.Counter(itl).items())),
                                            'sum_iters': sum(itl), 'B_max_iters': B * max(itl)}
        es['norm_spread'] = [round(float(nrm.min()), 6), round(float(nrm.max()), 6)]
        es['what'] = 'damped set, %s, num_iter 50, eps 1e-3' % args.bf16_mode
        line['early_stop'] = es
        del iid
        torch.cuda.empty_cache()
    if not args.no_configs and world == 1:
        # BASELINE configs[2], [3], [4] next to the headline (parity-test cases: tests/test_gpu_configs.py)
        torch.cuda.empty_cache()
        line['configs'] = other_configs(device, args.step_size, args.no_roofline, n_fly)
    if not args.no_strict_f64:
        # the float64 path (reference CPU numerics): same config at batch 32, the leg
        # that carries the end-to-end 1e-4 parity claim (tests/test_gpu_f64.py)
        del ii, pool
        built.clear()
        torch.cuda.empty_cache()
        b64 = min(B, 32)
        X64 = [x[:b64].to(torch.float64) for x in Xs[:2]]
        T64 = [t[:b64].to(torch.float64) for t in Ts[:2]]
        ii64 = make_pool(n_fly, lambda: build_model(device, concat_h, dtype=torch.float64)[0], X64[0], T64[0],
                         args.num_iter, args.step_size)
        f64 = leg(ii64, X64, T64, 4, 2)
        f64.update(dtype='f64', batch=b64, in_flight=n_fly,
                   note='float64 HIP kernels (v_mfma_f64_16x16x4_f64; Winograd F(2x2,3x3) on the wide '
                        '3x3 layers), identical loop and work eliminations; refined map within 1e-4 '
                        'of the float64 oracle end to end '
                        '(measured ~1e-11, tests/test_gpu_f64.py)')
        line['strict_f64'] = f64
        del ii64
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(fp, dp, args.num_iter, args.step_size, args.cpu_budget)
        detail = _write_detail(line)
        print(json.dumps(compact_line(line, detail)), flush=True)
    iidist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
