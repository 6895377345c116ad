"""CPU: host-side logic of the product -- metric reduction, result aggregation, checkpoint I/O,
synthetic data contract, sharding, and the world_size-2 all-reduce (gloo)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import metrics as ometrics
from iterative_inference_segm_amd import synthetic as S
from iterative_inference_segm_amd import weights
from iterative_inference_segm_amd.api import Metrics
from iterative_inference_segm_amd.dist import EvalAccumulator, shard_batches
from iterative_inference_segm_amd.helpers import results_line
from iterative_inference_segm_amd import dae as pdae, fcn8 as pfcn8
from oracle import dae as odae, fcn8 as ofcn8

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def confusion_np(y, t, C):
    p = y.argmax(1).ravel(); q = t.argmax(1).ravel()
    cm = np.zeros((C, C + 1))
    np.add.at(cm, (p, q), 1)
    tc = t[:, :C]
    sums = np.array([(((y - tc) ** 2).mean(1) * tc.sum(1)).sum(), tc.sum()])
    return cm, sums


def rand_batch(seed, B=2, C=11, H=12, W=10):
    rng = np.random.default_rng(seed)
    y = rng.random((B, C, H, W)); y /= y.sum(1, keepdims=True)
    return y, S.make_labels(B, H, W, n_classes=C, seed=seed, block=4, void_frac=0.2).astype(np.float64)


def test_metrics_reduce_host_matches_reference_formulas():
    y, t = rand_batch(0)
    cm, sums = confusion_np(y, t, 11)
    acc, jacc, mse = Metrics.reduce_host(cm, sums, 11)
    acc_r, jacc_r, mse_r = ometrics.val_fn(y, t, 11, [11])
    assert np.array_equal(jacc, jacc_r) and acc == pytest.approx(acc_r) and mse == pytest.approx(mse_r)


def test_eval_accumulator_is_means_of_batch_means_and_sum_then_divide_iou():
    acc = EvalAccumulator(11)
    rec_tot = acc_tot = 0.0
    jacc_tot = np.zeros((2, 11))
    for seed in range(3):
        y, t = rand_batch(seed)
        cm, sums = confusion_np(y, t, 11)
        a, j, m = Metrics.reduce_host(cm, sums, 11)
        acc.add_batch(cm, a, m)
        rec_tot += m; acc_tot += a; jacc_tot += j            # iterative_inference.py:288-290
    loss, a, miou, iou, nb = acc.results()
    loss_r, a_r, miou_r = results_line(rec_tot, acc_tot, jacc_tot, 3)      # helpers.py:172-177
    assert nb == 3 and loss == pytest.approx(loss_r) and a == pytest.approx(a_r)
    assert miou == pytest.approx(miou_r)


def test_shard_batches_partition():
    for n, w in [(10, 1), (10, 4), (3, 8), (64, 8)]:
        shards = [shard_batches(n, r, w) for r in range(w)]
        assert sorted(sum(shards, [])) == list(range(n))
        assert max(map(len, shards)) - min(map(len, shards)) <= 1


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from iterative_inference_segm_amd import dist as iidist
    r, w, dev = iidist.init_from_env('cpu')
    acc = iidist.EvalAccumulator(11)
    for b in iidist.shard_batches(5, r, w):                  # 5 batches over 2 ranks
        y, t = rand_batch(b)
        cm, sums = confusion_np(y, t, 11)
        a, _, m = Metrics.reduce_host(cm, sums, 11)
        acc.add_batch(cm, a, m)
    acc.all_reduce(dev)
    iidist.barrier()
    q.put((rank, acc.vec.copy()))
    torch.distributed.destroy_process_group()


def test_world_size_2_all_reduce_equals_single_process():
    """DP-sharded metrics == single-process metrics (gloo stands in for RCCL on CPU)."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    single = EvalAccumulator(11)
    for b in range(5):
        y, t = rand_batch(b)
        cm, sums = confusion_np(y, t, 11)
        a, _, m = Metrics.reduce_host(cm, sums, 11)
        single.add_batch(cm, a, m)
    assert np.allclose(got[0], got[1]) and np.allclose(got[0], single.vec)
    assert single.results()[4] == 5


def test_checkpoint_roundtrip_in_reference_arr_order(tmp_path):
    """np.savez(path, *get_all_param_values) layout: arr_0.. in P14 order, W then b."""
    fp = S.make_fcn8_params(width_div=32, fc_channels=8)
    path = str(tmp_path / 'fcn8_model.npz')
    weights.save_param_list(path, fp, pfcn8.PARAM_ORDER)
    with np.load(path) as f:
        assert len(f.files) == 42 and f['arr_0'].shape == fp['conv1_1'][0].shape
        assert f['arr_41'].shape == fp['upsample'][1].shape
    back = weights.load_param_list(path, pfcn8.PARAM_ORDER)
    assert all(np.array_equal(back[k][0], fp[k][0]) for k in fp)
    dp = S.make_dae_params(n_filters=2, h_channels=(4,))
    order = pdae.param_order()
    assert order == odae.param_order() and pfcn8.PARAM_ORDER == ofcn8.PARAM_ORDER
    weights.save_param_list(str(tmp_path / 'dae.npz'), dp, order)
    with np.load(str(tmp_path / 'dae.npz')) as f:
        assert len(f.files) == 24
    with pytest.raises(ValueError, match='expected'):
        weights.load_param_list(str(tmp_path / 'dae.npz'), order[:-1])
    # bn=1: every conv is followed by beta, gamma, mean, inv_std (fcn_down.py:112-114)
    dpb = S.make_dae_params(n_filters=2, h_channels=(4,), bn=1)
    ob = pdae.param_order(bn=1)
    assert ob == odae.param_order(bn=1) and ob[1] == 'conv1_1_bn' and ob[-1] == 'up_conv1_bn'
    weights.save_param_list(str(tmp_path / 'dae_bn.npz'), dpb, ob)
    with np.load(str(tmp_path / 'dae_bn.npz')) as f:
        assert len(f.files) == 24 + 12 * 4
    back = weights.load_param_list(str(tmp_path / 'dae_bn.npz'), ob)
    assert len(back['conv3_1_bn']) == 4 and np.array_equal(back['conv3_1_bn'][1], dpb['conv3_1_bn'][1])


def test_synthetic_data_contract():
    x = S.make_images(2, 20, 24)
    assert x.dtype == np.float32 and x.shape == (2, 3, 20, 24) and 0 <= x.min() and x.max() < 1
    t = S.make_labels(2, 20, 24, block=4, void_frac=0.2)
    assert t.shape == (2, 12, 20, 24) and np.all(t.sum(1) == 1)           # one-hot, void last
    assert 0 < t[:, 11].mean() < 0.5
    assert np.array_equal(S.make_images(2, 8, 8, seed=5), S.make_images(2, 8, 8, seed=5))
    k = S.bilinear_kernel(4)
    assert np.allclose(k, k.T) and np.allclose(k[0], [0.0625, 0.1875, 0.1875, 0.0625])


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    return port


@pytest.mark.parametrize('world', [8])
def test_bench_dry_run_world_size_8(world):
    """The N=8 launch line of the driver, rehearsed on CPU (gloo): 8 ranks rendezvous on 127.0.0.1,
    barrier-bracketed timing with max over ranks, the one metric all-reduce, ONE JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                        '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
                        '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', str(world),
                        '--dry-run', '--steps', '2', '--warmup', '1'], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == world and out['batches_reduced'] == 2 * world


def test_entry_point_dry_run_world_size_8(tmp_path):
    """iterative_inference.py under 8 ranks (gloo, --dry_run): 20 reference batches shard 3/3/3/3/2/2/2/2,
    one all-reduce puts all 20 into rank 0's summary."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'iterative_inference.py'), '--synthetic', '--dry_run', '--n_images', '40',
           '--batch_size', '2', '--image_size', '32', '40', '-step', '0.1', '--num_iter', '3',
           '-dae_dict', '{"kind": "standard", "concat_h": ["pool4"], "additional_pool": 2, '
                        '"unpool_type": "trackind", "skip": true}',
           '--savepath', str(tmp_path / 'save'), '--loadpath', str(tmp_path / 'load')]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count('>>>>> ITERATIVE INFERENCE:') == 1
    assert 'DRY RUN: 8 ranks, 20 batches reduced, rank 0 owned [0, 8, 16]' in r.stdout


def test_rccl_init_failure_exits_nonzero_with_the_error_text():
    """A rank whose process group cannot come up (here: the RCCL backend on a box without a GPU for
    it) must exit non-zero with the error on stderr -- no retry, no re-exec."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(WORLD_SIZE='2', RANK='0', LOCAL_RANK='7', MASTER_ADDR='127.0.0.1', MASTER_PORT='29695',
               IISEG_DIST_BACKEND='nccl')
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch\n"
            "from iterative_inference_segm_amd import dist as d\n"
            "torch.cuda.set_device = lambda *_: None\n"
            "d.init_from_env('cuda')\n" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and 'init failed on rank 0 / 2' in r.stderr
    import os as _os
    assert _os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0') == '0'


def test_bench_self_launches_two_ranks_dry_run():
    """`python bench.py --gpus 2` outside a torch.distributed environment must fan out to two
    workers by itself (the parent never touches the GPU), reduce over both ranks and print ONE
    JSON line from rank 0.  --dry-run swaps the HIP work for the protocol alone (gloo on CPU)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run',
                        '--steps', '3', '--warmup', '1'], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['dry_run'] is True
    assert out['batches_reduced'] == 6 and out['scaling'] == 'weak'
    # a failing worker makes the launcher fail (WORLD_SIZE mismatch inside the workers)
    bad = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                          '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
                          '29677', os.path.join(ROOT, 'bench.py'), '--gpus', '3', '--dry-run'],
                         env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0


def test_entry_point_two_ranks_dry_run(tmp_path):
    """`iterative_inference.py --synthetic` under torch.distributed.run with two ranks (gloo, CPU,
    --dry_run: everything but the HIP work): flags parsed, rendezvous on 127.0.0.1, rank 0 alone
    writes config.txt and prints, the 5 reference batches are sharded 3 + 2, ONE all-reduce puts all
    5 into the summary."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', '29683',
           os.path.join(ROOT, 'iterative_inference.py'), '--synthetic', '--dry_run', '--n_images', '10',
           '--batch_size', '2', '--image_size', '32', '40', '-step', '0.1', '--num_iter', '3',
           '-dae_dict', '{"kind": "standard", "concat_h": ["pool4"], "additional_pool": 2, '
                        '"unpool_type": "trackind", "skip": true}',
           '--savepath', str(tmp_path / 'save'), '--loadpath', str(tmp_path / 'load')]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count('>>>>> ITERATIVE INFERENCE:') == 1            # rank 0 only
    assert 'DRY RUN: 2 ranks, 5 batches reduced, rank 0 owned [0, 2, 4]' in r.stdout
    assert 'Jaccard: 1.0' in r.stdout
    cfgs = list((tmp_path / 'save').rglob('config.txt'))
    assert len(cfgs) == 1


def test_nonfinite_weights_are_refused_when_a_layer_is_built():
    """The conv kernels are compiled with relaxed NaN handling (build.py EXTRA_FLAGS): a NaN is not promised to
    propagate (the reference's Theano `rectify` would hand it through, models/fcn_down.py:102-104).  The
    product's promise instead: non-finite parameters never get in -- ops.Conv raises when it is built."""
    import numpy as np
    import pytest
    from iterative_inference_segm_amd import build, ops
    build.build()
    W = np.ones((4, 3, 3, 3), np.float32)
    b = np.zeros(4, np.float32)
    ops.Conv(W, b, pad=1, relu=True, device='cpu')
    for bad in (np.nan, np.inf, -np.inf):
        Wb = W.copy(); Wb[1, 2, 0, 1] = bad
        with pytest.raises(ValueError, match='non-finite'):
            ops.Conv(Wb, b, pad=1, relu=True, device='cpu')
        bb = b.copy(); bb[3] = bad
        with pytest.raises(ValueError, match='non-finite'):
            ops.Conv(W, bb, pad=1, relu=True, device='cpu')


def test_engine_pool_needs_an_engine_and_one_engine_is_the_plain_path():
    """api.EnginePool on the host: no engine is an error; a pool of one hands out that engine on the caller's
    stream (no stream objects are made, so it also runs where there is no GPU)."""
    from iterative_inference_segm_amd.api import EnginePool
    with pytest.raises(ValueError):
        EnginePool([])
    e = object()
    pool = EnginePool([e])
    assert len(pool) == 1 and pool.streams == [None]
    for _ in range(3):
        with pool.lane() as got:
            assert got is e
    pool.join()
    pool.synchronize()


def test_bench_line_is_short_and_ends_with_the_16_bit_leg():
    """bench.compact_line on the detail record of a real run (profiles/r05_bench_n1_detail.json): the ONE line rank
    0 prints carries the contract's keys first, stays under 5 KB, and ends with strict_f64, cpu_baseline and the
    bf16 leg -- its roofline fraction, the fraction over every 3x3 launch, the batches in flight and the value
    with one batch in flight -- where a truncated log tail keeps them."""
    import json
    import os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    detail = json.load(open(os.path.join(root, 'profiles', 'r05_bench_n1_detail.json')))
    out = bench.compact_line(detail, 'gpurun_out/bench_detail.json')
    text = json.dumps(out)
    assert len(text) < 5000
    keys = list(out)
    assert keys[:13] == ['metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                         'scaling', 'vs_baseline', 'dtype', 'data', 'config']
    assert keys[-3:] == ['strict_f64', 'cpu_baseline', 'bf16']
    assert out['value'] == detail['value'] and out['in_flight'] == detail['in_flight']['engines']
    assert out['one_in_flight'] == detail['in_flight']['one_in_flight']['value']
    b = out['bf16']
    assert b['in_flight'] == 2 and b['one_in_flight'] and 0.3 < b['roofline']['frac'] < 0.6
    assert 0.3 < b['roofline']['all_3x3_frac'] <= b['roofline']['frac']
    assert out['roofline']['bound'] == 'mfma' and out['cpu_baseline']['kind'] == 'port'
    tail = text[-1200:]
    assert '"all_3x3_frac"' in tail and '"one_in_flight"' in tail
