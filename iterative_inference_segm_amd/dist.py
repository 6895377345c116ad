"""Data-parallel evaluation over the GPUs of one node (absent from the reference, which is a
single process).  Images are independent, so they shard over ranks with NO
data-path collective; the only exchange is one all-reduce (RCCL over xGMI; gloo on CPU) of the
metric accumulator at the end of an evaluation:

    [ confusion counts C*(C+1) | sum of per-batch acc | sum of per-batch mse | n_batches ]

as one float64 vector (counts stay exact below 2^53).  IoU is sum-then-divide and therefore
shard-invariant; acc / mse stay "means of per-batch means" (reference helpers.py:175-176) as long
as every reference batch lives on exactly one rank, which `shard_batches` guarantees.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(device_type=None):
    """One process per GPU; reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world, device)."""
    # dmabuf IPC only on this pool: RCCL's / torch's cross-process buffer sharing fails with
    # `hipIpcGetMemHandle: invalid argument` without it.  Set here, before the first HIP call of the
    # process, so that a launch line the caller wrote (torch.distributed.run directly) gets it too.
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if device_type is None:
        device_type = 'cuda' if torch.cuda.is_available() else 'cpu'
    if device_type == 'cuda':
        # IISEG_FORCE_DEVICE: rehearse N ranks on one GPU (with IISEG_DIST_BACKEND=gloo; RCCL
        # itself needs one device per rank)
        local = int(os.environ.get('IISEG_FORCE_DEVICE', local))
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)
    else:
        device = torch.device('cpu')
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = os.environ.get('IISEG_DIST_BACKEND',
                                 'nccl' if device_type == 'cuda' else 'gloo')  # 'nccl' = RCCL on ROCm
        try:
            if backend == 'nccl':
                if torch.cuda.device_count() <= local:
                    raise RuntimeError('rank %d wants device %d, %d visible (RCCL needs one device '
                                       'per rank)' % (rank, local, torch.cuda.device_count()))
                dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                        device_id=device)
            else:
                dist.init_process_group(backend=backend, rank=rank, world_size=world)
        except Exception as e:       # never retried, never re-exec'd: the launcher sees a failed rank
            import sys
            sys.stderr.write('iiseg.dist: %s init failed on rank %d / %d: %r\n'
                             % (backend, rank, world, e))
            sys.stderr.flush()
            raise SystemExit(3)
        _STATE['device'] = device if backend == 'nccl' else None
    return rank, world, device


def shard_batches(n_batches, rank, world):
    """Batch indices owned by `rank`: round-robin so every rank gets floor/ceil(n/world)."""
    return list(range(rank, n_batches, world))


_STATE = {'device': None}      # the rank's GPU when the backend is RCCL


def barrier():
    if dist.is_available() and dist.is_initialized():
        dev = _STATE['device']
        if dev is not None:
            # RCCL: name the device, or the barrier's hidden all-reduce picks "the current one" by a
            # heuristic (and warns); gloo takes no device_ids
            dist.barrier(device_ids=[dev.index])
        else:
            dist.barrier()


class EvalAccumulator:
    """Running totals of one evaluation (the rec_tot/acc_tot/jacc_tot of
    iterative_inference.py:216-224,288-290), reducible across ranks."""

    def __init__(self, n_classes):
        self.C = n_classes
        self.vec = np.zeros(n_classes * (n_classes + 1) + 3, dtype=np.float64)

    def add_batch(self, cm, acc, mse):
        n = self.C * (self.C + 1)
        self.vec[:n] += np.asarray(cm, dtype=np.float64).reshape(-1)
        self.vec[n] += acc
        self.vec[n + 1] += mse
        self.vec[n + 2] += 1

    def all_reduce(self, device):
        """Sum over ranks (no-op for a single process).  One latency-bound collective."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.from_numpy(self.vec).to(device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            self.vec = t.cpu().numpy()
        return self

    def results(self):
        """(loss, acc, mean IoU, per-class IoU, n_batches) as print_results reports them."""
        C = self.C
        n = C * (C + 1)
        cm = self.vec[:n].reshape(C, C + 1)[:, :C]
        tp = np.diag(cm)
        denom = cm.sum(1) + cm.sum(0) - tp
        with np.errstate(invalid='ignore', divide='ignore'):
            iou = tp / denom
            miou = float(np.nanmean(iou))
        nb = max(self.vec[n + 2], 1.0)
        return self.vec[n + 1] / nb, self.vec[n] / nb, miou, iou, int(self.vec[n + 2])
