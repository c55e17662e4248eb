"""Data-parallel glue: one process per GPU, torch.distributed (backend 'nccl' = RCCL over xGMI on ROCm;
'gloo' in CPU tests).  The hot path has exactly ONE exchange step per gradient step: an all-reduce(sum) of the
flat [gradients | statistics] buffer; everything else shards with no communication."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or backend is not None) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # MPG_DIST_BACKEND=gloo: dry runs of a multi-rank launch on a box with fewer GPUs than ranks (RCCL refuses
            # two ranks on one device)
            backend = os.environ.get('MPG_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def backend():
    return dist.get_backend() if dist.is_initialized() else None


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def all_reduce_sum_(flat, force=False):
    """In-place sum over ranks of one flat float32 buffer (no-op on a single process unless `force`: a one-rank group
    still runs the collective - the way the RCCL path is exercised on a 1-GPU box)."""
    if dist.is_initialized() and (dist.get_world_size() > 1 or force):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(x):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return float(x)
