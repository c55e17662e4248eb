"""Data-parallel glue: one process per GPU, torch.distributed (backend 'nccl' = RCCL over xGMI on ROCm;
'gloo' in CPU tests).  The hot path has exactly ONE exchange step per gradient step: an all-reduce(sum) of the
flat [gradients | statistics] buffer; everything else shards with no communication."""
import os

import torch
import torch.distributed as dist


_oneshot = None          # OneShotAllReduce of this process (MPG_DIST_BACKEND=oneshot), built at the first exchange
_exchange = 'collective'  # 'collective': dist.all_reduce of the process group's backend; 'oneshot': the IPC one-shot form below


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank).
    MPG_DIST_BACKEND (or `backend`): 'nccl' (= RCCL, the default on GPUs), 'gloo' (dry runs of a multi-rank launch on a box
    with fewer GPUs than ranks: RCCL refuses two ranks on one device), or 'oneshot' - the gradient exchange runs as the
    one-shot all-reduce over IPC-mapped staging slots (OneShotAllReduce); the process group itself (rendezvous, handle
    exchange, host barrier, the bench's max-over-ranks) is then a gloo group."""
    global _exchange
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or backend is not None) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('MPG_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'oneshot':
            _exchange, backend = 'oneshot', 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class OneShotAllReduce(object):
    """One-shot all-reduce of a flat float32 buffer for the ranks of ONE node.

    Every rank owns a staging array [2 parities][world slots][n] in device memory and maps every peer's through HIP IPC
    (torch's CUDA-IPC reductions: hipIpcGetMemHandle / hipIpcOpenMemHandle; HSA_ENABLE_IPC_MODE_LEGACY=0 on this pool).  An
    exchange is
        1. rank r copies its buffer into slot r of EVERY rank's staging array (world device-to-device copies; between GPUs
           these are direct xGMI writes - one hop, all 7 links busy at once, no ring);
        2. a cross-process barrier: each rank waits for its own copies (stream synchronize) and then for everybody (host
           barrier of the process group).  No device-side spin-wait between processes: on the test box two ranks TIME-SHARE
           one GPU, and a kernel polling for a peer that is not scheduled would never return;
        3. every rank sums the `world` slots of its own array in rank order 0, 1, 2, ... (mpg_sum_slots): every replica computes
           the SAME association of the same numbers, so the replicas stay bit-identical by construction.
    Two staging parities alternate by call: a rank can run at most one barrier ahead of the slowest one, so nobody overwrites
    a slot that is still being summed.  What this form is for: the gradient message (821 KB) is latency-bound; a ring
    all-reduce pays 2 (world - 1) dependent hops, this pays one write + one barrier + one local sum.  Measured here only
    for correctness (two processes on one GPU); no multi-GPU number is claimed."""

    def __init__(self, n, device):
        from torch.multiprocessing.reductions import reduce_tensor
        from . import _lib as L
        self.L = L
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.n = int(n)
        self.stage = torch.zeros(2, self.world, self.n, dtype=torch.float32, device=device)
        torch.cuda.synchronize()
        fn, args = reduce_tensor(self.stage)
        handles = [None] * self.world
        dist.all_gather_object(handles, (fn, args))
        self.peers = []
        for r in range(self.world):
            if r == self.rank:
                self.peers.append(self.stage)
            else:
                f, a = handles[r]
                self.peers.append(f(*a))           # rebuild_cuda_tensor: opens the peer's allocation
        self.calls = 0
        dist.barrier()

    def all_reduce_sum_(self, flat):
        assert flat.numel() == self.n and flat.dtype == torch.float32 and flat.is_contiguous()
        par = self.calls & 1
        self.calls += 1
        for r in range(self.world):                 # 1. my buffer into slot `rank` of every rank's array
            self.peers[r][par, self.rank].copy_(flat, non_blocking=True)
        torch.cuda.current_stream().synchronize()   # 2. my writes have landed ...
        dist.barrier()                              #    ... and so have everybody else's
        L = self.L                                  # 3. fixed-order local sum
        L.call('mpg_sum_slots', L.ptr(self.stage[par]), L.c_int(self.world), L.c_int(self.n), L.ptr(flat), L.stream())
        return flat


def backend():
    if not dist.is_initialized():
        return None
    return 'oneshot (IPC staging slots; process group: %s)' % dist.get_backend() if _exchange == 'oneshot' else dist.get_backend()


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def all_reduce_sum_(flat, force=False):
    """In-place sum over ranks of one flat float32 buffer (no-op on a single process unless `force`: a one-rank group
    still runs the collective - the way the RCCL path is exercised on a 1-GPU box)."""
    global _oneshot
    if dist.is_initialized() and (dist.get_world_size() > 1 or force):
        if _exchange == 'oneshot':
            if _oneshot is None or _oneshot.n != flat.numel():
                _oneshot = OneShotAllReduce(flat.numel(), flat.device)
            _oneshot.all_reduce_sum_(flat)
        else:
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
