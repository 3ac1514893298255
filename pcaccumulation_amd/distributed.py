"""Data-parallel glue: one process per GPU, scenes sharded by rank, ONE flat-buffer gradient all-reduce per
optimizer step over RCCL/xGMI (backend "nccl" on ROCm; "gloo" for the CPU tests).

The reference is single-GPU (its only collectives sit in dead code, models/utils/norm.py:8-20), so nothing is
translated here.  Why a flat pre-zeroed buffer instead of per-parameter hooks: MotionNet has data-dependent
branches (models/motionnet.py:222,243) that leave whole sub-modules without gradients on some ranks; a flat
buffer all-reduces the same 11.1 M elements (44.5 MB fp32) on every rank regardless.  The training loop of
the reference swallows exceptions per iteration (libs/trainer.py:234-235); `all_ok` lets every rank agree to
skip a step instead of deadlocking in the collective.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them).
    Returns (rank, world_size, local_rank).  No-op for a single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


class FlatGradAllReduce(object):
    """Averages the gradients of `params` across ranks through one contiguous buffer."""

    def __init__(self, params, dtype=torch.float32):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel, dtype=dtype, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def __call__(self):
        ws = world_size()
        if ws == 1:
            return
        self.flat.zero_()
        have = [(p, v) for p, v in zip(self.params, self.views) if p.grad is not None]
        if have:                                          # two multi-tensor launches instead of 2 x 195 small copies
            torch._foreach_copy_([v for _, v in have], [p.grad for p, _ in have])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(ws)
        if have:
            torch._foreach_copy_([p.grad for p, _ in have], [v for _, v in have])
        for p, v in zip(self.params, self.views):
            if p.grad is None:                            # branch not taken on this rank: adopt the other ranks' mean
                p.grad = v.clone()


def all_ok(ok, device):
    """True iff every rank reports ok (1-element MIN all-reduce)."""
    if world_size() == 1:
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def max_over_ranks(value, device):
    if world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()
