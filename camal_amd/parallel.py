"""Multi-GPU layer: one process per GPU (``torch.distributed``; backend "nccl" = RCCL on ROCm,
"gloo" in CPU tests).

Inference: windows are independent units — ``[N,L]`` is cut into ``world``
contiguous row blocks (``N // world`` each, remainder to the last), every rank runs the whole
ensemble on its block, NO collective touches the data path.

Training: data-parallel per member like the reference's ``nn.DataParallel``
(``torch_trainer.py:82-90``): per-rank BatchNorm batch statistics, and ONE all-reduce (sum,
then divide by world) of a single flat fp32 gradient bucket per step (2-4 MB: latency-bound on
xGMI, so one fused call instead of per-parameter buckets).
"""
from __future__ import annotations

import os

import numpy as np
import torch


def world_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_bounds(n: int, world: int, rank: int):
    """[start, end) of rank's contiguous block of n windows."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per = n // world
    start = rank * per
    end = n if rank == world - 1 else start + per
    return start, end


def predict_sharded(camal, data, rank=None, world=None, **kw):
    """This rank's slice of ``camal.predict(data)``: returns (start, end, result) where result is
    what ``predict`` returns for rows [start, end).  No communication."""
    r, _, w = world_info()
    rank = r if rank is None else rank
    world = w if world is None else world
    data = np.asarray(data)
    if data.ndim == 1:
        data = data[None, :]
    s, e = shard_bounds(len(data), world, rank)
    if e == s:
        return s, e, None
    return s, e, camal.predict(data[s:e], **kw)


def gather_shards(parts, n, dist=None):
    """Optional convenience for small outputs: all-gather per-rank NumPy blocks (object gather;
    control path only — large results should stay where they were produced)."""
    if dist is None or not dist.is_initialized():
        return parts
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, parts)
    return out


def member_owner(idx: int, world: int) -> int:
    """Rank that trains candidate ``idx`` (kernel size x try, in the reference's loop order core.py:104-105) when
    the members are spread over the ranks: round robin, so 5 kernels on 8 GPUs take 5 ranks, on 2 GPUs 3 + 2."""
    return idx % max(1, world)


def merge_member_logs(local: dict, dist=None) -> dict:
    """{candidate name: (test loss, training log)} of every rank, identical on every rank afterwards.  Control path
    (object all-gather once per ``CamAL.train``): the members' training itself exchanges nothing."""
    if dist is None:
        import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(local)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local)
    merged = {}
    for p in parts:
        for k, v in p.items():
            if k in merged:
                raise RuntimeError(f"candidate {k} was trained on two ranks")
            merged[k] = v
    return merged


def allreduce_mean_(flat: torch.Tensor, dist=None):
    """In-place mean over ranks of the flat gradient bucket (one collective per step)."""
    if dist is None:
        import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    return flat


class PendingMean:
    """An in-flight ``allreduce_mean_start``: ``wait()`` makes the current stream wait for the collective (the host
    does not block on "nccl") and finishes the mean."""

    def __init__(self, flat, work, world):
        self.flat, self.work, self.world = flat, work, world

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.flat.div_(self.world)
            self.work = None
        return self.flat


def allreduce_mean_start(flat: torch.Tensor, dist=None) -> PendingMean:
    """Start the mean over ranks of a flat gradient bucket without waiting for it: the collective runs on the
    backend's own stream behind the work already queued on the current one, so the next member's forward and backward
    overlap it (xGMI all-reduces of 2-4 MB are latency-bound).  Nothing may touch ``flat`` before ``wait()``."""
    if dist is None:
        import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return PendingMean(flat, None, 1)
    return PendingMean(flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), dist.get_world_size())


def max_over_ranks(value: float, device=None, dist=None) -> float:
    if dist is None:
        import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
