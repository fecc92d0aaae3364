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


def _active(dist=None):
    """torch.distributed if a process group of more than one rank is up, else None.

    ``CAMAL_COLLECTIVES_AT_WORLD1=1`` (tests): a process group of ONE rank counts as well, so that every collective
    of the data-parallel path (gradient all-reduce, broadcasts of rank 0's buffers / decisions / shuffle seed, the
    per-epoch sums) really goes through the backend — RCCL on a one-GPU box — with results that must equal the
    no-collective run bit for bit."""
    if dist is None:
        import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return None
    if dist.get_world_size() == 1 and os.environ.get("CAMAL_COLLECTIVES_AT_WORLD1", "0") != "1":
        return None
    return dist


def collectives_on(dist=None) -> bool:
    """True when the data-parallel collectives are to be issued (world > 1, or the world-1 test switch)."""
    return _active(dist) is not None


def control_device(d, device=None):
    """Device of the small control-path tensors handed to a collective: the host under "gloo"; under RCCL ("nccl")
    a GPU — ``device`` if given, else the current one (a CPU tensor has no backend there and raises)."""
    if d.get_backend() == "gloo":
        return torch.device("cpu")
    if device is not None:
        return torch.device(device)
    return torch.device("cuda", torch.cuda.current_device())


def rank_world(dist=None):
    d = _active(dist)
    return (d.get_rank(), d.get_world_size()) if d is not None else (0, 1)


def _gloo_on_gpu(flat, dist):
    """CPU-test rehearsals run the "gloo" backend on device tensors (one GPU, several ranks): stage through the
    host there.  RCCL ("nccl") takes the device tensor as it is."""
    return flat.is_cuda and dist.get_backend() == "gloo"


def batch_shard(n: int, dist=None):
    """[lo, hi) of a step's n samples that this rank trains on: chunks of ceil(n / world) in rank order, the way
    nn.DataParallel's scatter cuts a batch (torch_trainer.py:82-90); the last ranks of a short batch get fewer
    samples or none."""
    r, w = rank_world(dist)
    chunk = -(-n // w)
    lo = min(n, r * chunk)
    return lo, min(n, lo + chunk)


def allreduce_mean_(flat: torch.Tensor, dist=None):
    """In-place mean over ranks of a flat bucket (one collective)."""
    d = _active(dist)
    if d is None:
        return flat
    allreduce_sum_(flat, d)
    flat.div_(d.get_world_size())
    return flat


def allreduce_sum_(flat: torch.Tensor, dist=None):
    """In-place sum over ranks of the flat gradient bucket: ONE collective per step.  The ranks' gradients arrive
    already weighted by their share of the batch (camal_trainer_set_grad_scale), so the sum is the gradient of the
    global mean loss and nothing is divided afterwards."""
    d = _active(dist)
    if d is None:
        return flat
    if _gloo_on_gpu(flat, d):
        h = flat.cpu()
        d.all_reduce(h, op=d.ReduceOp.SUM)
        flat.copy_(h)
    else:
        d.all_reduce(flat, op=d.ReduceOp.SUM)
    return flat


def broadcast_(t: torch.Tensor, src=0, dist=None):
    """In-place broadcast of rank src's tensor (initial weights, rank 0's BatchNorm running statistics)."""
    d = _active(dist)
    if d is None:
        return t
    if _gloo_on_gpu(t, d):
        h = t.cpu()
        d.broadcast(h, src)
        t.copy_(h)
    else:
        d.broadcast(t, src)
    return t


def broadcast_scalar(v: float, src=0, dist=None, device=None) -> float:
    """Rank src's value on every rank (control path: the validation loss the scheduler and the early stopper
    decide on must be the same number everywhere, or the ranks' collectives stop matching)."""
    d = _active(dist)
    if d is None:
        return v
    t = torch.tensor([float(v)], dtype=torch.float64, device=control_device(d, device))
    d.broadcast(t, src)
    return float(t.item())


def broadcast_scalar_int(v: int, src=0, dist=None, device=None) -> int:
    """Rank src's integer on every rank (the shuffle seed of an epoch)."""
    d = _active(dist)
    if d is None:
        return v
    t = torch.tensor([int(v)], dtype=torch.int64, device=control_device(d, device))
    d.broadcast(t, src)
    return int(t.item())


def sum_over_ranks(a: np.ndarray, dist=None, device=None) -> np.ndarray:
    """Element-wise sum over ranks of a small host array (per-epoch loss / accuracy bookkeeping)."""
    d = _active(dist)
    if d is None:
        return a
    t = torch.as_tensor(np.asarray(a, np.float64)).to(control_device(d, device))
    d.all_reduce(t, op=d.ReduceOp.SUM)
    return t.cpu().numpy()


class PendingMean:
    """An in-flight all-reduce of a flat bucket: ``wait()`` makes the current stream wait for the collective (the
    host does not block on "nccl") and, for the mean variant, finishes the division."""

    def __init__(self, flat, work, world, staged=None):
        self.flat, self.work, self.world, self.staged = flat, work, world, staged

    def wait(self):
        if self.work is not None:
            self.work.wait()
            if self.staged is not None:
                self.flat.copy_(self.staged)
                self.staged = None
            if self.world != 1:
                self.flat.div_(self.world)
            self.work = None
        return self.flat


def _start(flat, dist, divide):
    d = _active(dist)
    if d is None:
        return PendingMean(flat, None, 1)
    world = d.get_world_size() if divide else 1
    if _gloo_on_gpu(flat, d):
        h = flat.cpu()
        return PendingMean(flat, d.all_reduce(h, op=d.ReduceOp.SUM, async_op=True), world, staged=h)
    return PendingMean(flat, d.all_reduce(flat, op=d.ReduceOp.SUM, async_op=True), world)


def allreduce_mean_start(flat: torch.Tensor, dist=None) -> PendingMean:
    """Start the mean over ranks of a flat bucket without waiting for it: the collective runs on the backend's own
    stream behind the work already queued on the current one, so the next member's forward and backward overlap it
    (xGMI all-reduces of 2-4 MB are latency-bound).  Nothing may touch ``flat`` before ``wait()``."""
    return _start(flat, dist, True)


def allreduce_sum_start(flat: torch.Tensor, dist=None) -> PendingMean:
    """The same for the sum (gradient buckets weighted by the ranks' batch shares: nothing to divide)."""
    return _start(flat, dist, False)


def max_over_ranks(value: float, device=None, dist=None) -> float:
    if dist is None:
        import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
