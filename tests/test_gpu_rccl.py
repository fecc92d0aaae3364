"""GPU: RCCL (torch.distributed backend "nccl" on ROCm) runs on this library's buffers.  One GPU = one rank, so this is
a smoke test of the communicator and of the collectives the data-parallel step issues (all-reduce of the flat gradient
bucket, started asynchronously on a side stream and waited for before AdamW; broadcast of rank 0's buffers) with a
world of ONE: the arithmetic of N > 1 is covered by tests/test_gpu_dataparallel.py (gloo, two processes) and the
RCCL ring itself only by the driver's multi-GPU bench.  Runs in a child process: a process group does not leak into
the other tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import os, sys
sys.path.insert(0, os.environ["CAMAL_ROOT"])
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["CAMAL_PORT"], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
from camal_amd import parallel, synth
from camal_amd.resnet import CamALResNet
from camal_amd.trainer import MemberStreams, TrainEngine
x, y = synth.ideal_windows(16, 96, seed=1)
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
torch.manual_seed(0)
engines = [TrainEngine(CamALResNet(kernel_size=k), "cuda:0", 16, 96) for k in (5, 9)]
ref = []
for e in engines:
    e.forward_backward(xd, yd)
    ref.append(e.grads.clone())
# the collectives of the data-parallel step, through RCCL, on the flat buckets
for e, g in zip(engines, ref):
    w = dist.all_reduce(e.grads, op=dist.ReduceOp.SUM, async_op=True)   # what allreduce_sum_start issues for world > 1
    w.wait()
    dist.broadcast(e.params, 0)
    assert torch.equal(e.grads, g)                                      # a world of one: the sum is the bucket itself
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
# and the members' step on their own streams with the process group up (world 1: no collective is issued)
ms = MemberStreams(engines)
ms.step(xd, yd, 1e-3, 0.0)
torch.cuda.synchronize()
assert parallel.rank_world() == (0, 1) and all(np.isfinite(float(e.loss.item())) for e in engines)
dist.destroy_process_group()
print("RCCL_OK", dist.is_nccl_available())
"""


def test_rccl_collectives_on_the_gradient_buckets():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CAMAL_ROOT=root, CAMAL_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=170)
    assert p.returncode == 0 and "RCCL_OK True" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
