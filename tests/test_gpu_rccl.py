"""GPU: the data-parallel path's collectives run through RCCL (torch.distributed backend "nccl" on ROCm) on this
library's buffers.  One GPU = one rank, so the process group has a world of ONE; ``CAMAL_COLLECTIVES_AT_WORLD1=1``
(camal_amd/parallel.py) makes the library issue every collective of the N > 1 path anyway — the asynchronous gradient
all-reduce of TrainEngine.start_grad_allreduce -> optimizer_step(pending=...), MemberStreams.step on per-member
streams, sync_from_rank0's broadcasts, the control-path scalars (validation loss, shuffle seed, per-epoch sums) and two
epochs of BasedClassifTrainer.train — and with one rank every result must equal the run WITHOUT collectives bit for bit.
The arithmetic of N > 1 is covered by tests/test_gpu_dataparallel.py (gloo, two processes); what this adds is that the
exact calls the driver's 8-GPU run will make have executed on device tensors under "nccl" (torch_trainer.py:82-90 is
what they replace).  bench.py's own launcher path (init_process_group("nccl", device_id=...), barrier, MAX over ranks)
runs the same way with RANK=0 WORLD_SIZE=1.  Child processes: a process group does not leak into the other tests."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import os, sys, warnings
sys.path.insert(0, os.environ["CAMAL_ROOT"])
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["CAMAL_PORT"], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
from camal_amd import parallel, synth
from camal_amd.resnet import CamALResNet
from camal_amd.trainer import BasedClassifTrainer, MemberStreams, TrainEngine, TSLoader

SWITCH = "CAMAL_COLLECTIVES_AT_WORLD1"
calls = {"all_reduce": 0, "broadcast": 0}
_ar, _bc = dist.all_reduce, dist.broadcast
def counted_all_reduce(t, *a, **k):
    assert t.is_cuda, "a CPU tensor was handed to an RCCL collective"
    calls["all_reduce"] += 1
    return _ar(t, *a, **k)
def counted_broadcast(t, *a, **k):
    assert t.is_cuda, "a CPU tensor was handed to an RCCL collective"
    calls["broadcast"] += 1
    return _bc(t, *a, **k)
dist.all_reduce, dist.broadcast = counted_all_reduce, counted_broadcast

x, y = synth.ideal_windows(16, 96, seed=1)
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()

def engines():
    torch.manual_seed(0)
    return [TrainEngine(CamALResNet(kernel_size=k), "cuda:0", 16, 96) for k in (5, 9)]

def run_steps(on):
    os.environ[SWITCH] = "1" if on else "0"
    assert parallel.collectives_on() == on and parallel.rank_world() == (0, 1)
    out = []
    es = engines()  # TrainEngine.load_from_model -> sync_from_rank0: 4 broadcasts per engine when on
    assert all(e._dp() == on for e in es)
    for e in es:    # the trainer's order: forward/backward, all-reduce inside optimizer_step
        e.forward_backward(xd, yd, global_batch=16)
        e.optimizer_step(1e-3, 1e-2)
    for e in es:    # bench.py's order: start the all-reduce, wait for it just before AdamW
        e.forward_backward(xd, yd, global_batch=16)
    pend = [e.start_grad_allreduce() for e in es]
    for e, p in zip(es, pend):
        e.optimizer_step(1e-3, 1e-2, pending=p)
    ms = MemberStreams(es)  # every member on its own stream, collectives issued in member order
    for _ in range(2):
        ms.step(xd, yd, 1e-3, 1e-2, global_batch=16)
    torch.cuda.synchronize()
    for e in es:
        out += [e.params.clone(), e.grads.clone(), e.m.clone(), e.v.clone(), e.running.clone(), e.loss.clone()]
        e.close()
    return out

base = run_steps(False)
assert calls == {"all_reduce": 0, "broadcast": 0}
got = run_steps(True)
assert calls["all_reduce"] == 2 * 4 and calls["broadcast"] == 2 * 4, calls
for a, b in zip(base, got):
    assert torch.equal(a, b)  # a world of one: sums and broadcasts change nothing, bit for bit

# control path: rank 0's decision value, shuffle seed, per-epoch sums, as device tensors under nccl
os.environ[SWITCH] = "1"
n0 = dict(calls)
assert parallel.broadcast_scalar(0.625) == 0.625
assert parallel.broadcast_scalar(0.625, device=torch.device("cuda:0")) == 0.625
assert parallel.broadcast_scalar_int(2 ** 61 + 5) == 2 ** 61 + 5
assert np.array_equal(parallel.sum_over_ranks(np.array([1.5, -2.0])), [1.5, -2.0])
assert parallel.batch_shard(13) == (0, 13)
assert calls["broadcast"] - n0["broadcast"] == 3 and calls["all_reduce"] - n0["all_reduce"] == 1

# two epochs of the real trainer (valid loader, shuffle, scheduler, early stopper), with and without collectives
def run_trainer(on):
    os.environ[SWITCH] = "1" if on else "0"
    xs, ys = synth.ideal_windows(40, 64, seed=3)
    torch.manual_seed(11)
    model = CamALResNet(kernel_size=7)
    tl = TSLoader(xs[:32], ys[:32], batch_size=12, shuffle=True)   # 12 + 12 + 8: a ragged last batch
    vl = TSLoader(xs[32:], ys[32:], batch_size=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr = BasedClassifTrainer(model, tl, vl, learning_rate=1e-3, weight_decay=0, patience_es=5, patience_rlr=3,
                                 n_warmup_epochs=0, device="cuda:0", all_gpu=True)
    assert tl.sync_shuffle is True
    tr.train(2)
    sd = tr.model.state_dict()
    tr.engine.close()
    return tr.loss_train_history, tr.loss_valid_history, tr.accuracy_train_history, {k: np.asarray(v.cpu() if torch.is_tensor(v) else v) for k, v in sd.items()}

n0 = dict(calls)
t_off = run_trainer(False)
assert calls == n0
t_on = run_trainer(True)
# per epoch: 1 seed broadcast + 3 gradient all-reduces + 1 all-reduce of the epoch's sums + 1 broadcast of the
# validation loss; + the engine's initial sync (4) and rank 0's running statistics before every evaluation / snapshot
assert calls["all_reduce"] - n0["all_reduce"] == 2 * (3 + 1), calls
assert calls["broadcast"] - n0["broadcast"] >= 4 + 2 * 3, calls
assert t_off[0] == t_on[0] and t_off[1] == t_on[1] and t_off[2] == t_on[2]
assert t_off[3].keys() == t_on[3].keys() and all(np.array_equal(t_off[3][k], t_on[3][k]) for k in t_off[3])

dist.barrier()
assert parallel.max_over_ranks(1.5, device="cuda:0") == 1.5
dist.destroy_process_group()
print("RCCL_OK", dist.is_nccl_available(), calls)
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env(**extra):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CAMAL_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "CAMAL_COLLECTIVES_AT_WORLD1"):
        env.pop(k, None)
    env.update(extra)
    return root, env


def test_data_parallel_collectives_through_rccl_at_world_1():
    root, env = _env(CAMAL_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=170)
    assert p.returncode == 0 and "RCCL_OK True" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def test_bench_launcher_path_under_rccl_at_world_1():
    """bench.py as one rank of a launcher (RANK / WORLD_SIZE / MASTER_* in the environment): init_process_group("nccl",
    device_id=...), the barriers around the timed region and the MAX over ranks execute on RCCL, and with the world-1
    switch the training step's gradient all-reduces do as well."""
    root, env = _env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                     CAMAL_COLLECTIVES_AT_WORLD1="1")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "train_fp32", "--also", "none",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=170)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["process_group"] == "nccl"
    assert all(v == v and v < 5.0 for v in line["sanity"]["loss"])  # finite losses after the all-reduced steps


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """The WHOLE N > 1 flow of bench.py on real kernels before a multi-GPU node sees it: `--gpus 2` starts its own two
    ranks; CAMAL_BENCH_REHEARSAL=1 puts both on cuda:0 under "gloo" (RCCL refuses two ranks on one device).  Ranks other
    than 0, the barriers and MAX around every timed region, the step counts agreed between ranks, the data-parallel
    training step with its per-member all-reduce, every `also` workload, rank 0's single JSON line."""
    root, env = _env(CAMAL_BENCH_REHEARSAL="1")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--also", "all", "--also-seconds", "0.2", "--no-cpu-baseline", "--no-extras"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines  # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["process_group"] == "gloo" and line["scaling"] == "weak"
    assert set(line["also"]) == {"ensemble_bf16", "ensemble_fp32", "long_bf16", "train_fp32"}
    assert all("error" not in r and r["value"] > 0 for r in line["also"].values()), line["also"]
    assert "x2" in line["also"]["train_fp32"]["config"]["parallelism"]
    assert all(v == v and v < 5.0 for v in line["also"]["train_fp32"]["sanity"]["loss"])
