"""CPU, world_size 2, gloo: the N>1 paths (window sharding without collectives, the flat
gradient-bucket all-reduce, the max-over-ranks timing reduce)."""
import contextlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from camal_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeCamal:
    """Stands in for CamAL on CPU: a deterministic per-window function."""

    def predict(self, data, return_prob=False):
        states = (data > 1.0).astype(np.float64)
        return (states, data.mean(axis=1, keepdims=True)) if return_prob else states


def _worker(rank, world, port, n, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = np.random.default_rng(0).uniform(0, 3, size=(n, 16))
        s, e, res = parallel.predict_sharded(_FakeCamal(), data, return_prob=True)
        np.save(os.path.join(tmp, f"shard{rank}.npy"), np.concatenate([[s, e], res[0].ravel(), res[1].ravel()]))
        # flat gradient bucket: mean over ranks, identical everywhere
        g = torch.full((1000,), float(rank + 1))
        parallel.allreduce_mean_(g)
        assert torch.allclose(g, torch.full((1000,), (1 + world) / 2.0))
        # the same, started early and waited for later (several buckets in flight, as bench.py's training step does)
        gs = [torch.full((257,), float((rank + 1) * (i + 1))) for i in range(3)]
        pend = [parallel.allreduce_mean_start(t) for t in gs]
        for i, p in enumerate(pend):
            assert torch.allclose(p.wait(), torch.full((257,), (i + 1) * (1 + world) / 2.0))
            assert p.wait() is gs[i]  # a second wait is a no-op
        # timing reduce
        t = parallel.max_over_ranks(0.1 * (rank + 1))
        assert abs(t - 0.1 * world) < 1e-12
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 64, 1001):
        for w in (1, 2, 3, 8):
            blocks = [parallel.shard_bounds(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
    with pytest.raises(ValueError):
        parallel.shard_bounds(4, 2, 2)


def test_two_rank_sharded_predict_and_bucket_allreduce(tmp_path):
    n, world = 11, 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    data = np.random.default_rng(0).uniform(0, 3, size=(n, 16))
    whole_states, whole_prob = _FakeCamal().predict(data, return_prob=True)
    got_states = np.zeros_like(whole_states)
    got_prob = np.zeros_like(whole_prob)
    for r in range(world):
        a = np.load(tmp_path / f"shard{r}.npy")
        s, e = int(a[0]), int(a[1])
        k = (e - s) * 16
        got_states[s:e] = a[2:2 + k].reshape(e - s, 16)
        got_prob[s:e] = a[2 + k:].reshape(e - s, 1)
    assert np.array_equal(got_states, whole_states) and np.allclose(got_prob, whole_prob)


# ---- members spread over ranks (CamAL.member_parallel; SURVEY §8 e/f4) ----------------------------------------
def _train_worker(rank, world, port, tmp):
    import json
    import camal_amd.trainer as trainer_mod
    from camal_amd.core import CamAL
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class FakeTrainer:
        """Stands in for the GPU trainer: the orchestration around it is what runs here."""

        def __init__(self, model, train_loader, valid_loader=None, **kw):
            self.k, self.data_parallel, self.log = model.kernel_size, kw["data_parallel"], {}
            self.train_loader, self.valid_loader = train_loader, valid_loader
            trainer_mod.configure_loader_sync(train_loader, kw["data_parallel"], None)  # as BasedClassifTrainer.__init__

        def train(self, epochs):
            # the REAL loaders are walked, a different number of epochs per candidate (early stopping) and a different
            # number of candidates per rank (3 + 2): a collective inside the loader (the shuffle-seed
            # broadcast) would stop pairing up between the ranks and meet merge_member_logs' all_gather_object
            self.epochs_run = {5: 1, 7: 4, 9: 2, 15: 3, 25: 5}[self.k]
            self.batches = 0
            for _ in range(self.epochs_run):
                self.batches += sum(1 for _ in self.train_loader)
                sum(1 for _ in self.valid_loader)
            self.log = {"best_model_state_dict": {"w": torch.full((3,), float(self.k))}, "trained_on_rank": rank,
                        "data_parallel": self.data_parallel, "batches": self.batches}

        def restore_best_weights(self):
            pass

        def evaluate(self, loader):
            return abs(self.k - 9) + 0.5, None  # best: 9, then 7, then 5 / 15 ...

    try:
        trainer_mod.BasedClassifTrainer = FakeTrainer
        rng = np.random.default_rng(3)
        X = rng.normal(size=(40, 32)).astype(np.float32)
        y = (np.arange(40) % 2).astype(np.float32)
        camal = CamAL(n_predictors=3, device="cuda")
        camal.member_parallel = True
        camal.eval = lambda: None  # materialising the ensemble needs the GPU
        camal.train((X, y), (X[:4], y[:4]), (X[4:8], y[4:8]), list_kernel_sizes=[5, 7, 9, 15, 25], epochs=1, batch_size=16)
        logs = camal.camal_logs
        summary = {
            "ids": logs["camal_predictors_id"],
            "members": {n: [l["kernel_size"], l["trained_on_rank"], l["data_parallel"], "best_model_state_dict" in l]
                        for n, l in logs["ensemble_training_logs"].items()},
            "w9": float(logs["ensemble_training_logs"]["Predictor_2"]["best_model_state_dict"]["w"][0]),
            "batches": [l["batches"] for l in logs["ensemble_training_logs"].values()],
        }
        with open(os.path.join(tmp, f"train{rank}.json"), "w") as f:
            json.dump(summary, f)
    finally:
        dist.destroy_process_group()


def test_two_rank_member_parallel_training_orchestration(tmp_path):
    import json
    world = 2
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [json.load(open(tmp_path / f"train{r}.json")) for r in range(world)]
    assert got[0] == got[1]  # every rank ends with the same logs
    g = got[0]
    assert g["ids"] == ["Predictor_2", "Predictor_1", "Predictor_0"]  # kernels 9, 7, 5 by test loss
    assert list(g["members"]) == [f"Predictor_{i}" for i in range(5)]  # the reference's insertion order
    for i, (name, (k, trained_on, dp, has_sd)) in enumerate(g["members"].items()):
        assert k == [5, 7, 9, 15, 25][i]
        assert trained_on == parallel.member_owner(i, world) and dp is False
        assert has_sd == (name in g["ids"])  # the losers' snapshots are dropped (core.py:176-179)
    assert g["w9"] == 9.0
    assert g["batches"] == [e * 3 for e in (1, 4, 2, 3, 5)]  # 40 balanced samples, batch 16: 3 batches per epoch


def test_member_owner_round_robin():
    assert [parallel.member_owner(i, 8) for i in range(5)] == [0, 1, 2, 3, 4]
    assert [parallel.member_owner(i, 2) for i in range(5)] == [0, 1, 0, 1, 0]
    assert parallel.merge_member_logs({"a": 1}) == {"a": 1}  # no process group: identity


# ---- data-parallel training step: ordering of the asynchronous gradient all-reduce ----------------------------
class _FakeLib:
    """Stands in for libcamal_hip.so: records what the flat gradient bucket holds when AdamW is launched."""

    def __init__(self, eng):
        self.eng, self.seen = eng, []

    def camal_adamw_step(self, params, grads, m, v, n, step, *rest):
        self.seen.append((int(step), self.eng.grads.clone()))
        return 0


def _fake_engine(n, value):
    from camal_amd.trainer import TrainEngine
    e = object.__new__(TrainEngine)  # no GPU, no library: only the buffers the collective path touches
    e.device, e.data_parallel, e.n, e.step_count = None, True, n, 0
    e.grads = torch.full((n,), float(value))
    e.params, e.m, e.v = torch.zeros(n), torch.zeros(n), torch.zeros(n)
    e.running = torch.zeros(4)
    e._stream = lambda: None
    e._device_ctx = contextlib.nullcontext
    e.lib = _FakeLib(e)
    return e


def _dp_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # three members as bench.py steps them: start every all-reduce, then wait + update member by member
        engs = [_fake_engine(1000 + i, (rank + 1) * (i + 1)) for i in range(3)]
        pend = [e.start_grad_allreduce() for e in engs]
        for e, p in zip(engs, pend):
            e.optimizer_step(1e-3, 0.0, pending=p)
        for i, e in enumerate(engs):
            (step, g), = e.lib.seen
            want = sum((r + 1) * (i + 1) for r in range(world))  # SUM: the shares already carry the 1 / world
            assert step == 1 and torch.equal(g, torch.full((1000 + i,), float(want))), (i, g[:3])
        # without a pending handle optimizer_step reduces by itself; a member-parallel engine reduces nothing
        e = _fake_engine(10, rank + 1)
        e.optimizer_step(1e-3, 0.0)
        assert torch.equal(e.lib.seen[0][1], torch.full((10,), 3.0))
        e = _fake_engine(10, rank + 1)
        e.data_parallel = False
        e.optimizer_step(1e-3, 0.0, pending=e.start_grad_allreduce())
        assert torch.equal(e.lib.seen[0][1], torch.full((10,), float(rank + 1)))
        # replicas start from rank 0's buffers, and rank 0's running statistics are the ones that persist
        e = _fake_engine(10, 0)
        e.params += rank + 5
        e.running += rank + 7
        e.sync_from_rank0()
        assert torch.equal(e.params, torch.full((10,), 5.0)) and torch.equal(e.running, torch.full((4,), 7.0))
        # batch shards (nn.DataParallel's scatter), shared shuffle seed, shared decision value
        assert parallel.batch_shard(13) == ((0, 7) if rank == 0 else (7, 13))
        assert parallel.batch_shard(1) == ((0, 1) if rank == 0 else (1, 1))
        from camal_amd.trainer import TSLoader
        torch.manual_seed(50 + rank)  # different RNG states: the order must still be rank 0's
        loader = TSLoader(np.zeros((9, 4)), np.arange(9), batch_size=4, shuffle=True)
        assert loader.sync_shuffle is False  # opt-in: only a data-parallel trainer turns it on
        loader.sync_shuffle = True
        order = [lab.tolist() for _, lab in loader]
        torch.manual_seed(50)
        torch.empty((), dtype=torch.int64).random_()
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        ref = torch.randperm(9, generator=torch.Generator().manual_seed(seed)).tolist()
        assert sum(order, []) == ref
        assert parallel.broadcast_scalar(0.25 * (rank + 1)) == 0.25
        assert np.array_equal(parallel.sum_over_ranks(np.array([1.0, rank])), [2.0, 1.0])
        open(os.path.join(tmp, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_step_ordering(tmp_path):
    mp.spawn(_dp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
