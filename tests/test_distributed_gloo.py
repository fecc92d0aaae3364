"""CPU, world_size 2, gloo: the N>1 paths (window sharding without collectives, the flat
gradient-bucket all-reduce, the max-over-ranks timing reduce)."""
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
