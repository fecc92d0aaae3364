"""GPU: the data-parallel training step against the reference's ``all_gpu=True`` semantics
(single-process nn.DataParallel, torch_trainer.py:82-90: per-replica BatchNorm statistics, one mean loss over the
gathered batch, replica 0's running statistics persist) restated by oracle/torch_train.dp_train_steps.

1. Two TrainEngines on ONE GPU play two ranks by hand (ragged shards 7 + 6): weighted gradients summed, AdamW.
2. Two PROCESSES on the one GPU run the whole BasedClassifTrainer under torch.distributed ("gloo": RCCL refuses two
   ranks on one device; the collectives are the same calls) with a ragged last batch.
Tolerances as tests/test_gpu_train.py: gradients 1e-3 of the tensor's max, loss 1e-4, histories 2e-3."""
import os
import socket

import numpy as np
import pytest

from camal_amd import weights
from tests.helpers import is_bn_fed_bias, model_for

pytestmark = pytest.mark.gpu


def _flat_to_dict(eng, model, flat):
    out, o = {}, 0
    for kk in eng.keys:
        n = model._params[kk].size
        out[kk] = flat[o:o + n].reshape(model._params[kk].shape)
        o += n
    return out


# (shapes chosen so that no BatchNorm output of the batch sits within float32 rounding of the ReLU kink: there the
# mask, and with it one term of the weight gradients, legitimately differs between float32 implementations — e.g.
# B = 13, L = 96 of this seed has bn = 2.6e-6 on channel 23 of block 3's first layer, 1.7e-2 of that tensor's max)
@pytest.mark.parametrize("k,B,L", [(7, 13, 100), (15, 8, 61)])
def test_two_shards_by_hand_equal_dataparallel_oracle(k, B, L):
    import torch
    from camal_amd import parallel, synth
    from camal_amd.trainer import TrainEngine
    from oracle import torch_train
    x, y = synth.ideal_windows(B, L, seed=11)
    sd0 = weights.make_state_dict(k, seed=23, head_gain=1.0)
    lr = 1e-3
    # float64 oracle: with 6-7 windows per shard torch's own float32 rounding is of the order of the tolerance
    losses_ref, g_ref, sd_ref = torch_train.dp_train_steps(sd0, [(x, y)], world=2, lr=lr, dtype=torch.float64)
    chunk = -(-B // 2)
    shards = [(0, chunk), (chunk, B)]
    engines = []
    for _ in shards:
        m = model_for(sd0)
        m.to("cuda:0")
        m.load_state_dict(sd0)
        engines.append((m, TrainEngine(m, "cuda:0", B, L, data_parallel=False)))
    loss = 0.0
    for (m, e), (lo, hi) in zip(engines, shards):
        l, _ = e.forward_backward(x[lo:hi], y[lo:hi], global_batch=B)
        loss += float(l.item()) * (hi - lo) / B
    assert abs(loss - losses_ref[0]) < 1e-4
    m0, e0 = engines[0]
    e0.grads.add_(engines[1][1].grads)  # what the sum all-reduce leaves on every rank
    grads = _flat_to_dict(e0, m0, e0.grads.cpu().numpy())
    gmax = max(np.abs(g).max() for g in g_ref.values())
    for kk in e0.keys:
        if is_bn_fed_bias(kk) or np.abs(g_ref[kk]).max() < 1e-5 * gmax:
            assert np.abs(grads[kk]).max() < 1e-4 * gmax, kk
            continue
        err = np.abs(grads[kk] - g_ref[kk]).max() / np.abs(g_ref[kk]).max()
        assert err < 1e-3, (kk, err)
    e0.optimizer_step(lr, 0.0)
    e0.store_to_model(m0)
    sd = {kk: v.numpy() for kk, v in m0.state_dict().items()}
    # the first AdamW step moves every weight by ~lr * sign(g): compare where the gradient is not noise
    for kk in ("linear.weight", "layers.2.layers.2.layers.1.weight", "layers.1.residual.0.weight"):
        assert np.abs(sd[kk] - sd_ref[kk]).max() < 2e-4, kk
    # replica 0's running statistics are the ones that persist
    for kk in sd:
        if kk.endswith("running_mean") or kk.endswith("running_var"):
            assert np.allclose(sd[kk], sd_ref[kk], rtol=1e-4, atol=1e-6), kk
    for _, e in engines:
        e.close()
    assert parallel.rank_world() == (0, 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, tmp, k, n, L, bs, epochs):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from camal_amd import synth
        from camal_amd.trainer import BasedClassifTrainer, TSLoader
        x, y = synth.ideal_windows(n, L, seed=5)
        # every rank but 0 starts from OTHER weights: the trainer must bring rank 0's to all of them
        sd0 = weights.make_state_dict(k, seed=40 + rank, head_gain=1.0)
        m = model_for(sd0)
        m.load_state_dict(sd0)
        torch.manual_seed(100 + rank)  # and from another RNG state: the shuffle must follow rank 0's
        tr = BasedClassifTrainer(m, train_loader=TSLoader(x, y, batch_size=bs, shuffle=True),
                                 valid_loader=TSLoader(x[:8], y[:8], batch_size=1), learning_rate=1e-3, weight_decay=0,
                                 patience_es=5, patience_rlr=3, device="cuda:0", n_warmup_epochs=0)
        tr.train(epochs)
        sd = m.state_dict()
        np.savez(os.path.join(tmp, f"rank{rank}.npz"), loss=np.array(tr.loss_train_history),
                 acc=np.array(tr.accuracy_train_history), vloss=np.array(tr.loss_valid_history),
                 lw=sd["linear.weight"].numpy(), w=sd["layers.2.layers.0.layers.0.weight"].numpy(),
                 rm=sd["layers.1.layers.1.layers.1.running_mean"].numpy())
    finally:
        dist.destroy_process_group()


def test_two_processes_run_the_trainer_data_parallel(tmp_path):
    import torch
    import torch.multiprocessing as mp
    from camal_amd import synth
    from oracle import torch_train
    k, n, L, bs, epochs, world = 7, 30, 64, 12, 2, 2
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path), k, n, L, bs, epochs), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    for key in r0.files:  # the replicas never drift apart: bit-identical weights, statistics and histories
        assert np.array_equal(r0[key], r1[key]), key
    # the oracle walks the batches rank 0's shuffle produced (DataLoader's draws under rank 0's seed)
    x, y = synth.ideal_windows(n, L, seed=5)
    torch.manual_seed(100)
    batches = []
    for _ in range(epochs):
        torch.empty((), dtype=torch.int64).random_()
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        order = torch.randperm(n, generator=torch.Generator().manual_seed(seed)).numpy()
        batches += [(x[order[s:s + bs]], y[order[s:s + bs]]) for s in range(0, n, bs)]
        torch.empty((), dtype=torch.int64).random_()  # the epoch's validation pass creates a loader iterator too
    sd0 = weights.make_state_dict(k, seed=40, head_gain=1.0)
    losses, _, sd_ref = torch_train.dp_train_steps(sd0, batches, world=world, lr=1e-3)
    per_epoch = np.array(losses).reshape(epochs, -1).mean(axis=1)
    assert np.allclose(r0["loss"], per_epoch, rtol=2e-3), (r0["loss"], per_epoch)
    assert np.abs(r0["lw"] - sd_ref["linear.weight"]).max() < 1e-3
    assert np.allclose(r0["rm"], sd_ref["layers.1.layers.1.layers.1.running_mean"], rtol=2e-2, atol=1e-3)
