"""``BasedClassifTrainer`` — mirror of the reference's classifier trainer
(``src/helpers/torch_trainer.py:15-325``): same constructor / ``train`` / ``evaluate`` /
``restore_best_weights`` / ``log`` schema, with the per-batch forward + backward + AdamW running in
libcamal_hip.so (``camal_train_fwd_bwd`` / ``camal_adamw_step``).

``all_gpu=True`` in the reference wraps the model in single-process ``nn.DataParallel``
(:82-90).  Here multi-GPU training is one process per GPU: when ``torch.distributed`` is
initialised, every step all-reduces ONE flat fp32 gradient bucket (mean over ranks) on RCCL
between backward and the optimizer; BatchNorm statistics stay per rank, like DataParallel.
"""
from __future__ import annotations

import ctypes as C
import os
import time
import warnings

import numpy as np
import torch

from . import _lib, parallel, weights
from .metrics import Classifmetrics


class TSLoader:
    """Minimal stand-in for ``DataLoader(TSDataset(X, y), batch_size, shuffle)``
    (reference core.py:92-119, torch_dataset.py:197-250): yields ``(ts [b,1,L], labels [b])``
    NumPy batches.  ``shuffle=True`` consumes the global torch RNG exactly like ``DataLoader(shuffle=True)``
    does per epoch (one int64 for the iterator's base seed, one for ``RandomSampler``'s seed, then
    ``randperm`` from a fresh generator seeded with it), so the batch order under ``torch.manual_seed`` is the
    reference's.  ``sync_shuffle`` (set by a DATA-PARALLEL trainer only, never by default): every rank uses rank 0's
    seed, so the ranks cut the SAME batches — one broadcast per shuffled pass, which every rank of the process group
    must then make (ranks that train different members, CamAL.member_parallel, iterate their loaders a different
    number of times and must not)."""

    sync_shuffle = False
    sync_device = None  # device of the seed tensor under RCCL (None: the current GPU)

    def __init__(self, X, y, batch_size=1, shuffle=False, generator=None):
        """generator: like DataLoader's — a torch.Generator the shuffles draw from instead of the global RNG (members
        trained concurrently, CamAL.member_streams, each get their own, so that a run stays reproducible)."""
        self.generator = generator
        X = np.asarray(X)
        self.samples = X[:, None, :] if X.ndim == 2 else X
        self.labels = np.asarray(y).ravel()
        assert len(self.samples) == len(self.labels), (
            f"Number of X sample {len(self.samples)} doesn't match number of y sample {len(self.labels)}.")
        self.batch_size, self.shuffle = int(batch_size), bool(shuffle)

    def __len__(self):
        return (len(self.samples) + self.batch_size - 1) // self.batch_size

    def draw_base_seed(self):
        """Every DataLoader iterator, shuffling or not, draws one int64 from the global torch RNG (or the loader's
        generator) when it is created (``_BaseDataLoaderIter``'s base seed): a pass over a loader here consumes the
        same, so that a run under ``torch.manual_seed`` sees the reference's shuffles epoch after epoch (its
        validation passes included)."""
        torch.empty((), dtype=torch.int64).random_(generator=getattr(self, "generator", None))

    def epoch_order(self):
        """The sample order of one pass, drawing from the RNG exactly what ``DataLoader``'s iterator would."""
        n = len(self.samples)
        self.draw_base_seed()
        if self.shuffle and self.generator is not None:  # RandomSampler with a generator permutes from it directly
            order = torch.randperm(n, generator=self.generator).numpy()
        elif self.shuffle:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())  # RandomSampler.__iter__
            if self.sync_shuffle:
                seed = int(parallel.broadcast_scalar_int(seed, device=self.sync_device))
            order = torch.randperm(n, generator=torch.Generator().manual_seed(seed)).numpy()
        else:
            order = np.arange(n)
        return order

    def __iter__(self):
        n = len(self.samples)
        order = self.epoch_order()
        for s in range(0, n, self.batch_size):
            idx = order[s:s + self.batch_size]
            yield self.samples[idx], self.labels[idx]


class TSDataset(torch.utils.data.Dataset):
    """The reference's map-style dataset (torch_dataset.py:197-250): ``samples`` [n, C, L] (a 2-D X gets the channel
    axis), ``labels`` [n] or None, optional per-channel StandardScaler (sklearn, as there).  What callers of the
    reference wrap in a ``DataLoader`` and hand to ``BasedClassifTrainer``: the trainer here takes that loader as it is
    (``as_tsloader``)."""

    def __init__(self, X, labels=None, scaler=False, scale_dim=None):
        X = X.values if hasattr(X, "values") else X          # a pandas DataFrame
        labels = labels.values if hasattr(labels, "values") else labels
        if scaler:
            from sklearn.preprocessing import StandardScaler
            if len(X.shape) == 3:
                self.scaler_list, self.samples = [], X
                for i in (range(X.shape[1]) if scale_dim is None else scale_dim):
                    self.scaler_list.append(StandardScaler())
                    self.samples[:, i, :] = self.scaler_list[-1].fit_transform(X[:, i, :].T).T.astype(np.float32)
            else:
                self.scaler_list = [StandardScaler()]
                self.samples = self.scaler_list[0].fit_transform(X.T).T.astype(np.float32)
        else:
            self.samples = X
        if len(self.samples.shape) == 2:
            self.samples = np.expand_dims(self.samples, axis=1)
        if labels is not None:
            self.labels = labels.ravel()
            assert len(self.samples) == len(self.labels), (
                f"Number of X sample {len(self.samples)} doesn't match number of y sample {len(self.labels)}.")
        else:
            self.labels = labels

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        if self.labels is None:
            return self.samples[idx]
        return self.samples[idx], self.labels[idx]


def as_tsloader(loader):
    """A ``TSLoader`` for what the caller passed: a TSLoader itself, or a ``torch.utils.data.DataLoader`` over a dataset
    with ``samples`` / ``labels`` arrays (the reference's ``DataLoader(TSDataset(X, y), batch_size, shuffle)``,
    core.py:92-119) — same batch size, shuffling iff its sampler is a RandomSampler, its generator.  The trainer needs the
    arrays themselves (the training set stays resident in HBM, evaluation runs as one batched forward), not a stream of
    collated batches.  None stays None; anything else is returned unchanged (an iterable of (ts, labels) batches still
    trains, one host-to-device copy per batch)."""
    if loader is None or isinstance(loader, TSLoader):
        return loader
    ds = getattr(loader, "dataset", None)
    if isinstance(loader, torch.utils.data.DataLoader) and hasattr(ds, "samples") and getattr(ds, "labels", None) is not None:
        if loader.batch_size is None or getattr(loader, "drop_last", False):
            raise ValueError("BasedClassifTrainer takes DataLoaders with a batch_size and drop_last=False")
        shuffle = isinstance(loader.sampler, torch.utils.data.RandomSampler)
        return TSLoader(np.asarray(ds.samples), np.asarray(ds.labels), batch_size=loader.batch_size, shuffle=shuffle,
                        generator=loader.generator)
    return loader


class EarlyStopper:
    """torch_trainer.py:1098-1113."""

    def __init__(self, patience=1, min_delta=0):
        self.patience, self.min_delta = patience, min_delta
        self.counter, self.min_validation_loss = 0, np.inf

    def early_stop(self, validation_loss):
        if validation_loss < self.min_validation_loss:
            self.min_validation_loss = validation_loss
            self.counter = 0
        elif validation_loss > (self.min_validation_loss + self.min_delta):
            self.counter += 1
            if self.counter >= self.patience:
                return True
        return False


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau('min', patience, eps=1e-7) as configured at
    torch_trainer.py:63-66 (factor 0.1, rel threshold 1e-4, no cooldown, min_lr 0)."""

    def __init__(self, lr, patience=10, factor=0.1, threshold=1e-4, eps=1e-7, min_lr=0.0):
        self.lr, self.patience, self.factor, self.threshold, self.eps, self.min_lr = lr, patience, factor, threshold, eps, min_lr
        self.best, self.num_bad_epochs = float("inf"), 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad_epochs = metric, 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            new_lr = max(self.lr * self.factor, self.min_lr)
            if self.lr - new_lr > self.eps:
                self.lr = new_lr
            self.num_bad_epochs = 0
        return self.lr


class TrainEngine:
    """Flat device buffers (parameters, gradients, Adam moments, BN running stats) of one
    ``CamALResNet`` + the ``camal_trainer`` workspace."""

    BETAS, EPS, BN_MOMENTUM = (0.9, 0.999), 1e-8, 0.1  # torch defaults (AdamW, BatchNorm1d)

    def __init__(self, model, device, max_batch, length, data_parallel=True):
        self.lib = _lib.load()
        # False: this rank trains the member alone (members spread over ranks, CamAL.train)
        self.data_parallel = bool(data_parallel)
        d = torch.device(device)
        if d.type != "cuda":
            raise _lib.CamalHipError(f"camal_amd trains on MI355X only (device={device!r}); there is no CPU path.")
        self.device = torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device())
        if (model.in_channels, model.nb_class) != (1, 2) or model.mid_channels not in (32, 64, 128):
            raise NotImplementedError("the HIP training path implements in_channels=1, nb_class=2, mid_channels in (32, 64, 128)")
        self.k = model.kernel_size
        self.mid = int(model.mid_channels)
        self.arch = getattr(model, "_arch", weights.ARCH_CAMAL)
        arch_id = {weights.ARCH_CAMAL: _lib.ARCH_CAMAL, weights.ARCH_RESNET: _lib.ARCH_RESNET}[self.arch]
        self.keys = weights.trainable_keys(self.k, mid_channels=self.mid, arch=self.arch)
        self.bn_prefixes = [k[:-len(".running_mean")] for k in model._params if k.endswith("running_mean")]
        self.n = int(self.lib.camal_trainer_param_count_widths(self.k, arch_id, self.mid))
        assert self.n == sum(model._params[k].size for k in self.keys)
        self.n_running = int(self.lib.camal_trainer_running_count_widths(self.mid))
        h = C.c_void_p()
        _lib.check(self.lib.camal_trainer_create_widths(self.device.index, self.k, int(max_batch), int(length), arch_id,
                                                        self.mid, C.byref(h)))
        self._h = h
        self.max_batch, self.length = int(max_batch), int(length)
        with torch.cuda.device(self.device):
            self.params = torch.empty(self.n, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros_like(self.params)
            self.m = torch.zeros_like(self.params)
            self.v = torch.zeros_like(self.params)
            self.running = torch.empty(self.n_running, dtype=torch.float32, device=self.device)
            self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
            self.ncorrect = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.step_count = 0
        self._grad_scale = 1.0
        self.load_from_model(model)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.camal_trainer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- model <-> flat buffers ------------------------------------------------------------
    def load_from_model(self, model):
        flat = np.concatenate([np.asarray(model._params[k], np.float32).ravel() for k in self.keys])
        run = np.concatenate([np.concatenate([model._params[p + ".running_mean"].ravel(),
                                              model._params[p + ".running_var"].ravel()]) for p in self.bn_prefixes])
        self.params.copy_(torch.from_numpy(flat))
        self.running.copy_(torch.from_numpy(run.astype(np.float32)))
        self.num_batches_tracked = int(model._params[self.bn_prefixes[0] + ".num_batches_tracked"])
        self.sync_from_rank0()

    def _dp(self):
        return self.data_parallel and parallel.collectives_on()

    def sync_from_rank0(self):
        """Data-parallel replicas start from ONE model: rank 0's parameters, running statistics and Adam moments
        (nn.DataParallel replicates device 0's module, torch_trainer.py:89)."""
        if self._dp():
            for t in (self.params, self.running, self.m, self.v):
                parallel.broadcast_(t, 0)

    def store_to_model(self, model):
        if self._dp():
            # BatchNorm running statistics are per rank during a step; the ones that persist are replica 0's, as
            # with nn.DataParallel (every forward re-replicates device 0's buffers)
            parallel.broadcast_(self.running, 0)
        flat = self.params.cpu().numpy()
        run = self.running.cpu().numpy()
        o = 0
        for k in self.keys:
            n = model._params[k].size
            model._params[k] = flat[o:o + n].reshape(model._params[k].shape).copy()
            o += n
        o = 0
        for p in self.bn_prefixes:
            c = model._params[p + ".running_mean"].size
            model._params[p + ".running_mean"] = run[o:o + c].copy()
            model._params[p + ".running_var"] = run[o + c:o + 2 * c].copy()
            model._params[p + ".num_batches_tracked"] = np.array(self.num_batches_tracked, np.int64)
            o += 2 * c
        model._drop_engine()

    # ---- one batch ----------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _device_ctx(self):
        return torch.cuda.device(self.device)

    def _prep(self, ts, labels):
        x = torch.as_tensor(np.ascontiguousarray(np.asarray(ts, dtype=np.float32))) if not torch.is_tensor(ts) else ts.float()
        if x.dim() == 3:
            x = x[:, 0, :]
        x = x.to(self.device).contiguous()
        y = None
        if labels is not None:
            y = labels if torch.is_tensor(labels) else torch.as_tensor(np.asarray(labels))
            if not (y.dtype == torch.int32 and y.is_cuda):  # labels.float() ... .long() of torch_trainer.py:285,290
                y = y.to(self.device).float().long().to(torch.int32)
            y = y.contiguous()
        return x, y

    def forward_backward(self, ts, labels, update_running=True, global_batch=None, loss_out=None, ncorrect_out=None,
                         criterion=None):
        """Train-mode forward + backward on THIS RANK's samples; fills ``self.grads``.
        -> (loss tensor[1] = this rank's mean loss, ncorrect tensor[1]).

        Data-parallel (``torch.distributed`` up, ``data_parallel``): the gradients are weighted by this rank's share
        ``len(ts) / global_batch`` of the step's batch (default: equal shares, 1 / world), so the SUM all-reduce in
        ``optimizer_step`` yields the gradient of the mean loss over the whole batch; a rank whose shard is empty
        contributes zeros.  ``loss_out`` / ``ncorrect_out``: one-element device tensors to write into instead of
        ``self.loss`` / ``self.ncorrect`` (the trainer's per-epoch history: no host sync per batch).

        ``criterion``: a caller-supplied training loss (torch_trainer.py:20,52,290: any callable on (logits, labels
        as long)); None = the reference's default nn.CrossEntropyLoss(), computed in the library.  With a criterion
        the step is forward (library) -> loss and its gradient with respect to the [b, 2] logits (torch autograd) ->
        backward from that gradient (``camal_train_bwd_dlogits``); the loss written out is the criterion's."""
        loss_out = self.loss if loss_out is None else loss_out
        ncorrect_out = self.ncorrect if ncorrect_out is None else ncorrect_out
        n = len(ts)
        world = parallel.rank_world()[1] if self._dp() else 1
        scale = (n / float(global_batch)) if global_batch else 1.0 / world
        if n == 0:
            self.grads.zero_()
            loss_out.zero_()
            ncorrect_out.zero_()
            return loss_out, ncorrect_out
        x, y = self._prep(ts, labels)
        B, L = x.shape
        if scale != self._grad_scale:
            _lib.check(self.lib.camal_trainer_set_grad_scale(self._h, float(scale)))
            self._grad_scale = scale
        with torch.cuda.device(self.device):
            if criterion is None:
                _lib.check(self.lib.camal_train_fwd_bwd(
                    self._h, _lib.ptr(self.params), _lib.ptr(self.running) if update_running else None, _lib.ptr(x),
                    _lib.ptr(y), B, L, self.BN_MOMENTUM, _lib.ptr(self.grads), None, _lib.ptr(loss_out),
                    _lib.ptr(ncorrect_out), self._stream()))
            else:
                logits = torch.empty((B, 2), dtype=torch.float32, device=self.device)
                _lib.check(self.lib.camal_train_fwd_bwd(
                    self._h, _lib.ptr(self.params), _lib.ptr(self.running) if update_running else None, _lib.ptr(x),
                    None, B, L, self.BN_MOMENTUM, None, _lib.ptr(logits), None, None, self._stream()))
                lg = logits.requires_grad_(True)
                with torch.enable_grad():
                    loss = criterion(lg.float(), y.long())  # torch_trainer.py:290
                dlog, = torch.autograd.grad(loss, lg)
                dlog = dlog.contiguous().float()
                ce = torch.empty(1, dtype=torch.float32, device=self.device)
                _lib.check(self.lib.camal_train_bwd_dlogits(
                    self._h, _lib.ptr(self.params), _lib.ptr(x), _lib.ptr(y), _lib.ptr(dlog), B, L, _lib.ptr(self.grads),
                    _lib.ptr(ce), _lib.ptr(ncorrect_out), self._stream()))
                loss_out.copy_(loss.detach().reshape(1))
        if update_running:
            self.num_batches_tracked += 1
        return loss_out, ncorrect_out

    def forward_train(self, ts):
        """Train-mode logits (batch statistics), no backward, running stats untouched."""
        x, _ = self._prep(ts, None)
        B, L = x.shape
        with torch.cuda.device(self.device):
            logits = torch.empty((B, 2), dtype=torch.float32, device=self.device)
            _lib.check(self.lib.camal_train_fwd_bwd(self._h, _lib.ptr(self.params), None, _lib.ptr(x), None, B, L,
                                                    self.BN_MOMENTUM, None, _lib.ptr(logits), None, None, self._stream()))
        return logits

    PEEK = {"z": 0, "act": 1, "mean": 2, "invstd": 3}

    def peek(self, what, index, B, L):
        """A copy of what the last forward pass (B windows of L steps) left in the workspace (camal_trainer_peek):
        ``what`` = "z" (pre-BatchNorm conv output of slot ``index`` -> [B, L, C]), "act" (ConvBlock / block output ->
        [B, L, C]), "mean" / "invstd" (batch statistics -> [C]).  For layer-level parity tests (the reference's
        counterpart is a forward hook on the module)."""
        import ctypes as C
        ch = C.c_int()
        code = self.PEEK[what]
        _lib.check(self.lib.camal_trainer_peek(self._h, code, int(index), int(B), int(L), None, 0, C.byref(ch), None))
        stat = what in ("mean", "invstd")
        n = ch.value if stat else int(B) * int(L) * ch.value
        with torch.cuda.device(self.device):
            out = torch.empty(n, dtype=torch.float32, device=self.device)
            _lib.check(self.lib.camal_trainer_peek(self._h, code, int(index), int(B), int(L), _lib.ptr(out), n, C.byref(ch),
                                                   self._stream()))
        return out if stat else out.view(int(B), int(L), ch.value)

    # ---- the training set resident in HBM ------------------------------------------------------------
    RESIDENT_FRACTION = 0.25  # of the GPU's free memory a training set may take to stay resident

    def make_resident(self, loader):
        """Copy a TSLoader's windows and labels to the GPU once (pinned staging, one asynchronous copy) if they fit in a
        quarter of the free HBM (288 GB per MI355X: every set the reference's data builders produce does): an epoch then
        moves a permutation to the device instead of one synchronous pageable copy per batch (torch_trainer.py:284-285).
        -> (X [n, L] float32, y [n] int32) device tensors, or None."""
        X = loader.samples[:, 0, :] if loader.samples.ndim == 3 else loader.samples
        need = X.size * 4 + loader.labels.size * 4
        with torch.cuda.device(self.device):
            if need > self.RESIDENT_FRACTION * torch.cuda.mem_get_info(self.device)[0]:
                return None
            hx = torch.empty(X.shape, dtype=torch.float32).pin_memory()
            hx.numpy()[...] = X
            hy = torch.from_numpy(np.asarray(loader.labels, dtype=np.float32).astype(np.int32)).pin_memory()
            xd = hx.to(self.device, non_blocking=True)
            yd = hy.to(self.device, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()  # the pinned staging buffers go away with this frame
        return xd, yd

    def start_grad_allreduce(self):
        """Begin this member's gradient all-reduce and return at once; hand the result to ``optimizer_step(...,
        pending=...)`` later (callers that step several members overlap it with the next member's work)."""
        if not self.data_parallel:
            return parallel.PendingMean(self.grads, None, 1)
        return parallel.allreduce_sum_start(self.grads)

    def optimizer_step(self, lr, weight_decay, pending=None):
        if pending is not None:
            pending.wait()
        elif self.data_parallel:
            parallel.allreduce_sum_(self.grads)  # no-op unless torch.distributed is initialised
        self.step_count += 1
        with self._device_ctx():
            _lib.check(self.lib.camal_adamw_step(_lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.m),
                                                 _lib.ptr(self.v), self.n, self.step_count, float(lr), self.BETAS[0],
                                                 self.BETAS[1], self.EPS, float(weight_decay), self._stream()))

    def optimizer_state_dict(self, lr, weight_decay):
        return {"state": {"step": self.step_count, "exp_avg": self.m.cpu(), "exp_avg_sq": self.v.cpu(),
                          "param_order": list(self.keys)},
                "param_groups": [{"lr": lr, "betas": self.BETAS, "eps": self.EPS, "weight_decay": weight_decay}]}


class MemberStreams:
    """The members of an ensemble are independent (core.py:104-158): their training steps run side by side on one
    GPU, one HIP stream per member, so that one member's short kernels (BatchNorm passes, reductions, kernel tails)
    fill the gaps of another's.  Measured on 5 members, 128 x 510: 16.1 -> 13.8 ms per ensemble step.
    Collectives (data-parallel) are still issued from this one host thread in member order, the same on every rank."""

    def __init__(self, engines):
        self.engines = list(engines)
        self.streams = [torch.cuda.Stream(e.device) for e in self.engines]

    def step(self, x, y, lr, weight_decay, global_batch=None):
        """forward + backward + (all-reduce) + AdamW of every member on its own stream; returns when everything is
        enqueued, the caller's current stream waits for all of it."""
        cur = torch.cuda.current_stream(self.engines[0].device)
        pending = []
        for e, s in zip(self.engines, self.streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                e.forward_backward(x, y, global_batch=global_batch)
                pending.append(e.start_grad_allreduce())
        for e, s, p in zip(self.engines, self.streams, pending):
            with torch.cuda.stream(s):
                e.optimizer_step(lr, weight_decay, pending=p)
        for s in self.streams:
            cur.wait_stream(s)


def forward_train(model, x):
    """``CamALResNet.forward`` in train() mode outside a trainer: batch-statistics logits."""
    xs = np.asarray(x.cpu() if torch.is_tensor(x) else x, dtype=np.float32)
    if xs.ndim == 3:
        xs = xs[:, 0, :]
    eng = TrainEngine(model, model.device, xs.shape[0], xs.shape[1], data_parallel=False)
    try:
        return eng.forward_train(xs)
    finally:
        eng.close()


def configure_loader_sync(train_loader, data_parallel, device):
    """Only a DATA-PARALLEL trainer shares rank 0's shuffle (one seed broadcast per epoch that every rank of the
    process group must make); a trainer whose rank trains its members alone (CamAL.member_parallel) must not."""
    if hasattr(train_loader, "sync_shuffle"):
        train_loader.sync_shuffle = bool(data_parallel)
        train_loader.sync_device = device


def _is_default_ce(criterion):
    """nn.CrossEntropyLoss() as the reference constructs it: mean reduction, no class weights, no smoothing."""
    return (isinstance(criterion, torch.nn.CrossEntropyLoss) and criterion.weight is None and criterion.reduction == "mean"
            and criterion.label_smoothing == 0.0 and criterion.ignore_index == -100)


class BasedClassifTrainer(object):
    def __init__(self, model, train_loader, valid_loader=None, learning_rate=1e-3, weight_decay=1e-2,
                 criterion=None, patience_es=None, patience_rlr=None, device="cuda", all_gpu=False,
                 valid_criterion=None, n_warmup_epochs=0, f_metrics=Classifmetrics(), verbose=False,
                 plotloss=False, save_fig=False, path_fig=None, save_checkpoint=False, path_checkpoint=None,
                 data_parallel=True):
        # criterion (torch_trainer.py:20,52): None or the default nn.CrossEntropyLoss() -> loss and its gradient in the
        # library; anything else is evaluated by torch on the [b, 2] logits of every batch, and the library runs the
        # backward pass from its gradient (camal_train_bwd_dlogits)
        self.train_criterion = None if criterion is None or _is_default_ce(criterion) else criterion
        # valid_criterion (torch_trainer.py:23,53-56: the training criterion unless given): the default cross-entropy
        # -> the HIP evaluation kernel; anything else is applied to the eval-mode logits batch by batch, in torch
        if valid_criterion is None:
            valid_criterion = criterion
        self.valid_criterion = None if valid_criterion is None or _is_default_ce(valid_criterion) else valid_criterion
        train_loader, valid_loader = as_tsloader(train_loader), as_tsloader(valid_loader)
        self._init_host(model, train_loader, valid_loader, learning_rate, weight_decay, patience_es, patience_rlr, device,
                        all_gpu, n_warmup_epochs, f_metrics, verbose, plotloss, save_fig, path_fig, save_checkpoint,
                        path_checkpoint)
        self.model.to(self.device)
        length = train_loader.samples.shape[-1]
        # data_parallel is not in the reference: see CamAL.member_parallel
        self.engine = TrainEngine(self.model, self.device, train_loader.batch_size, length, data_parallel=data_parallel)
        if all_gpu and not parallel.collectives_on() and torch.cuda.device_count() > 1:
            # the reference's all_gpu=True wraps the model in single-process nn.DataParallel (torch_trainer.py:82-90);
            # here data-parallel training is one process per GPU, which a single process cannot become by itself
            warnings.warn("all_gpu=True: this process trains on one GPU.  Multi-GPU data-parallel training is one process "
                          "per GPU: launch the same script with `python -m torch.distributed.run --nproc-per-node N ...` "
                          "and call torch.distributed.init_process_group('nccl') first.  CamAL.train(all_gpu=True) / "
                          "CamAL.devices use every GPU from one process by training the ensemble's members on different "
                          "GPUs (no gradient exchange).")
        configure_loader_sync(train_loader, data_parallel, self.engine.device)

    def _init_host(self, model, train_loader, valid_loader, learning_rate, weight_decay, patience_es, patience_rlr,
                   device, all_gpu, n_warmup_epochs, f_metrics, verbose=False, plotloss=False, save_fig=False,
                   path_fig=None, save_checkpoint=False, path_checkpoint=None):
        """Everything of torch_trainer.py:33-80 that is host state (no device, no library): the epoch logic of
        ``train`` runs on it alone, which is how the CPU suite checks it against the reference's decisions."""
        self.model = model
        self.train_loader, self.valid_loader = train_loader, valid_loader
        self.f_metrics, self.device, self.all_gpu = f_metrics, device, all_gpu
        self.verbose, self.plotloss = verbose, plotloss
        self.save_checkpoint, self.save_fig, self.path_fig = save_checkpoint, save_fig, path_fig
        self.path_checkpoint = path_checkpoint if path_checkpoint is not None else os.getcwd() + os.sep + "model"
        self.patience_rlr, self.patience_es, self.n_warmup_epochs = patience_rlr, patience_es, n_warmup_epochs
        self.lr, self.weight_decay = learning_rate, weight_decay
        if not hasattr(self, "valid_criterion"):
            self.valid_criterion = None
        self.scheduler = ReduceLROnPlateau(learning_rate, patience=patience_rlr, eps=1e-7) if patience_rlr is not None else None

        self.log = {}
        self.train_time = self.eval_time = self.voter_time = 0
        self.passed_epochs = 0
        self.best_loss = np.inf
        self.loss_train_history, self.loss_valid_history = [], []
        self.accuracy_train_history, self.accuracy_valid_history = [], []
        if self.patience_es is not None:
            self.early_stopping = EarlyStopper(patience=self.patience_es)

    # ------------------------------------------------------------- train (torch_trainer.py:92-171)
    def train(self, n_epochs=10):
        tmp_time = time.time()
        for epoch in range(n_epochs):
            train_loss, train_accuracy = self.__train()
            self.loss_train_history.append(train_loss)
            self.accuracy_train_history.append(train_accuracy)
            if self.valid_loader is not None:
                valid_loss, valid_accuracy = self.__evaluate()
                self.loss_valid_history.append(valid_loss)
                self.accuracy_valid_history.append(valid_accuracy)
            else:
                valid_loss = train_loss
            if self.engine._dp():
                # every rank must take the same scheduler / early-stop / snapshot decisions (their collectives have
                # to match): all of them act on rank 0's number
                valid_loss = parallel.broadcast_scalar(valid_loss, 0, device=self.engine.device)

            if self.scheduler:
                self.lr = self.scheduler.step(valid_loss)

            if self.patience_es is not None:
                if self.passed_epochs > self.n_warmup_epochs:
                    if self.early_stopping.early_stop(valid_loss):
                        self.passed_epochs += 1
                        if self.verbose:
                            print("Early stopping after {} epochs !".format(epoch + 1))
                        break

            if self.verbose:
                print("Epoch [{}/{}]".format(epoch + 1, n_epochs))
                print("    Train loss : {:.4f}, Train acc : {:.2f}%".format(train_loss, train_accuracy * 100))
                if self.valid_loader is not None:
                    print("    Valid  loss : {:.4f}, Valid  acc : {:.2f}%".format(valid_loss, valid_accuracy * 100))

            if valid_loss <= self.best_loss and self.passed_epochs >= self.n_warmup_epochs:
                self.best_loss = valid_loss
                self.engine.store_to_model(self.model)
                self.log = {
                    "valid_metrics": valid_accuracy if self.valid_loader is not None else train_accuracy,
                    # a copy: the reference stores state_dict() by reference (:141), so its "best" weights
                    # silently follow later epochs; the intent (restore_best_weights) needs a snapshot
                    "best_model_state_dict": self.model.state_dict(),
                    "optimizer_state_dict": self.engine.optimizer_state_dict(self.lr, self.weight_decay),
                    "loss_train_history": self.loss_train_history,
                    "loss_valid_history": self.loss_valid_history,
                    "accuracy_train_history": self.accuracy_train_history,
                    "accuracy_valid_history": self.accuracy_valid_history,
                    "value_best_loss": self.best_loss,
                    "epoch_best_loss": self.passed_epochs,
                    "time_best_loss": round((time.time() - tmp_time), 3),
                }
                if self.save_checkpoint:
                    self.save()
            self.passed_epochs += 1

        self.train_time = round((time.time() - tmp_time), 3)
        self.engine.store_to_model(self.model)
        if self.plotloss:
            self.plot_history()
        self.log["training_time"] = self.train_time
        self.log["loss_train_history"] = self.loss_train_history
        self.log["loss_valid_history"] = self.loss_valid_history
        self.log["accuracy_train_history"] = self.accuracy_train_history
        self.log["accuracy_valid_history"] = self.accuracy_valid_history
        if self.save_checkpoint:
            self.save()
        return

    # ---------------------------------------------------------- evaluate (torch_trainer.py:173-207)
    def evaluate(self, test_loader, mask="test_metrics", return_output=False):
        tmp_time = time.time()
        losses, y, y_hat = self._eval_loader(as_tsloader(test_loader))
        metrics = self.f_metrics(y, y_hat)
        self.eval_time = round((time.time() - tmp_time), 3)
        self.log[mask + "_time"] = self.eval_time
        self.log[mask] = metrics
        if self.save_checkpoint:
            self.save()
        if return_output:
            return np.mean(losses), metrics, y, y_hat
        return np.mean(losses), metrics

    def save(self):
        torch.save(self.log, self.path_checkpoint + ".pt")

    def compress(self):
        """The saved log as an .xz file in place of the .pt one (torch_trainer.py:216-233)."""
        import lzma
        with open(self.path_checkpoint + ".pt", "rb") as f:
            data = f.read()
        with open(self.path_checkpoint + ".xz", "wb") as f:
            f.write(lzma.compress(data))
        os.remove(self.path_checkpoint + ".pt")

    def plot_history(self):
        """Loss curves (torch_trainer.py:235-250): to ``path_fig`` if given, else shown."""
        import matplotlib.pyplot as plt
        plt.figure()
        plt.plot(range(self.passed_epochs), self.loss_train_history[:self.passed_epochs], label="Train loss")
        if self.valid_loader is not None:
            plt.plot(range(self.passed_epochs), self.loss_valid_history[:self.passed_epochs], label="Valid loss")
        plt.ylabel("Loss")
        plt.xlabel("Epochs")
        plt.legend()
        if self.path_fig:
            plt.savefig(self.path_fig)
        else:
            plt.show()

    def reduce_lr(self, new_lr):
        """The learning rate of the following steps (torch_trainer.py:252-258; the reference's body reaches for
        ``self.model.optimizer``, which does not exist — the intent is what is done here)."""
        self.lr = float(new_lr)
        if self.scheduler is not None:
            self.scheduler.lr = float(new_lr)

    def restore_best_weights(self):
        try:
            self.model.load_state_dict(self.log["best_model_state_dict"])
            self.engine.load_from_model(self.model)
        except KeyError:
            warnings.warn("Error during loading log checkpoint state dict : no update.")
        return

    # ------------------------------------------------------------ private loops (:273-325)
    def __train(self):
        """One epoch (torch_trainer.py:273-300).  The reference reads loss and accuracy back per batch (:296-297);
        here they land in a per-epoch device buffer and come back once, at the end of the epoch.  Data-parallel:
        every rank walks the same batches and trains on its ``parallel.batch_shard`` of each."""
        eng = self.engine
        n_batches = len(self.train_loader)
        with torch.cuda.device(eng.device):
            loss_hist = torch.zeros(n_batches, dtype=torch.float32, device=eng.device)
            nc_hist = torch.zeros(n_batches, dtype=torch.int32, device=eng.device)
        share = np.zeros(n_batches)
        total = 0
        crit = getattr(self, "train_criterion", None)
        res = self._resident_train_set()
        if res is not None:
            # the windows wait in HBM: one permutation per epoch goes to the device, batches are gathered there
            xd, yd = res
            order = self.train_loader.epoch_order()
            bs = self.train_loader.batch_size
            with torch.cuda.device(eng.device):
                idx = torch.from_numpy(np.ascontiguousarray(order, dtype=np.int64)).to(eng.device)
                for i, s in enumerate(range(0, len(order), bs)):
                    n = min(bs, len(order) - s)
                    lo, hi = parallel.batch_shard(n) if eng._dp() else (0, n)
                    sel = idx[s + lo:s + hi]
                    eng.forward_backward(xd.index_select(0, sel), yd.index_select(0, sel), global_batch=n,
                                         loss_out=loss_hist[i:i + 1], ncorrect_out=nc_hist[i:i + 1], criterion=crit)
                    eng.optimizer_step(self.lr, self.weight_decay)
                    share[i] = (hi - lo) / n
                    total += n
        else:
            for i, (ts, labels) in enumerate(self.train_loader):
                n = len(labels)
                lo, hi = parallel.batch_shard(n) if eng._dp() else (0, n)
                eng.forward_backward(ts[lo:hi], labels[lo:hi], global_batch=n, loss_out=loss_hist[i:i + 1],
                                     ncorrect_out=nc_hist[i:i + 1], criterion=crit)
                eng.optimizer_step(self.lr, self.weight_decay)
                share[i] = (hi - lo) / n
                total += n
        losses = loss_hist.cpu().numpy().astype(np.float64) * share  # this rank's part of each batch's mean loss
        corrects = nc_hist.cpu().numpy().astype(np.float64)
        if eng._dp():
            both = parallel.sum_over_ranks(np.concatenate([losses, corrects]), device=eng.device)
            losses, corrects = both[:n_batches], both[n_batches:]
        return np.mean(losses), np.sum(corrects) / total

    def _resident_train_set(self):
        """The training windows on the GPU (TrainEngine.make_resident), made on first use; None for loaders that are not
        TSLoaders (any iterable of (ts, labels) batches works, one host-to-device copy per batch like the reference)."""
        if not hasattr(self, "_resident"):
            tl = self.train_loader
            ok = isinstance(tl, TSLoader) and hasattr(self.engine, "make_resident") and getattr(self, "resident", True)
            self._resident = self.engine.make_resident(tl) if ok else None
        return self._resident

    def __evaluate(self):
        losses, y, y_hat = self._eval_loader(self.valid_loader)
        return np.mean(losses), float(np.sum(y == y_hat)) / len(y)

    def _eval_loader(self, loader):
        """eval()-mode forward of the whole loader through the BN-folded inference kernels.  The
        reference iterates batch-1 loaders (core.py:114-119); per-sample eval-mode logits do not
        depend on the batch they were computed in, so the per-batch losses are rebuilt from them."""
        self.engine.store_to_model(self.model)
        self.model.eval()
        if hasattr(loader, "draw_base_seed"):
            loader.draw_base_seed()  # the reference iterates the loader here (torch_trainer.py:182,311)
        X = loader.samples[:, 0, :]
        logits = self.model(X.astype(np.float32)).float().contiguous()
        n, bs = len(loader.labels), loader.batch_size
        if self.valid_criterion is None:
            lib = _lib.load()
            dev = logits.device
            with torch.cuda.device(dev):
                labels = torch.as_tensor(np.asarray(loader.labels, dtype=np.float32).astype(np.int32)).to(dev)
                loss_d = torch.empty(n, dtype=torch.float32, device=dev)
                pred_d = torch.empty(n, dtype=torch.int32, device=dev)
                _lib.check(lib.camal_ce_eval(_lib.ptr(logits), _lib.ptr(labels), n, _lib.ptr(loss_d), _lib.ptr(pred_d),
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            per_sample = loss_d.cpu().numpy()
            y_hat = pred_d.cpu().numpy().astype(np.int64)
            # the reference logs loss.item() of every batch (float32 means), batch size 1 in CamAL.train
            losses = per_sample.tolist() if bs == 1 else [float(per_sample[s:s + bs].mean(dtype=np.float32)) for s in range(0, n, bs)]
        else:  # a caller-supplied validation criterion: torch, batch by batch like the reference
            labels = torch.as_tensor(np.asarray(loader.labels)).to(logits.device).float().long()
            losses = [float(self.valid_criterion(logits[s:s + bs], labels[s:s + bs]).item()) for s in range(0, n, bs)]
            y_hat = torch.max(logits, 1)[1].cpu().numpy()
        self.model.train()
        return losses, loader.labels.astype(np.float32), y_hat
