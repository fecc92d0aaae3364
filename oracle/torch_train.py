"""CPU restatement (PyTorch fp32 autograd) of one CamAL training step.

TEST INFRASTRUCTURE ONLY (see oracle/camal_oracle.py's header).  The reference's training
arithmetic is torch's own: ``CamALResNet`` in ``train()`` mode, ``nn.CrossEntropyLoss()``,
``loss.backward()``, ``optim.AdamW.step()`` (src/helpers/torch_trainer.py:35,281-292).  This module
restates the model as ``torch.nn.functional`` calls on leaf tensors keyed by the reference's
state-dict names, so gradients / updates can be compared tensor by tensor with the HIP path.
Pinned against the reference's own classes by tests/golden/train_*.npz
(tests/golden/make_golden.py::train_case).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from camal_amd import weights


class FunctionalResNet:
    def __init__(self, sd, dtype=torch.float32, relu_masks=None):
        """dtype=torch.float64: the same arithmetic without float32 rounding (what both float32 implementations
        approximate), for tests whose tolerance is close to the rounding noise of small batches.
        relu_masks: {(block, conv): bool array [B, C, L]} — the ReLU of that ConvBlock becomes a multiplication by the
        given 0/1 mask (the mask another implementation's forward pass took): the step then differentiates exactly
        the piecewise-linear function that implementation evaluated, whichever side of zero float32 rounding put its
        borderline pre-activations on."""
        self.dtype = dtype
        self.relu_masks = relu_masks or {}
        sd = weights.to_numpy_state_dict(sd)
        self.k = weights.kernel_size_of(sd)
        self.arch = weights.arch_of(sd)  # "resnet": classifiers/resnet.py's ResNet, zero "same" padding (:62)
        self.mid = int(np.asarray(sd["layers.0.layers.0.layers.0.weight"]).shape[0])  # mid_channels (camal_resnet.py:8)
        self.keys = weights.trainable_keys(self.k, mid_channels=self.mid, arch=self.arch)
        self.p = {k: torch.tensor(np.array(sd[k]), dtype=dtype, requires_grad=k in self.keys)
                  for k in sd if not k.endswith("num_batches_tracked")}
        self.nbt = int(sd["layers.0.layers.0.layers.1.num_batches_tracked"])

    def parameters(self):
        return [self.p[k] for k in self.keys]

    def _bn(self, z, prefix, training):
        return F.batch_norm(z, self.p[prefix + ".running_mean"], self.p[prefix + ".running_var"],
                            self.p[prefix + ".weight"], self.p[prefix + ".bias"], training=training,
                            momentum=0.1, eps=1e-5)

    def forward(self, x, training=True):
        p = self.p
        if x.dim() == 2:
            x = x[:, None, :]
        for b in range(3):
            h = x
            for i in range(3):
                pre = f"layers.{b}.layers.{i}"
                w = p[pre + ".layers.0.weight"]
                k = w.shape[-1]
                hp = F.pad(h, (k // 2, k // 2)) if self.arch == weights.ARCH_RESNET else F.pad(h, (k // 2, k // 2), mode="replicate")
                h = F.conv1d(hp, w, p[pre + ".layers.0.bias"])
                o = self._bn(h, pre + ".layers.1", training)
                if (b, i) in self.relu_masks:
                    h = o * torch.as_tensor(np.asarray(self.relu_masks[(b, i)]), dtype=self.dtype)
                else:
                    h = F.relu(o)
            if f"layers.{b}.residual.0.weight" in p:
                r = F.conv1d(x, p[f"layers.{b}.residual.0.weight"], p[f"layers.{b}.residual.0.bias"])
                h = h + self._bn(r, f"layers.{b}.residual.1", training)
            x = h
        return F.linear(x.mean(dim=-1), p["linear.weight"], p["linear.bias"])

    def state_dict(self):
        out = {k: v.detach().numpy().copy() for k, v in self.p.items()}
        for k in list(out):
            if k.endswith("running_mean"):
                out[k[:-len("running_mean")] + "num_batches_tracked"] = np.array(self.nbt, np.int64)
        return out


def loss_and_grads(sd, x, y, criterion=None, dtype=torch.float32, relu_masks=None):
    """One train-mode forward/backward -> (loss, ncorrect, {key: grad}, updated state dict).  criterion: the trainer's
    `criterion` argument (torch_trainer.py:20,290); None = nn.CrossEntropyLoss().  dtype=torch.float64: the same step
    without float32 rounding (see FunctionalResNet), to tell an implementation's error from the reference's own noise."""
    m = FunctionalResNet(sd, dtype=dtype, relu_masks=relu_masks)
    logits = m.forward(torch.as_tensor(np.asarray(x, np.float32)).to(dtype))
    labels = torch.as_tensor(np.asarray(y)).float().long()
    if dtype == torch.float32:
        logits_l = logits.float()
    else:
        logits_l = logits
    loss = F.cross_entropy(logits_l, labels) if criterion is None else criterion(logits_l, labels)
    loss.backward()
    m.nbt += 1
    grads = {k: m.p[k].grad.numpy().copy() for k in m.keys}
    ncorrect = int((torch.max(logits, 1)[1] == labels).sum())
    return float(loss.detach()), ncorrect, grads, m.state_dict()


def train_steps(sd, batches, lr=1e-3, weight_decay=0.0):
    """AdamW over a list of (x, y) batches -> (losses, final state dict)."""
    m = FunctionalResNet(sd)
    opt = torch.optim.AdamW(m.parameters(), lr=lr, weight_decay=weight_decay)
    losses = []
    for x, y in batches:
        opt.zero_grad()
        logits = m.forward(torch.as_tensor(np.asarray(x, np.float32)))
        loss = F.cross_entropy(logits.float(), torch.as_tensor(np.asarray(y)).float().long())
        loss.backward()
        opt.step()
        m.nbt += 1
        losses.append(float(loss.detach()))
    return losses, m.state_dict()


def _dp_chunks(n, world):
    """nn.DataParallel's scatter: chunks of ceil(n / world) in device order (the last may be short or missing)."""
    chunk = -(-n // world)
    return [(lo, min(n, lo + chunk)) for lo in range(0, n, chunk)]


def dp_train_steps(sd, batches, world, lr=1e-3, weight_decay=0.0, dtype=torch.float32):
    """The reference's ``all_gpu=True`` step (single-process ``nn.DataParallel``, torch_trainer.py:82-90,281-292)
    restated on CPU: the batch is cut in ``world`` chunks, every replica runs the train-mode forward on its own chunk
    (BatchNorm batch statistics per replica), the logits are gathered, ONE mean cross-entropy over the whole batch,
    backward, AdamW.  Only replica 0's running statistics persist (the others' buffers are copies that die with
    the step).  -> (losses, [first step's gradients {key: array}], final state dict)."""
    m = FunctionalResNet(sd, dtype=dtype)
    opt = torch.optim.AdamW(m.parameters(), lr=lr, weight_decay=weight_decay)
    run_keys = [k for k in m.p if k.endswith("running_mean") or k.endswith("running_var")]
    losses, first_grads = [], None
    for x, y in batches:
        x = torch.as_tensor(np.asarray(x, np.float32)).to(dtype)
        labels = torch.as_tensor(np.asarray(y)).float().long()
        opt.zero_grad()
        chunks = _dp_chunks(len(x), world)
        start = {k: m.p[k].clone() for k in run_keys}
        outs = [None] * len(chunks)
        for r in reversed(range(len(chunks))):  # replica 0 last: its running-statistics update is the one that stays
            for k in run_keys:  # a fresh buffer per replica (autograd keeps the ones a forward touched)
                m.p[k] = start[k].clone()
            lo, hi = chunks[r]
            outs[r] = m.forward(x[lo:hi])
        loss = F.cross_entropy(torch.cat(outs, 0), labels)
        loss.backward()
        if first_grads is None:
            first_grads = {k: m.p[k].grad.numpy().copy() for k in m.keys}
        opt.step()
        m.nbt += 1
        losses.append(float(loss.detach()))
    return losses, first_grads, m.state_dict()
