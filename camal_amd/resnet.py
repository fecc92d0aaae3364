"""``CamALResNet`` — host-side mirror of the reference classifier
(``src/models/camal/classifiers/camal_resnet.py:4-23``): same constructor, same
``state_dict()`` key names and shapes, ``forward(x[B,1,L]) -> logits[B,2]``.

It is not an ``nn.Module``: parameters live in a plain ordered dict and all arithmetic runs in
libcamal_hip.so.  ``eval()`` mode uses the BN-folded inference kernels
(``camal_forward_cam``); ``train()`` mode is driven by ``camal_amd.trainer``.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from . import weights


class _BlockHandle:
    """What ``resnet._modules['layers'][i]`` returns: names a ResNetBlock (used by ``CAM``
    as ``last_conv_layer``, reference core.py:403)."""

    def __init__(self, owner, index):
        self.owner, self.index = owner, index


class _LinearHandle:
    """What ``resnet._modules['linear']`` returns (``fc_layer_name`` of ``CAM``, core.py:404)."""

    def __init__(self, owner):
        self.owner = owner

    def parameters(self):
        sd = self.owner._params
        return [torch.from_numpy(sd["linear.weight"]), torch.from_numpy(sd["linear.bias"])]


class CamALResNet:
    _arch = weights.ARCH_CAMAL

    def __init__(self, kernel_size: int = 7, in_channels: int = 1, mid_channels: int = 64,
                 norm: str = "BatchNorm", nb_class: int = 2):
        if kernel_size % 2 != 1:
            raise ValueError("kernel_size must be odd (replicate 'same' padding, camal_resnet.py:83)")
        self.kernel_size, self.in_channels, self.mid_channels = kernel_size, in_channels, mid_channels
        self.norm, self.nb_class = norm, nb_class  # `norm` is dead in the reference too (quirk 1)
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.training = True
        self._engine = None
        self._params = OrderedDict()
        # torch's default initialisers (Conv1d / Linear: U(-1/sqrt(fan_in), 1/sqrt(fan_in)); BN: 1, 0)
        spec = weights.state_dict_spec(kernel_size, in_channels, mid_channels, nb_class, self._arch)
        for key, (shape, dt) in spec.items():
            if key.endswith("num_batches_tracked"):
                v = np.array(0, np.int64)
            elif key.endswith("running_mean"):
                v = np.zeros(shape, np.float32)
            elif key.endswith("running_var"):
                v = np.ones(shape, np.float32)
            elif ".layers.1." in key or ".residual.1." in key:
                v = np.ones(shape, np.float32) if key.endswith("weight") else np.zeros(shape, np.float32)
            else:
                wkey = key[:-4] + "weight" if key.endswith("bias") else key
                wshape = spec[wkey][0]
                fan_in = int(np.prod(wshape[1:]))
                bound = 1.0 / np.sqrt(fan_in)
                v = torch.empty(shape).uniform_(-bound, bound).numpy().astype(np.float32)
            self._params[key] = v
        self._modules = OrderedDict(
            layers=[_BlockHandle(self, 0), _BlockHandle(self, 1), _BlockHandle(self, 2)],
            linear=_LinearHandle(self))

    # ---- nn.Module-like surface ---------------------------------------------------------
    def state_dict(self):
        return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in self._params.items())

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = weights.to_numpy_state_dict(state_dict)
        missing = [k for k in self._params if k not in sd]
        unexpected = [k for k in sd if k not in self._params]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: missing {missing}, unexpected {unexpected}")
        for k in self._params:
            if k in sd:
                v = np.asarray(sd[k])
                if tuple(v.shape) != tuple(self._params[k].shape):
                    raise RuntimeError(f"size mismatch for {k}: {v.shape} vs {self._params[k].shape}")
                self._params[k] = v.astype(self._params[k].dtype).copy()
        self._drop_engine()
        return self

    def named_parameters(self):
        for k in self._params:
            if not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                yield k, torch.from_numpy(self._params[k])

    def parameters(self):
        return [p for _, p in self.named_parameters()]

    def to(self, device):
        d = torch.device(device)
        if d != self.device:
            self.device = d
            self._drop_engine()
        return self

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        self.training = mode
        return self

    # ---- forward --------------------------------------------------------------------------
    def _drop_engine(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def engine(self):
        from .engine import Ensemble
        if self._engine is None:
            self._engine = Ensemble([self._params], device=self.device, precision="fp32")
        return self._engine

    def forward(self, x):
        """x [B,1,L] (or [B,L]) -> logits [B,2] on the device (camal_resnet.py:21-23)."""
        if self.training:
            from .trainer import forward_train
            return forward_train(self, x)
        logits, _ = self.engine().forward_cam(x, want_cam=False)
        return logits[0]

    __call__ = forward


class ResNet(CamALResNet):
    """Mirror of the reference's second classifier (``src/models/camal/classifiers/resnet.py:3-23``): the same three
    ResNetBlocks, state-dict key names and head, with kernel sizes 7, 5, 3 in every block (:31) and zero ``"same"``
    padding (:62) instead of the replicate mode.  Inference (``eval()``), CAM, the ensemble post-processing and the
    training step run on it through the same entry points (``camal_create_arch`` / ``camal_trainer_create_arch`` with
    ``CAMAL_ARCH_RESNET``)."""
    _arch = weights.ARCH_RESNET

    def __init__(self, in_channels: int = 1, mid_channels: int = 64, nb_class: int = 2):
        super().__init__(kernel_size=7, in_channels=in_channels, mid_channels=mid_channels, nb_class=nb_class)
        self.input_args = {"in_channels": in_channels, "nb_class": nb_class}  # resnet.py:8-11
