"""State-dict schema of ``CamALResNet`` and a seeded NumPy weight generator.

The key names / shapes are the reference's (``camal_resnet.py:13-19``; 79 entries for
the default widths) so that reference checkpoints load unchanged and
``camal_logs`` pickles stay exchangeable.

``make_state_dict`` is the deterministic generator used by tests, fixtures and the bench
(the same code runs in the dev container and on the GPU box, so no weight files are
committed: fixtures only store the expected outputs).
"""
from __future__ import annotations

import numpy as np

INNER_KERNELS = (None, 7, 3)  # camal_resnet.py:31 — only the first conv uses `kernel_size`
RESNET_KERNELS = (7, 5, 3)    # classifiers/resnet.py:31 — the fixed-kernel variant (zero "same" padding, :62)
ARCH_CAMAL, ARCH_RESNET = "camal", "resnet"


def block_channels(in_channels=1, mid_channels=64):
    """(Cin, Cout) of the three ResNetBlocks — camal_resnet.py:14-16."""
    return [(in_channels, mid_channels), (mid_channels, 2 * mid_channels), (2 * mid_channels, 2 * mid_channels)]


def conv_layers(kernel_size, in_channels=1, mid_channels=64, arch=ARCH_CAMAL):
    """Flat list of the 9 ConvBlocks as (key_prefix, Cin, Cout, k)."""
    out = []
    for b, (cin, cout) in enumerate(block_channels(in_channels, mid_channels)):
        chans = [cin, cout, cout, cout]
        for i in range(3):
            if arch == ARCH_RESNET:
                k = RESNET_KERNELS[i]
            else:
                k = kernel_size if i == 0 else INNER_KERNELS[i]
            out.append((f"layers.{b}.layers.{i}", chans[i], chans[i + 1], k))
    return out


def residual_layers(in_channels=1, mid_channels=64):
    """Blocks that have a BN(Conv1x1) skip (Cin != Cout) — camal_resnet.py:41,51-54."""
    return [(f"layers.{b}.residual", cin, cout)
            for b, (cin, cout) in enumerate(block_channels(in_channels, mid_channels)) if cin != cout]


def state_dict_spec(kernel_size, in_channels=1, mid_channels=64, nb_class=2, arch=ARCH_CAMAL):
    """Ordered {key: (shape, dtype)} exactly as ``CamALResNet.state_dict()`` yields it
    (``arch="resnet"``: ``classifiers/resnet.py``'s ``ResNet``, same names, kernels 7/5/3)."""
    spec = {}

    def bn(prefix, c):
        spec[prefix + ".weight"] = ((c,), np.float32)
        spec[prefix + ".bias"] = ((c,), np.float32)
        spec[prefix + ".running_mean"] = ((c,), np.float32)
        spec[prefix + ".running_var"] = ((c,), np.float32)
        spec[prefix + ".num_batches_tracked"] = ((), np.int64)

    convs = conv_layers(kernel_size, in_channels, mid_channels, arch)
    res = {p: (ci, co) for p, ci, co in residual_layers(in_channels, mid_channels)}
    for b in range(3):
        for i in range(3):
            p, ci, co, k = convs[3 * b + i]
            spec[p + ".layers.0.weight"] = ((co, ci, k), np.float32)
            spec[p + ".layers.0.bias"] = ((co,), np.float32)
            bn(p + ".layers.1", co)
        rp = f"layers.{b}.residual"
        if rp in res:
            ci, co = res[rp]
            spec[rp + ".0.weight"] = ((co, ci, 1), np.float32)
            spec[rp + ".0.bias"] = ((co,), np.float32)
            bn(rp + ".1", co)
    spec["linear.weight"] = ((nb_class, 2 * mid_channels), np.float32)
    spec["linear.bias"] = ((nb_class,), np.float32)
    return spec


def trainable_keys(kernel_size, in_channels=1, mid_channels=64, arch=ARCH_CAMAL):
    """Keys of ``model.parameters()`` in torch's registration order (what AdamW sees)."""
    return [k for k in state_dict_spec(kernel_size, in_channels, mid_channels, arch=arch)
            if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]


def count_parameters(kernel_size, in_channels=1, mid_channels=64):
    spec = state_dict_spec(kernel_size, in_channels, mid_channels)
    return int(sum(np.prod(spec[k][0], dtype=np.int64) for k in trainable_keys(kernel_size, in_channels, mid_channels)))


def make_state_dict(kernel_size, seed=0, in_channels=1, mid_channels=64, nb_class=2,
                    head_gain=1.0, head_bias=(0.0, 0.0), randomize_bn=True, head_aligned=True, arch=ARCH_CAMAL):
    """Seeded NumPy state dict (He-scaled convs, randomised BN statistics).

    ``head_gain`` scales ``linear.weight`` and ``head_bias`` sets ``linear.bias`` — the
    fixture generator calibrates them so that detection probabilities spread around 0.5
    (with default random init every window gives p≈0.504 and all-zero states).
    ``head_aligned`` draws ``W_fc[1] = |N|`` and ``W_fc[0] = -W_fc[1]`` so that the class-1
    CAM is non-negative and grows with the (post-ReLU) feature energy, i.e. members agree
    on high-power windows and CAM peaks sit on the pulses, like a trained detector.
    """
    rng = np.random.default_rng([seed, kernel_size, in_channels, mid_channels] + ([1] if arch == ARCH_RESNET else []))
    sd = {}
    for key, (shape, dt) in state_dict_spec(kernel_size, in_channels, mid_channels, nb_class, arch).items():
        if key.endswith("num_batches_tracked"):
            sd[key] = np.array(100 if randomize_bn else 0, dtype=np.int64)
        elif key.endswith("running_mean"):
            sd[key] = (rng.normal(0.0, 0.1, shape) if randomize_bn else np.zeros(shape)).astype(np.float32)
        elif key.endswith("running_var"):
            sd[key] = (rng.uniform(0.5, 1.5, shape) if randomize_bn else np.ones(shape)).astype(np.float32)
        elif ".layers.1." in key or ".residual.1." in key:  # BN affine
            if key.endswith("weight"):
                sd[key] = (rng.uniform(0.8, 1.2, shape) if randomize_bn else np.ones(shape)).astype(np.float32)
            else:
                sd[key] = (rng.normal(0.0, 0.1, shape) if randomize_bn else np.zeros(shape)).astype(np.float32)
        elif key == "linear.weight":
            w = rng.normal(0.0, 1.0 / np.sqrt(shape[1]), shape)
            if head_aligned and shape[0] == 2:
                w[1] = np.abs(w[1])
                w[0] = -w[1]
            sd[key] = (head_gain * w).astype(np.float32)
        elif key == "linear.bias":
            sd[key] = np.asarray(head_bias, dtype=np.float32).reshape(shape)
        elif key.endswith("weight"):  # conv weight [Cout, Cin, k]
            fan_in = shape[1] * shape[2]
            sd[key] = rng.normal(0.0, np.sqrt(2.0 / fan_in), shape).astype(np.float32)
        else:  # conv bias
            sd[key] = rng.normal(0.0, 0.05, shape).astype(np.float32)
    return sd


def fold_conv_bn(w, b, gamma, beta, mean, var, eps=1e-5):
    """Fold eval-mode BatchNorm into the preceding conv: y = conv(x; w', b').

    w' = w * gamma/sqrt(var+eps);  b' = (b - mean) * gamma/sqrt(var+eps) + beta.
    Computed in float64, returned as float32.
    """
    s = np.asarray(gamma, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps)
    w2 = np.asarray(w, np.float64) * s[:, None, None]
    b2 = (np.asarray(b, np.float64) - np.asarray(mean, np.float64)) * s + np.asarray(beta, np.float64)
    return w2.astype(np.float32), b2.astype(np.float32)


def to_numpy_state_dict(sd):
    """Accept a torch or NumPy state dict (also DataParallel's ``module.`` prefix)."""
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def kernel_size_of(sd):
    return int(np.asarray(sd["layers.0.layers.0.layers.0.weight"]).shape[2])


def arch_of(sd):
    """Which classifier a state dict belongs to: the two share every key name; the second ConvBlock of a block has
    kernel 7 in ``CamALResNet`` (camal_resnet.py:31) and 5 in ``ResNet`` (classifiers/resnet.py:31)."""
    k2 = int(np.asarray(sd["layers.0.layers.1.layers.0.weight"]).shape[2])
    if k2 == 7:
        return ARCH_CAMAL
    if k2 == 5 and kernel_size_of(sd) == 7:
        return ARCH_RESNET
    raise ValueError(f"state dict is neither CamALResNet's nor ResNet's (second kernel {k2})")
