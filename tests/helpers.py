"""Shared helpers for the parity tests (fixtures are outputs of the reference itself,
see tests/golden/make_golden.py)."""
import glob
import os

import numpy as np

from camal_amd import weights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star tolerances
REL_TOL_FP32 = 1e-3      # probabilities and CAMs, fp32 path
STATE_AGREE_MIN = 0.999  # fraction of timestamps with identical on/off state


def golden_cases(prefix=""):
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def case_state_dicts(case):
    return [weights.make_state_dict(int(k), seed=int(case["seed"]), head_gain=float(g), head_bias=tuple(b))
            for k, g, b in zip(case["members"], case["head_gain"], case["head_bias"])]


def width_state_dicts(case):
    """Members of a ``width*`` fixture: CamALResNet(kernel_size, mid_channels=mid)."""
    return [weights.make_state_dict(int(k), seed=int(case["seed"]), mid_channels=int(case["mid"]), head_gain=float(g),
                                    head_bias=tuple(b))
            for k, g, b in zip(case["members"], case["head_gain"], case["head_bias"])]


def resnetvar_state_dicts(case):
    """Members of a ``resnetvar_*`` fixture: classifiers/resnet.py's ResNet (kernels 7/5/3, zero padding)."""
    return [weights.make_state_dict(7, seed=int(sd), arch=weights.ARCH_RESNET, head_gain=float(g), head_bias=tuple(b))
            for sd, g, b in zip(case["seeds"], case["head_gain"], case["head_bias"])]


def rel_err(a, b):
    """max|a-b| / max|b| per row (last axis), worst row — the 'relative' of north_star:
    element-wise relative error is meaningless at the zero crossings of a CAM."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    num = np.abs(a - b).max(axis=-1)
    den = np.maximum(np.abs(b).max(axis=-1), 1e-12)
    return float((num / den).max())


# ---- training fixtures (tests/golden/train_*.npz, make_golden.py::train_case) ----------------
def train_case_inputs(case):
    from camal_amd import synth
    k, seed = int(case["k"]), int(case["seed"])
    x, y = synth.ideal_windows(int(case["n"]), int(case["length"]), seed=seed)
    arch = weights.ARCH_RESNET if "resnet_variant" in case and bool(case["resnet_variant"]) else weights.ARCH_CAMAL
    mid = int(case["mid"]) if "mid" in case else 64
    sd0 = weights.make_state_dict(k, seed=seed, head_gain=1.0, arch=arch, mid_channels=mid)
    return k, x, y, sd0


def model_for(sd0):
    """The mirror class a state dict belongs to (CamALResNet, or ResNet for the 7/5/3 zero-padded variant)."""
    from camal_amd.resnet import CamALResNet, ResNet
    mid = int(np.asarray(sd0["layers.0.layers.0.layers.0.weight"]).shape[0])
    if weights.arch_of(sd0) == weights.ARCH_RESNET:
        return ResNet(mid_channels=mid)
    return CamALResNet(kernel_size=weights.kernel_size_of(sd0), mid_channels=mid)


def is_bn_fed_bias(key):
    """Conv biases that feed a train-mode BatchNorm: exact gradient 0, torch yields rounding noise."""
    return key.endswith("layers.0.bias") or key.endswith("residual.0.bias")


def pack_train_views(keys, grads):
    """Same views of a gradient dict as the fixture stores."""
    norms = np.array([np.linalg.norm(np.asarray(grads[k], np.float64)) for k in keys])
    small = np.concatenate([np.asarray(grads[k]).ravel() for k in keys
                            if np.asarray(grads[k]).size <= 2048 and not is_bn_fed_bias(k)])
    big = np.concatenate([np.asarray(grads[k]).ravel()[::97] for k in keys if np.asarray(grads[k]).size > 2048])
    return norms, small, big


# ---- scripted control-flow scenarios of BasedClassifTrainer.train (tests/golden/trainer_control_flow.npz) -------
def scripted_trainer_scenarios():
    """Per-epoch (train loss, train acc, valid loss, valid acc) scripts + trainer settings, from fixed seeds: fed to
    the REFERENCE trainer by make_golden.py (its private epoch methods replaced by the script) and to this repo's
    trainer by tests/test_abi_and_host.py, so that scheduler / early stop / warm-up / snapshot logic is compared
    decision by decision without any arithmetic noise in between."""
    rng = np.random.default_rng(77)
    out = []
    for i in range(16):
        n_epochs = int(rng.integers(6, 25))
        base = np.abs(np.cumsum(rng.normal(-0.03, 0.08, size=n_epochs))) + 0.1
        valid = np.round(base + rng.normal(0, 0.02, size=n_epochs), int(rng.integers(2, 5)))
        if i % 3 == 0:  # a long plateau: LR drops, then early stop
            valid[n_epochs // 3:] = valid[n_epochs // 3] + 0.01
        train = np.round(base * 0.9, 4)
        out.append(dict(
            n_epochs=n_epochs, train_loss=train, valid_loss=np.abs(valid) + 0.01,
            train_acc=np.round(rng.uniform(0.5, 1.0, n_epochs), 3), valid_acc=np.round(rng.uniform(0.5, 1.0, n_epochs), 3),
            has_valid=bool(i % 4 != 3), patience_es=[None, 1, 2, 5][i % 4] if i % 5 else 3,
            patience_rlr=[None, 0, 1, 3][(i // 2) % 4], n_warmup_epochs=int([0, 1, 2, 4][(i // 3) % 4]),
            lr=float([1e-3, 3e-3, 1e-4][i % 3])))
    return out


def trainer_valid_inputs(seed, n_train, n_valid, length, flip=0.3):
    """Windows and labels of the ``trainer_valid_*`` fixture: IDEAL-shaped windows with a share of the labels flipped
    (so that the validation loss turns upwards and the scheduler / early stopper have something to decide on)."""
    from camal_amd import synth
    x, y = synth.ideal_windows(n_train + n_valid, length, seed=seed)
    noise = np.random.default_rng(seed + 1000).random(len(y)) < flip
    y = np.where(noise, 1 - y, y)
    return x[:n_train], y[:n_train], x[n_train:], y[n_train:]
