"""Head calibration for synthetic weights (bench / smoke / tests), through the HIP path.

Random-init members give p ≈ 0.504 on every window and all-zero states, which
would let the post-processing skip its work.  ``calibrated_state_dicts`` rescales
``linear.weight`` and sets ``linear.bias`` so that the logit difference over a sample of
windows has median 0 and the requested spread.  The forward used for it is the product's
own (camal_forward_cam), never the oracle.
"""
from __future__ import annotations

import numpy as np

from . import weights
from .engine import Ensemble


def calibrated_state_dicts(kernel_sizes, seed, sample_x, device="cuda:0", spread=2.0, precision="fp32"):
    base = [weights.make_state_dict(k, seed=seed, head_gain=1.0) for k in kernel_sizes]
    ens = Ensemble(base, device=device, precision=precision)
    logits, _ = ens.forward_cam(sample_x, want_cam=False)
    logits = logits.double().cpu().numpy()
    ens.close()
    out = []
    for e, k in enumerate(kernel_sizes):
        d = logits[e, :, 1] - logits[e, :, 0]
        g = float(np.float32(spread / max(d.std(), 1e-6)))
        b = float(np.float32(-g * np.median(d)))
        out.append(weights.make_state_dict(k, seed=seed, head_gain=g, head_bias=(0.0, b)))
    return out
