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


def rel_err(a, b):
    """max|a-b| / max|b| per row (last axis), worst row — the 'relative' of north_star:
    element-wise relative error is meaningless at the zero crossings of a CAM."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    num = np.abs(a - b).max(axis=-1)
    den = np.maximum(np.abs(b).max(axis=-1), 1e-12)
    return float((num / den).max())
