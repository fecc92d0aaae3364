"""Randomised sweep of the training step (tools/fuzz_train.py) as a GPU test: 30 drawn cases — first kernel sizes 1..31,
mid_channels 32 / 64 / 128, the classifiers/resnet.py variant, batches of 1..48 windows (some on a trainer created for
more), lengths on and next to the kernels' tile boundaries — every gradient tensor within 1e-3 of the float64 autograd
step taken on the ReLU masks the HIP forward pass took (see the tool's header for why the masks are imposed).  The
committed profiles/r03_fuzz_train.json is the same sweep over 300 cases."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_training_cases_against_float64_autograd():
    import fuzz_train as fz
    rng = np.random.default_rng(2027)
    failing = []
    for i in range(30):
        c = fz.draw(rng)
        bad = fz.check(c, fz.run_case(c))
        if bad:
            failing.append((i, c, bad))
    assert not failing, failing


def test_wide_model_large_first_kernels_train():
    """mid_channels = 128 with first kernel sizes from 19 up: the data gradient of block 2's first ConvBlock (256 -> 128
    channels) takes one four-wave workgroup per CU when the eight-wave tile does not fit the LDS (it used to fail with
    'kernel size too large for the LDS tile': CamAL(mid_channels=128).train with the default kernel list hit it at 25)."""
    import fuzz_train as fz
    for k in (19, 25, 49):
        c = {"arch": "camal", "mid": 128, "k": k, "L": 150, "B": 5, "max_batch": 5, "seed": 11 + k}
        assert not fz.check(c, fz.run_case(c)), k


def test_random_trainer_runs_stay_with_the_cpu_trajectory():
    """tools/fuzz_trainer.py: epochs of AdamW steps through BasedClassifTrainer (ragged last batches down to one window,
    resident and per-batch-copy paths, weight decay) against torch autograd + AdamW on the CPU; thresholds in the tool."""
    import fuzz_trainer as ft
    rng = np.random.default_rng(2030)
    failing = []
    for i in range(24):
        c = ft.draw(rng)
        bad = ft.check(c, ft.run_case(c))
        if bad:
            failing.append((i, c, bad))
    assert not failing, failing
