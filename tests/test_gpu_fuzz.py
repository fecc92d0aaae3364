"""Randomised parity sweep (tools/fuzz_parity.py) as a GPU test: 40 drawn cases — member sets (first kernel sizes 1..41,
mid_channels 32 / 64 / 128, the classifiers/resnet.py variant), window lengths on and next to the kernels' tile
boundaries, chunking, precision, moving-average width, gate, float64 and signed data — against the CPU oracle with
the tolerances of tests/test_gpu_parity.py.  The committed profiles/r03_fuzz_parity.json is the same sweep over 300
cases."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_cases_against_the_oracle():
    import fuzz_parity as fz
    rng = np.random.default_rng(2026)
    failing = []
    for i in range(40):
        c = fz.draw(rng)
        bad = fz.check(c, fz.run_case(c))
        if bad:
            failing.append((i, c, bad))
    assert not failing, failing


def test_random_postprocessing_cases_exact():
    """tools/fuzz_post.py: camal_postprocess on drawn CAM rows / logits (1..12 members, any length, moving-average widths
    1..15, gates, float64 data, NaN / inf / all-negative rows) against the oracle: states and averaged CAM bit for bit."""
    import fuzz_post as fp
    rng = np.random.default_rng(2028)
    failing = []
    for i in range(200):
        c = fp.draw(rng)
        bad = fp.check(c, fp.run_case(c))
        if bad:
            failing.append((i, c, bad))
    assert not failing, failing


def test_random_host_side_cases():
    """tools/fuzz_host.py: the window builder and the metric sums against their oracles (bit for bit / 1e-12), and the host
    pipeline (localize_host, localize_host_sharded: chunking, result dtypes, out= arrays, gates) against one call of
    localize on all windows (bit for bit)."""
    import fuzz_host as fh
    rng = np.random.default_rng(2029)
    failing = []
    for i in range(120):
        draw, run = list(fh.KINDS.values())[i % 3]
        c = draw(rng)
        bad = run(c)
        if bad:
            failing.append((i, c, bad))
    assert not failing, failing
