#!/usr/bin/env python3
"""CamAL(mid_channels=...).train -> predict -> test end to end on a small synthetic set, for every supported width and the
reference's default kernel list (core.py:47-56): tools/e2e_widths.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camal_amd import CamAL, synth  # noqa: E402

L = 128
x, y = synth.ideal_windows(96, L, seed=3)
xv, yv = synth.ideal_windows(16, L, seed=4)
xt, yt = synth.ideal_windows(16, L, seed=5)
for mid in (32, 64, 128):
    for prec in ("fp32", "bf16"):
        t0 = time.time()
        m = CamAL(n_predictors=3, device="cuda:0", mid_channels=mid)
        m.precision = prec
        m.train((x, y), (xv, yv), (xt, yt), list_kernel_sizes=[5, 7, 9, 15, 25], n_try_by_kernel=1, epochs=2, batch_size=32)
        states, prob = m.predict(xt, return_prob=True)
        assert states.shape == xt.shape and prob.shape == (len(xt), 1) and np.isfinite(prob).all()
        data4 = np.zeros((len(xt), 2, 2, L))
        data4[:, 0, 0, :] = xt
        data4[:, 1, 0, :] = xt * 0.5
        data4[:, 1, 1, :] = (xt > 1.0)
        loc, clf = m.test(data4, appliance_mean_on_power=1500.0)
        print(f"mid {mid} {prec}: kept {m.camal_logs['camal_predictors_id']}  prob {prob[:4, 0].round(3).tolist()}  "
              f"on-fraction {float((states != 0).mean()):.3f}  F1 {loc.get('F1_SCORE')}  {time.time() - t0:.1f} s", flush=True)
print("ok")
