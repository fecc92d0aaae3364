#!/usr/bin/env python3
"""CamAL.train end to end at configs[3]'s shape on ONE GPU: N IDEAL-shaped windows of 510 steps (NumPy, host memory), the
reference's five kernel sizes, batches of 128, a validation and a test set, E epochs -> the fitted ensemble.
tools/run_train_full.py [N] [epochs] [member_streams] [out.json]
Reports ensemble-steps per second against bench.py's resident step (five members' forward + backward + AdamW on one batch)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from camal_amd import CamAL, synth  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 16384
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 5
outp = sys.argv[4] if len(sys.argv) > 4 else None
L = 510
x, y = synth.ideal_windows(N, L, seed=1)
xv, yv = synth.ideal_windows(1024, L, seed=2)
xt, yt = synth.ideal_windows(1024, L, seed=3)
torch.manual_seed(0)
m = CamAL(n_predictors=5, device="cuda:0")
m.member_streams = streams
t0 = time.time()
m.train((x, y), (xv, yv), (xt, yt), list_kernel_sizes=[5, 7, 9, 15, 25], n_try_by_kernel=1, epochs=epochs, batch_size=128,
        patience_es=None, patience_rlr=None)
torch.cuda.synchronize()
dt = time.time() - t0
n_bal = int(2 * min((y == 0).sum(), (y == 1).sum()))  # the undersampler balances the classes (core.py:66-68)
steps = epochs * ((n_bal + 127) // 128)
states, prob = m.predict(xt[:256], return_prob=True)
logs = m.camal_logs["ensemble_training_logs"]
res = {"train_windows": N, "balanced_windows": n_bal, "window_len": L, "epochs": epochs, "member_streams": streams,
       "kernel_sizes": [5, 7, 9, 15, 25], "seconds": dt, "ensemble_steps": steps, "ms_per_ensemble_step": dt / steps * 1e3,
       "trained_windows_per_s": n_bal * epochs / dt,
       "final_train_loss": {k: float(v["loss_train_history"][-1]) for k, v in logs.items()},
       "test_accuracy_of_vote": float(((prob[:, 0] > 0.5) == (yt[:256] > 0)).mean()),
       "what": "CamAL.train on NumPy arrays: undersampling, five members (side by side on member_streams), per-epoch validation "
               "(1024 windows) and the final test evaluation included; one ensemble step = one batch of 128 through every member"}
print(json.dumps(res), flush=True)
if outp:
    json.dump(res, open(outp, "w"), indent=1)
