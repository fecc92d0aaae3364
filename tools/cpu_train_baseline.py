#!/usr/bin/env python3
"""The training step of configs[3] on the host CPU: oracle/torch_train.train_steps (torch autograd + AdamW, the reference's
arithmetic) on one batch of 128 x 510 for each of the five members, threads as given to the job.  A reported baseline beside
bench.py's train_fp32 figure (test infrastructure; never on the product path).  tools/cpu_train_baseline.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from camal_amd import synth, weights  # noqa: E402
from oracle import torch_train  # noqa: E402

avail = len(os.sched_getaffinity(0))
threads = min(avail, int(os.environ.get("OMP_NUM_THREADS", "0")) or avail)
torch.set_num_threads(threads)
x, y = synth.ideal_windows(128, 510, seed=2000)
per_member = {}
for k in (5, 7, 9, 15, 25):
    sd = weights.make_state_dict(k, seed=k, head_gain=1.0)
    torch_train.train_steps(sd, [(x, y)])  # warm-up
    t0 = time.perf_counter()
    torch_train.train_steps(sd, [(x, y)] * 3)
    per_member[k] = (time.perf_counter() - t0) / 3
    print(k, per_member[k], flush=True)
step = sum(per_member.values())
res = {"seconds_per_ensemble_step": step, "windows_per_s": 128 / step, "threads": threads, "per_member_seconds": per_member,
       "what": "forward + backward + AdamW of the five members on one batch of 128 x 510, torch on the host CPU (oracle/torch_train)"}
print(json.dumps(res))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
