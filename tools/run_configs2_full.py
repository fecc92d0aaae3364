#!/usr/bin/env python3
"""BASELINE configs[2] at its stated scale on ONE GPU, host to host: N (default 10 M) synthetic REFIT-shaped meter-day
windows (L = 1440, float32) in host memory -> Ensemble.localize_host (bf16 ensemble of five members) -> int8 states,
float64 probabilities and float32 power in host memory.   tools/run_configs2_full.py [N] [out.json] [L]
Host memory: 4 + 1 + 4 bytes per timestamp = 130 GB at 10 M windows.  L = 10080: configs[4]'s week windows the same way."""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from camal_amd import synth  # noqa: E402
from camal_amd.calibrate import calibrated_state_dicts  # noqa: E402
from camal_amd.engine import Ensemble  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
outp = sys.argv[2] if len(sys.argv) > 2 else None
L, ks = (int(sys.argv[3]) if len(sys.argv) > 3 else 1440), [5, 7, 9, 15, 25]
t0 = time.time()
base = synth.refit_windows(8192 if L <= 2000 else 1024, L, seed=5)
sds = calibrated_state_dicts(ks, seed=0, sample_x=base[:64], device="cuda:0", precision="fp32")
ens = Ensemble(sds, device="cuda:0", precision="bf16")
x = np.empty((N, L), np.float32)
for s in range(0, N, len(base)):  # distinct windows per block: rolled by the block index
    n = min(len(base), N - s)
    x[s:s + n] = np.roll(base[:n], s // len(base), axis=1)
    if (s // len(base)) % 200 == 0:
        print(f"input {s / N:.0%}  {time.time() - t0:.0f} s", flush=True)
states = np.empty((N, L), np.int8)
power = np.empty((N, L), np.float32)
prob = np.empty((N,), np.float64)
ens.localize_host(base[:min(4096, len(base))], p_on=2000.0, scale_mul=1000.0)  # pins the pipeline's buffers
print(f"inputs ready {time.time() - t0:.0f} s", flush=True)
stop = threading.Event()


def beat():
    while not stop.wait(20):
        print(f"  running {time.time() - t1:.0f} s", flush=True)


t1 = time.time()
threading.Thread(target=beat, daemon=True).start()
ens.localize_host(x, p_on=2000.0, scale_mul=1000.0, out={"states": states, "prob": prob, "power": power})
dt = time.time() - t1
stop.set()
res = {"windows": N, "window_len": L, "members": ks, "precision": "bf16 (f32 accumulate)", "seconds": dt, "windows_per_s": N / dt,
       "host_bytes": int(x.nbytes + states.nbytes + power.nbytes + prob.nbytes), "state_on_fraction": float((states[::997] != 0).mean()),
       "windows_voted_on": float((prob > 0.5).mean()), "what": "NumPy float32 in host memory -> int8 states + float64 prob + float32 power in host memory, one GPU"}
print(json.dumps(res), flush=True)
if outp:
    json.dump(res, open(outp, "w"), indent=1)
