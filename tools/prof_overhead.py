#!/usr/bin/env python3
"""Wall clock of the ensemble_bf16 step with and without the library's per-launch HIP events (camal_set_profiling)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from camal_amd import synth  # noqa: E402
from camal_amd.calibrate import calibrated_state_dicts  # noqa: E402
from camal_amd.engine import Ensemble  # noqa: E402

B, L, ks = 4096, 1440, [5, 7, 9, 15, 25]
xh = synth.refit_windows(B, L, seed=1000)
sds = calibrated_state_dicts(ks, seed=0, sample_x=xh[:64], device="cuda:0", precision="fp32")
ens = Ensemble(sds, device="cuda:0", precision="bf16")
x = torch.from_numpy(xh).to("cuda:0")
out = {"prob": torch.empty(B, dtype=torch.float64, device="cuda:0"), "state": torch.empty(B, L, device="cuda:0"),
       "power": torch.empty(B, L, device="cuda:0")}
for _ in range(3):
    ens.localize(x, p_on=2000.0, scale_mul=1000.0, out=out)
for rep in range(2):
    for prof in (False, True):
        ens.set_profiling(prof)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            ens.localize(x, p_on=2000.0, scale_mul=1000.0, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 10
        if prof:
            ens.get_profile()
        print(f"profiling={prof}: {dt * 1e3:.2f} ms/step, {B / dt:.0f} windows/s", flush=True)
