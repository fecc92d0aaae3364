#!/usr/bin/env python3
"""Throughput of the classifiers/resnet.py variant (kernels 7/5/3, zero padding) through camal_localize:
    python tools/bench_resnet_variant.py [fp32|bf16] [B] [L]
One member, inputs resident in HBM, wall clock over 10 steps after 3 warm-up steps."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from camal_amd import synth, weights  # noqa: E402
from camal_amd.engine import Ensemble  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
L = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
sd = weights.make_state_dict(7, seed=1, arch=weights.ARCH_RESNET, head_gain=3.0)
ens = Ensemble([sd], device="cuda:0", precision=prec)
x = torch.from_numpy(synth.ukdale_windows(B, L, seed=5)).to("cuda:0")
out = {"prob": torch.empty(B, dtype=torch.float64, device="cuda:0"),
       "state": torch.empty(B, L, dtype=torch.float32, device="cuda:0"),
       "power": torch.empty(B, L, dtype=torch.float32, device="cuda:0")}
for _ in range(3):
    ens.localize(x, p_on=2000.0, scale_mul=1000.0, out=out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    ens.localize(x, p_on=2000.0, scale_mul=1000.0, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 10
# MFMA conv MACs per timestep: every layer with Cin >= 64 (the Cin = 1 stem and skip run on the VALU)
macs = 64 * 64 * (5 + 3) + 64 * 128 * 7 + 128 * 128 * (5 + 3) + 64 * 128 + 128 * 128 * (7 + 5 + 3)
tf = 2.0 * macs * L * B / dt / 1e12
peak = {"fp32": 157.3, "bf16": 2500.0}[prec]
print(f"resnet variant {prec} B={B} L={L}: {B / dt:.0f} windows/s, {dt * 1e3:.2f} ms/step, {tf:.1f} TFLOP/s = {tf / peak:.3f} of {peak}")
