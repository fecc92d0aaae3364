#!/usr/bin/env python3
"""One member's training step alone (no other stream beside it), for kernel traces: tools/train_one.py [k] [steps].
Under `rocprofv3 --kernel-trace` the trace of the last step lists every launch in order with its own duration
(bench.py's train_fp32 runs five members on five streams: their kernels overlap and the durations inflate)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from camal_amd import synth  # noqa: E402
from camal_amd.resnet import CamALResNet  # noqa: E402
from camal_amd.trainer import TrainEngine  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 7
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B, L = 128, 510
x, y = synth.ideal_windows(B, L, seed=1)
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda().to(torch.int32)
torch.manual_seed(k)
eng = TrainEngine(CamALResNet(kernel_size=k), "cuda:0", B, L)
for _ in range(3):
    eng.forward_backward(xd, yd)
    eng.optimizer_step(1e-3, 0.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.forward_backward(xd, yd)
    eng.optimizer_step(1e-3, 0.0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
macs = 24576 * k + 376832
print(f"k={k} {dt * 1e3:.3f} ms/step  {3 * 2.0 * L * B * macs / dt / 1e12:.1f} TFLOP/s (3x forward conv FLOPs)")
