#!/usr/bin/env python3
"""Cycle stamps of the third 64-step tile of the 128 -> 128, 7-tap weight-gradient kernel (diagnostic build
hipcc ... -DCAMAL_STAMPS=4).   CAMAL_HIP_LIB=.../lib_stamps4.so python tools/stamps_wgrad.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import torch
from camal_amd import synth, weights
from camal_amd.resnet import CamALResNet
from camal_amd.trainer import TrainEngine
x, y = synth.ideal_windows(128, 510, seed=0)
m = CamALResNet(kernel_size=7)
eng = TrainEngine(m, "cuda:0", max_batch=128, length=510, data_parallel=False)
xs = torch.from_numpy(x).cuda(); ys = torch.from_numpy(np.asarray(y)).cuda()
for _ in range(3):
    eng.forward_backward(xs, ys)
torch.cuda.synchronize()
lib = C.CDLL(os.environ["CAMAL_HIP_LIB"])
buf = np.zeros((256, 9, 18), np.uint64)
assert lib.camal_debug_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
st = buf[:128, :4, :6].astype(np.int64)
d = np.diff(st, axis=2)
names = ["barrier 1 (tile free)", "stash (registers -> LDS)", "barrier 2", "fetch issue (next tile)", "MFMA loop (32 steps x 7 taps)"]
print(f"tile = {(st[:, :, 5] - st[:, :, 0]).mean():.0f} cycles; MFMA issue of one wave 14336")
for i, n in enumerate(names):
    print(f"  {n:34s} mean {d[:, :, i].mean():8.0f}  min {d[:, :, i].min():7d}  max {d[:, :, i].max():7d}")
