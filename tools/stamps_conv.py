#!/usr/bin/env python3
"""Cycle stamps of a tile of the fp32 per-layer kernel of the 128 -> 128, k = 3 ConvBlock with the skip add (block 2's last
layer): diagnostic build hipcc ... -DCAMAL_STAMPS=3.   CAMAL_HIP_LIB=.../lib_stamps3.so python tools/stamps_conv.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from camal_amd import synth, weights
from camal_amd.engine import Ensemble
x = synth.ukdale_windows(1024, 1024, seed=3)
ens = Ensemble([weights.make_state_dict(7, seed=1)], device="cuda:0", precision="fp32")
for _ in range(3):
    ens.localize(x)
torch.cuda.synchronize()
lib = C.CDLL(os.environ["CAMAL_HIP_LIB"])
buf = np.zeros((256, 9, 18), np.uint64)
assert lib.camal_debug_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
st = buf[:, :4, :4].astype(np.int64)
d = np.diff(st, axis=2)
names = ["K loop (3 taps, 128 ch)", "epilogue (residual loads, add, stores)", "end-of-tile barrier"]
tot = (st[:, :, 3] - st[:, :, 0]).mean()
print(f"tile = {tot:.0f} cycles (mean over 256 workgroups x 4 waves); MFMA issue of one wave 24576, of the two that share a SIMD 49152")
for i, n in enumerate(names):
    print(f"  {n:42s} mean {d[:, :, i].mean():8.0f}  min {d[:, :, i].min():7d}  max {d[:, :, i].max():7d}")
