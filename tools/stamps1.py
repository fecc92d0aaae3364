#!/usr/bin/env python3
"""Cycle stamps of a fused block-1 tile (diagnostic build -DCAMAL_STAMPS; run with CAMAL_STAMPS_BLOCK1=1 semantics:
only block 1 runs, so that its stamps are not overwritten).   CAMAL_HIP_LIB=build/libcamal_stamps.so python tools/stamps1.py [k]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from camal_amd import synth, weights
from camal_amd.engine import Ensemble
k = int(sys.argv[1]) if len(sys.argv) > 1 else 7
x = synth.refit_windows(1024, 1440, seed=3)
ens = Ensemble([weights.make_state_dict(k, seed=1)], device="cuda:0", precision="bf16")
for _ in range(2):
    ens.localize(x)
torch.cuda.synchronize()
lib = C.CDLL(os.environ["CAMAL_HIP_LIB"])
buf = np.zeros((256, 9, 18), np.uint64)
assert lib.camal_debug_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
st = buf[:, :8, 10:18].astype(np.int64)
d = np.diff(st, axis=2)
names = ["stem (VALU) + A1 store", "barrier", "K2 conv 7", "epi2 + barrier", "K3 conv 3", "final epilogue (HBM store)", "x swap + END barrier"]
tot = (st[:, :, 7] - st[:, :, 0]).mean()
print(f"block 1, k={k}: tile = {tot:.0f} cycles")
for i, n in enumerate(names):
    print(f"  {n:28s} mean {d[:, :, i].mean():8.0f}  min {d[:, :, i].min():7d}  max {d[:, :, i].max():7d}   {100 * d[:, :, i].mean() / tot:5.1f} %")
print(f"  MFMA cycles per SIMD per tile (2 waves): {10 * 2 * 16 * 16 * 2}  = {100 * 10 * 2 * 16 * 16 * 2 / tot:.1f} % of the tile")
