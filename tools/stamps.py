#!/usr/bin/env python3
"""Where a fused block-3 tile spends its cycles (diagnostic build: hipcc ... -DCAMAL_STAMPS -o build/libcamal_stamps.so).
   CAMAL_HIP_LIB=build/libcamal_stamps.so python tools/stamps.py [k]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from camal_amd import synth, weights, _lib
from camal_amd.engine import Ensemble
k = int(sys.argv[1]) if len(sys.argv) > 1 else 7
x = synth.refit_windows(512, 1440, seed=3)
ens = Ensemble([weights.make_state_dict(k, seed=1)], device="cuda:0", precision="bf16")
for _ in range(3):
    ens.localize(x)
torch.cuda.synchronize()
lib = C.CDLL(os.environ["CAMAL_HIP_LIB"])
buf = np.zeros((256, 9, 18), np.uint64)
assert lib.camal_debug_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
NW = int(os.environ.get("NW", "8"))
st = buf[:, :NW, :10].astype(np.int64)       # compute waves
d = np.diff(st, axis=2)                     # [wg][wave][9]
names = ["K1 conv k", "epi1 (LDS store)", "barrier B1", "K2 conv 7", "epi2", "barrier B2", "K3 conv 3", "head epilogue", "barrier END"]
tot = (st[:, :, 9] - st[:, :, 0]).mean()
print(f"k={k}: tile = {tot:.0f} cycles (mean over {st.shape[0]} workgroups x {NW} waves)")
for i, n in enumerate(names):
    print(f"  {n:18s} mean {d[:, :, i].mean():8.0f}  min {d[:, :, i].min():7d}  max {d[:, :, i].max():7d}   {100 * d[:, :, i].mean() / tot:5.1f} %")
mf = {5: 15, 7: 17, 9: 19, 15: 25, 25: 35}[k] * 4 * 16 * 16 * 2
print(f"  MFMA cycles per SIMD per tile (2 waves): {mf}  = {100 * mf / tot:.1f} % of the tile")
