#!/usr/bin/env python3
"""Forward-pass accuracy of the training step, layer by layer, against the float64 step: the HIP path's pre-BatchNorm
conv outputs z and batch statistics (camal_trainer_peek) beside torch-float32's: tools/diag_train_fwd.py '<case json>'"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camal_amd import synth, weights  # noqa: E402
from camal_amd.trainer import TrainEngine  # noqa: E402
from oracle import torch_train  # noqa: E402
from tests.helpers import model_for  # noqa: E402

c = json.loads(sys.argv[1])
x, y = synth.ideal_windows(c["B"], c["L"], seed=c["seed"])
sd0 = weights.make_state_dict(c["k"], seed=c["seed"] % 997, head_gain=1.0, arch=c["arch"], mid_channels=c["mid"])


def torch_layers(dt):
    """[(z, bn output)] in slot order 0..10 (block 1: conv 0,1,2, skip; block 2: conv 0,1,2, skip; block 3: conv 0,1,2)."""
    m = torch_train.FunctionalResNet(sd0, dtype=dt)
    rec = []
    orig = m._bn

    def bn(z, prefix, training):
        o = orig(z, prefix, training)
        rec.append((prefix, z.detach().numpy().copy(), o.detach().numpy().copy()))
        return o
    m._bn = bn
    m.forward(torch.as_tensor(np.asarray(x, np.float32)).to(dt))
    return rec


r64, r32 = torch_layers(torch.float64), torch_layers(torch.float32)
mm = model_for(sd0)
mm.to("cuda:0")
mm.load_state_dict(sd0)
eng = TrainEngine(mm, "cuda:0", c["max_batch"], c["L"])
eng.forward_backward(x, y)
B, L = c["B"], c["L"]
# torch's _bn call order: block b: conv 0, 1, 2 then the skip -> the library's slot order: block 1: 0,1,2,3; 2: 4,5,6,7; 3: 8,9,10
# (the library computes the skip of block 2 right after its conv 0: the slot index is still parameters() order)
for slot, ((prefix, z64, o64), (_p, z32, o32)) in enumerate(zip(r64, r32)):
    zh = eng.peek("z", slot, B, L).cpu().numpy().transpose(0, 2, 1)  # [B, C, L]
    mean, invstd = eng.peek("mean", slot, B, L).cpu().numpy(), eng.peek("invstd", slot, B, L).cpu().numpy()
    zs = np.abs(z64).max()
    m64 = z64.mean(axis=(0, 2))
    is64 = 1.0 / np.sqrt(z64.var(axis=(0, 2)) + 1e-5)
    m32 = z32.astype(np.float64).mean(axis=(0, 2))
    g = np.asarray(sd0[prefix + ".weight"], np.float64)[None, :, None]
    bt = np.asarray(sd0[prefix + ".bias"], np.float64)[None, :, None]
    oh = (zh.astype(np.float32) - mean[None, :, None]) * invstd[None, :, None] * g.astype(np.float32) + bt.astype(np.float32)
    print(f"{slot:2d} {prefix:28s} |z|max {zs:8.3f}  z err: hip {np.abs(zh - z64).max() / zs:8.2e} (rms {np.sqrt(((zh - z64) ** 2).mean()) / zs:8.2e})"
          f"  torch32 {np.abs(z32 - z64).max() / zs:8.2e} (rms {np.sqrt(((z32 - z64) ** 2).mean()) / zs:8.2e})"
          f" | mean err hip {np.abs(mean - m64).max():8.2e}  invstd rel err hip {np.abs(invstd / is64 - 1).max():8.2e}"
          f" | bn out err: hip {np.abs(oh - o64).max():8.2e} torch32 {np.abs(o32 - o64).max():8.2e}"
          f" | sign flips vs f64: hip {int(((oh > 0) != (o64 > 0)).sum())} torch32 {int(((o32 > 0) != (o64 > 0)).sum())} of {o64.size}")
eng.close()
