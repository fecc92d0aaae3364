#!/usr/bin/env python3
"""Which channels of one BatchNorm's d beta are off, and how many pre-ReLU outputs of that layer sit near zero:
tools/diag_train_mask.py '<case json>' <bn call index 0..10> <param key of the beta>"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camal_amd import synth, weights  # noqa: E402
from camal_amd.trainer import TrainEngine  # noqa: E402
from oracle import torch_train  # noqa: E402
from tests.helpers import model_for  # noqa: E402

c = json.loads(sys.argv[1])
idx, key = int(sys.argv[2]), sys.argv[3]
x, y = synth.ideal_windows(c["B"], c["L"], seed=c["seed"])
sd0 = weights.make_state_dict(c["k"], seed=c["seed"] % 997, head_gain=1.0, arch=c["arch"], mid_channels=c["mid"])
outs = {}
for dt in (torch.float32, torch.float64):
    m = torch_train.FunctionalResNet(sd0, dtype=dt)
    rec = []
    orig = m._bn
    m._bn = lambda z, prefix, training, _o=orig, _r=rec: (_r.append(_o(z, prefix, training)) or _r[-1])
    logits = m.forward(torch.as_tensor(np.asarray(x, np.float32)).to(dt))
    loss = torch.nn.functional.cross_entropy(logits, torch.as_tensor(np.asarray(y)).float().long())
    rec[idx].retain_grad()
    loss.backward()
    outs[dt] = (rec[idx].detach().numpy(), m.p[key].grad.numpy().copy(), rec[idx].grad.numpy().copy())
o64, g64, dy64 = outs[torch.float64]
o32, g32, _ = outs[torch.float32]
print("pre-ReLU BN output: shape", o64.shape, " |o| < 1e-6:", int((np.abs(o64) < 1e-6).sum()), " < 1e-5:", int((np.abs(o64) < 1e-5).sum()),
      " < 1e-4:", int((np.abs(o64) < 1e-4).sum()), " sign differs f32/f64:", int(((o32 > 0) != (o64 > 0)).sum()))
mm = model_for(sd0)
mm.to("cuda:0")
mm.load_state_dict(sd0)
eng = TrainEngine(mm, "cuda:0", c["max_batch"], c["L"])
eng.forward_backward(x, y)
flat = eng.grads.cpu().numpy()
o = 0
for kk in eng.keys:
    n = mm._params[kk].size
    if kk == key:
        g = flat[o:o + n]
    o += n
eng.close()
err = g - g64
order = np.argsort(-np.abs(err))[:8]
print("max |d beta| (f64)", np.abs(g64).max())
for ch in order:
    near = np.abs(o64[:, ch, :]) < 1e-4
    print(f"channel {ch:3d}: f64 {g64[ch]: .6e} hip {g[ch]: .6e} torch32 {g32[ch]: .6e}  err {err[ch]: .3e}   near-zero outputs {int(near.sum())}, "
          f"sum |dy| there {np.abs(dy64[:, ch, :][near]).sum():.3e}, typical |dy| {np.abs(dy64[:, ch, :]).mean():.3e}")
