#!/usr/bin/env python3
"""Randomised sweep of the HIP training step against torch autograd on the CPU (oracle/torch_train; test
infrastructure):  python tools/fuzz_train.py [cases] [seed] [out.json]

Every case draws the first kernel size (odd, 1..31), mid_channels (32 / 64 / 128), the architecture, a batch of
1..48 windows (sometimes on a trainer created for a larger batch) and a window length on or next to the tile
boundaries of the forward / data-gradient / weight-gradient kernels (64, 128, 510, 512 steps) or anywhere in
8..700, and compares loss, correct count and EVERY gradient tensor.

What the gradients are compared with.  A ReLU's backward mask at a pre-activation within float32 rounding of zero is
decided by that rounding: two correct float32 implementations take dy or 0 there, the channel's d beta moves by one
dy and everything upstream follows (seen: 8e-3 of a tensor's maximum on a batch of one window, where torch-float32
itself sits 1e-6 from the float64 step because ITS rounding happened to agree with float64's).  So the reference
here is the float64 step of the piecewise-linear function the HIP forward pass actually evaluated: the ReLU masks
are read back from the trainer's workspace (camal_trainer_peek) and imposed on the float64 autograd
(FunctionalResNet(relu_masks=...)).  Against that every tensor must agree to north_star's 1e-3 of its maximum; the
number of mask elements that differ from the float64 forward pass's own is reported, and must stay below 1e-5 of
all elements (seen: at most 7 of some 6 M per case; about half the cases have one or more)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from camal_amd import synth, weights  # noqa: E402
from oracle import torch_train  # noqa: E402
from tests.helpers import is_bn_fed_bias, model_for  # noqa: E402

EDGES = [64, 128, 192, 256, 510, 512]
TOL = 1e-3  # north_star's gradient tolerance; seen over 300 cases: 5.6e-4 at worst, 1e-5 typically


def draw(rng):
    c = {"arch": "resnet" if rng.random() < 0.15 else "camal"}
    c["mid"] = 64 if c["arch"] == "resnet" else int(rng.choice([64, 64, 32, 128]))
    c["k"] = 7 if c["arch"] == "resnet" else int(rng.choice([1, 3, 5, 7, 9, 11, 15, 21, 25, 31]))
    c["L"] = int(max(8, rng.choice(EDGES) + rng.integers(-3, 4))) if rng.random() < 0.6 else int(rng.integers(8, 701))
    c["B"] = int(rng.integers(1, 49))
    if c["mid"] == 128:  # keep the CPU autograd of a case within seconds
        c["B"] = min(c["B"], 16)
    c["max_batch"] = c["B"] if rng.random() < 0.6 else int(c["B"] + rng.integers(1, 80))
    c["seed"] = int(rng.integers(0, 1 << 20))
    return c


def hip_relu_masks(eng, sd0, B, L):
    """{(block, conv): the 0/1 mask the HIP forward pass took} from the workspace: the ConvBlock's output > 0, or for the
    last ConvBlock of blocks 1 and 2 (whose stored output already has the skip added) the BatchNorm output recomputed
    from z and the batch statistics in the kernel's float32 arithmetic."""
    masks = {}
    for b in range(3):
        for i in range(3):
            if i < 2 or b == 2:
                a = eng.peek("act", 3 * b + i, B, L).cpu().numpy()
                masks[(b, i)] = (a > 0).transpose(0, 2, 1)
            else:
                slot = 4 * b + i
                z = eng.peek("z", slot, B, L).cpu().numpy()
                m, s = eng.peek("mean", slot, B, L).cpu().numpy(), eng.peek("invstd", slot, B, L).cpu().numpy()
                pre = f"layers.{b}.layers.{i}.layers.1"
                g, bt = np.asarray(sd0[pre + ".weight"], np.float32), np.asarray(sd0[pre + ".bias"], np.float32)
                # the kernels' own arithmetic: xh = (z - mean) * invstd in float32, then ONE fused multiply-add
                # fma(xh, gamma, beta) (hipcc contracts `xh * g + bt`; forward and backward kernels alike): its sign is
                # the sign of the exact xh * gamma + beta, which float64 holds (a float32 product is exact in 53 bits)
                xh = ((z - m).astype(np.float32) * s).astype(np.float32)
                o = xh.astype(np.float64) * g.astype(np.float64) + bt.astype(np.float64)
                masks[(b, i)] = (o > 0).transpose(0, 2, 1)
    return masks


def f64_relu_masks(sd0, x):
    import torch
    m = torch_train.FunctionalResNet(sd0, dtype=torch.float64)
    rec, orig = [], m._bn

    def bn(z, prefix, training):
        o = orig(z, prefix, training)
        rec.append((prefix, (o.detach().numpy() > 0)))
        return o
    m._bn = bn
    with torch.no_grad():
        m.forward(torch.as_tensor(np.asarray(x, np.float32)).to(torch.float64))
    out = {}
    for prefix, mk in rec:
        if "residual" in prefix:
            continue
        parts = prefix.split(".")  # layers.b.layers.i.layers.1
        out[(int(parts[1]), int(parts[3]))] = mk
    return out


def run_case(c):
    import torch
    from camal_amd.trainer import TrainEngine
    B, L = c["B"], c["L"]
    x, y = synth.ideal_windows(B, L, seed=c["seed"])
    sd0 = weights.make_state_dict(c["k"], seed=c["seed"] % 997, head_gain=1.0, arch=c["arch"], mid_channels=c["mid"])
    loss_ref, nc_ref, g32, _ = torch_train.loss_and_grads(sd0, x, y)
    m = model_for(sd0)
    m.to("cuda:0")
    m.load_state_dict(sd0)
    eng = TrainEngine(m, "cuda:0", c["max_batch"], L)
    try:
        loss, ncorrect = eng.forward_backward(x, y)
        flat = eng.grads.cpu().numpy()
        loss, ncorrect = float(loss.item()), int(ncorrect.item())
        keys = list(eng.keys)
        masks = hip_relu_masks(eng, sd0, B, L)
    finally:
        eng.close()
    _l, _n, g64, _ = torch_train.loss_and_grads(sd0, x, y, dtype=torch.float64, relu_masks=masks)
    m64 = f64_relu_masks(sd0, x)
    flips = int(sum((masks[k] != m64[k]).sum() for k in masks))
    elements = int(sum(mk.size for mk in masks.values()))
    grads, o = {}, 0
    for kk in keys:
        n = m._params[kk].size
        grads[kk] = flat[o:o + n].reshape(m._params[kk].shape)
        o += n
    gmax = max(np.abs(g).max() for g in g64.values())
    worst, worst_key, zero_noise, worst32 = 0.0, "", 0.0, 0.0
    for kk in keys:
        if is_bn_fed_bias(kk) or np.abs(g64[kk]).max() < 1e-5 * gmax:
            zero_noise = max(zero_noise, float(np.abs(grads[kk]).max() / gmax))
            continue
        err = float(np.abs(grads[kk] - g64[kk]).max() / np.abs(g64[kk]).max())
        if err > worst:
            worst, worst_key = err, kk
        worst32 = max(worst32, float(np.abs(grads[kk] - g32[kk]).max() / np.abs(g32[kk]).max()))
    return {"loss_err": abs(loss - loss_ref), "ncorrect_same": ncorrect == nc_ref, "grad_err": worst, "grad_key": worst_key,
            "zero_grad_noise": zero_noise, "relu_masks_off_float64": flips, "relu_elements": elements,
            "grad_err_vs_torch_float32": worst32}


def check(c, r):
    bad = []
    if not r["loss_err"] < 1e-4:
        bad.append(f"loss {r['loss_err']:.3g}")
    if not r["ncorrect_same"]:
        bad.append("correct count")
    if not r["grad_err"] < TOL:
        bad.append(f"gradient {r['grad_key']}: {r['grad_err']:.3g} off the float64 step on the same ReLU masks")
    if not r["zero_grad_noise"] < 1e-4:
        bad.append(f"zero gradients {r['zero_grad_noise']:.3g}")
    if not r["relu_masks_off_float64"] <= max(2, 1e-5 * r["relu_elements"]):
        bad.append(f"{r['relu_masks_off_float64']} of {r['relu_elements']} ReLU masks differ from the float64 forward pass")
    return bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    outp = sys.argv[3] if len(sys.argv) > 3 else None
    rng = np.random.default_rng(seed)
    fails = []
    worst = {"loss_err": 0.0, "grad_err": 0.0, "zero_grad_noise": 0.0, "grad_err_vs_torch_float32": 0.0, "relu_masks_off_float64": 0}
    flipped_cases, elements, worst_case = 0, 0, None
    t0 = time.time()
    for i in range(n):
        c = draw(rng)
        try:
            r = run_case(c)
            bad = check(c, r)
        except Exception as e:  # noqa: BLE001
            r, bad = {}, [f"{type(e).__name__}: {e}"]
        if bad:
            fails.append({"case": c, "result": r, "bad": bad})
            print("FAIL", i, json.dumps(c), bad, flush=True)
        if r.get("grad_err", 0.0) > worst["grad_err"]:
            worst_case = {"case": c, "key": r["grad_key"], "grad_err": r["grad_err"]}
        for k in worst:
            if k in r:
                worst[k] = max(worst[k], r[k])
        flipped_cases += int(bool(r.get("relu_masks_off_float64")))
        elements += r.get("relu_elements", 0)
        if i % 10 == 9:
            print(f"{i + 1} cases, {len(fails)} failing, {time.time() - t0:.0f} s", flush=True)
    summary = {"cases": n, "seed": seed, "failures": len(fails), "tolerance_same_masks": TOL, "worst": worst,
               "worst_gradient": worst_case, "cases_with_a_relu_mask_off_float64": flipped_cases, "relu_elements_total": elements, "failing": fails[:20]}
    print(json.dumps({k: v for k, v in summary.items() if k != "failing"}))
    if outp:
        with open(outp, "w") as f:
            json.dump(summary, f, indent=1)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
