#!/usr/bin/env python3
"""Randomised parity sweep of the HIP path against the CPU oracle (test infrastructure, like tests/):

    python tools/fuzz_parity.py [cases] [seed] [out.json]

Every case draws a member set (first kernel sizes, mid_channels, architecture), a window length that sits on or
next to a tile boundary of one of the kernels (64 / 128-step fp32 tiles, 248 / 504 valid columns of the fused bf16
blocks, 248-step segments of the streaming post kernel) or anywhere in 1..2100, a window count, an internal chunk
size, the precision, the moving-average width, the gate and the float64 data path, and compares

  * CAM rows and member probabilities (camal_forward_cam)  with oracle.cam_batch,
  * ensemble probability, states (sign of zero included), averaged CAM and power (camal_localize) with
    oracle.camal_predict / power_estimate,

to the tolerances of tests/test_gpu_parity.py.  Prints one line per failing case and a summary; exit code 1 on any
failure.  The summary JSON keeps the worst margins per precision."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from camal_amd import synth, weights  # noqa: E402
from camal_amd.engine import Ensemble  # noqa: E402
from oracle import camal_oracle as oracle  # noqa: E402


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float((np.abs(a - b).max(axis=-1) / np.maximum(np.abs(b).max(axis=-1), 1e-12)).max())


def softmax(lg):
    e = np.exp(lg - lg.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


EDGES = [64, 128, 248, 256, 496, 504, 512, 744, 1008, 1024, 1240, 1440, 1488, 1512, 2016]


def draw(rng):
    c = {}
    c["precision"] = "bf16" if rng.random() < 0.5 else "fp32"
    c["arch"] = weights.ARCH_RESNET if rng.random() < 0.15 else weights.ARCH_CAMAL
    c["mid"] = int(rng.choice([64, 64, 64, 32, 128]))
    if c["arch"] == weights.ARCH_RESNET:
        c["mid"] = 64
        c["ks"] = [7] * int(rng.integers(1, 3))
    else:
        pool = [1, 3, 5, 7, 9, 11, 15, 21, 25, 31, 41]
        c["ks"] = [int(k) for k in rng.choice(pool, size=int(rng.integers(1, 4)), replace=False)]
        if rng.random() < 0.08:  # large ensembles: past 8 members the tiled post-processing kernel takes over, 16 is the limit
            c["ks"] = [int(k) for k in rng.choice(pool[:7], size=int(rng.integers(6, 17)))]
            c["mid"] = 64
    if rng.random() < 0.6:
        c["L"] = int(max(1, rng.choice(EDGES) + rng.integers(-3, 4)))
    else:
        c["L"] = int(rng.integers(1, 2101))
    c["B"] = int(rng.integers(1, 7))
    if len(c["ks"]) > 3:
        c["L"], c["B"] = min(c["L"], 600), min(c["B"], 3)  # keeps the oracle's share of a large ensemble in seconds
    c["chunk"] = int(rng.integers(1, 5)) if rng.random() < 0.3 else 0
    c["w_ma"] = 4 if rng.random() < 0.15 else 5
    c["gated"] = bool(rng.random() < 0.2)
    c["use_y"] = bool(rng.random() < 0.15)  # the caller's per-window labels (predict's y, core.py:297-300)
    c["f64"] = bool(rng.random() < 0.2)
    c["signed"] = bool(rng.random() < 0.15)
    c["seed"] = int(rng.integers(0, 1 << 30))
    c["fused"] = bool(rng.random() < 0.8)
    # a window shorter than the moving average: np.convolve(..., 'same') returns w values there and the reference's own
    # predict fails on the shapes (out of contract); the sweep keeps L >= w_ma
    c["L"] = max(c["L"], c["w_ma"])
    return c


def run_case(c):
    seed = c["seed"]
    L, B = c["L"], c["B"]
    x = synth.ukdale_windows(B, max(L, 8), seed=seed % 100000)[:, :L]
    if c["signed"]:
        x = (x - np.float32(0.4)).astype(np.float32)
    # heads calibrated with the oracle so that the members' votes spread around 0.5 on these windows (random heads
    # saturate at 0 or 1 and the vote logic would go untested): logits are linear in the head's gain and bias
    sds = []
    for i, k in enumerate(c["ks"]):
        kw = dict(seed=seed % 1000 + i, arch=c["arch"], mid_channels=c["mid"])
        lg = oracle.resnet_forward(weights.make_state_dict(k, head_gain=1.0, **kw), x[:, None, :])
        d = (lg[:, 1] - lg[:, 0]).astype(np.float64)
        med = float(np.median(d))
        scale = max(float(d.std()), 0.1 * abs(med), 1e-6)
        g = 1.5 / scale
        sds.append(weights.make_state_dict(k, head_gain=g, head_bias=(g * (med - 0.2 * scale), 0.0), **kw))
    data = x.astype(np.float64) if c["f64"] else x
    ens = Ensemble(sds, device="cuda:0", precision=c["precision"], chunk_windows=c["chunk"] or None)
    if not c["fused"]:
        ens.set_fused(False)
    out = {}
    try:
        logits, cam = ens.forward_cam(x)
        logits, cam = logits.cpu().numpy(), cam.cpu().numpy()
        y = np.random.default_rng(seed).integers(0, 2, B).astype(np.int8) if c.get("use_y") else None
        res = ens.localize(x, x64=data if c["f64"] else None, y=y, w_ma=c["w_ma"], gate_mode=int(c["gated"]), pred_f64=c["f64"],
                           want_avg_cam=True, p_on=2000.0, scale_mul=1000.0)
        torch.cuda.synchronize()
        prob, state = res["prob"].cpu().numpy(), res["state"].cpu().numpy()
        avg, power = res["avg_cam"].cpu().numpy(), res["power"].cpu().numpy()
    finally:
        ens.close()
    ostates, oprob, oavg, outs = oracle.camal_predict(sds, data, y=y, w_ma=c["w_ma"], gated=c["gated"] and y is None,
                                                      return_intermediate=True)
    oprob = np.asarray(oprob).reshape(-1)
    cam_err = prob_err = 0.0
    for e, (cams, _labels, probs) in enumerate(outs):
        cam_err = max(cam_err, rel_err(cam[e], cams))
        prob_err = max(prob_err, float(np.abs(softmax(logits[e]) - probs).max()))
    # member votes near 0.5 flip legitimately: only windows where every member's p1 and the ensemble's vote are firm count
    firm_tol = 2e-3 if c["precision"] == "fp32" else 5e-2
    member_p1 = np.stack([np.asarray(o[2])[:, 1] for o in outs])
    firm = (np.abs(member_p1 - 0.5) > firm_tol).all(axis=0) & (np.abs(oprob - 0.5) > firm_tol)
    out["cam_err"], out["prob_err"] = cam_err, prob_err
    out["ens_prob_err"] = float(np.abs(prob - oprob).max())
    out["firm"] = int(firm.sum())
    if firm.any():
        same = state[firm] == ostates[firm]
        out["state_agree"] = float(same.mean())
        out["sign_agree"] = float((np.signbit(state[firm]) == np.signbit(ostates[firm])).mean())
        out["avg_err"] = float(np.abs(avg[firm] - oavg[firm]).max())
        ref_power = oracle.power_estimate(state, x, 2000.0, scale_mul=1000.0)  # from the HIP states: the power step alone
        out["power_err"] = float(np.abs(power - ref_power).max())
    return out


def check(c, r):
    fp32 = c["precision"] == "fp32"
    cam_tol = 1e-3 if fp32 else 5e-2
    p_tol = 1e-3 if fp32 else 2e-2
    bad = []
    if not r["cam_err"] < cam_tol:
        bad.append(f"cam {r['cam_err']:.3g}")
    if not r["prob_err"] < p_tol:
        bad.append(f"member prob {r['prob_err']:.3g}")
    if r["firm"] == c["B"] and not r["ens_prob_err"] < p_tol:
        bad.append(f"ensemble prob {r['ens_prob_err']:.3g}")
    if "state_agree" in r:
        if r["state_agree"] < (0.999 if fp32 else 0.995):
            bad.append(f"states {r['state_agree']:.5f}")
        if fp32 and r["sign_agree"] < 0.999:
            bad.append(f"sign of zero {r['sign_agree']:.5f}")
        if fp32 and not r["avg_err"] < 1e-3:
            bad.append(f"avg cam {r['avg_err']:.3g}")
        if not r["power_err"] < 1e-2:
            bad.append(f"power {r['power_err']:.3g}")
    return bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    outp = sys.argv[3] if len(sys.argv) > 3 else None
    rng = np.random.default_rng(seed)
    worst = {"fp32": {}, "bf16": {}}
    fails = []
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
        w = worst[c["precision"]]
        for k, v in r.items():
            if k in ("firm",):
                continue
            w[k] = min(w.get(k, v), v) if k.endswith("agree") else max(w.get(k, v), v)
        if i % 20 == 19:
            print(f"{i + 1} cases, {len(fails)} failing, {time.time() - t0:.0f} s", flush=True)
    summary = {"cases": n, "seed": seed, "failures": len(fails), "worst": worst, "failing": fails[:20]}
    print(json.dumps(summary["worst"]))
    if outp:
        with open(outp, "w") as f:
            json.dump(summary, f, indent=1)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
