#!/usr/bin/env python3
"""Randomised sweep of the post-processing kernels alone (camal_postprocess: CAM rows and logits the caller brings)
against the oracle's restatement of core.py:286-324, 415-437, 216-218 — test infrastructure:

    python tools/fuzz_post.py [cases] [seed] [out.json]

Draws 1..12 members, 1..9 windows, lengths 1..2100 (multiples of 4 or not: 16-byte or dword accesses), moving-average
widths 1..15, the two gate modes and caller labels y, float32 / float64 data (float32 or float64 accumulation of the
ensemble CAM), signed aggregates, CAM rows with NaN / +-inf, rows that are all <= 0 (max 0: the member adds nothing),
members voted off.  This arithmetic is restated operation by operation, so the comparison is EXACT where the
operations are IEEE ones: states bit for bit (sign of zero included) and the averaged CAM equal as float32 values.  The
float32 softmax goes through expf (the device's and NumPy's differ in the last bit): probabilities within 1e-6; the
power step multiplies a float32 aggregate in float64 here and in float32 in NumPy: within 1e-6 relative."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from camal_amd import _lib  # noqa: E402
from oracle import camal_oracle as oracle  # noqa: E402


def draw(rng):
    c = {"E": int(rng.integers(1, 13)), "B": int(rng.integers(1, 10))}
    r = rng.random()
    c["L"] = int(4 * rng.integers(1, 526)) if r < 0.5 else int(rng.integers(1, 2101))
    c["w_ma"] = int(rng.choice([1, 2, 3, 4, 5, 5, 5, 6, 7, 9, 15]))
    c["gate"] = int(rng.random() < 0.25)
    c["use_y"] = bool(rng.random() < 0.2)
    c["f64"] = bool(rng.random() < 0.3)
    c["signed"] = bool(rng.random() < 0.2)
    c["specials"] = bool(rng.random() < 0.3)
    c["want_avg"] = bool(rng.random() < 0.5)
    c["seed"] = int(rng.integers(0, 1 << 30))
    return c


def run_case(c):
    rng = np.random.default_rng(c["seed"])
    E, B, L = c["E"], c["B"], c["L"]
    t = np.arange(L)
    cams = np.empty((E, B, L), np.float32)
    for e in range(E):
        for b in range(B):
            bumps = sum(rng.uniform(0.2, 3.0) * np.exp(-0.5 * ((t - rng.uniform(0, L)) / rng.uniform(2, 60)) ** 2) for _ in range(3))
            cams[e, b] = (bumps + rng.normal(0, 0.3, L) - rng.uniform(0, 1.0)).astype(np.float32)
            if rng.random() < 0.1:
                cams[e, b] = -np.abs(cams[e, b])  # nothing above zero: max(clip) == 0
    if c["specials"]:
        for _ in range(int(rng.integers(1, 6))):
            cams[rng.integers(E), rng.integers(B), rng.integers(L)] = rng.choice([np.nan, np.inf, -np.inf])
    logits = rng.normal(0, 1.5, (E, B, 2)).astype(np.float32)
    x = np.abs(rng.normal(1.0, 1.5, (B, L))).astype(np.float32)
    if c["signed"]:
        x = (x - 1.0).astype(np.float32)
    data = x.astype(np.float64) * (1.0 + 1e-9 * rng.normal(size=(B, L))) if c["f64"] else x
    y = rng.integers(0, 2, B).astype(np.int8) if c["use_y"] else None
    p_on = float(rng.uniform(50, 3000))

    dev = "cuda:0"
    d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    cam_d, lg_d, x_d, x64_d, y_d = d(cams), d(logits), d(x), d(data if c["f64"] else None), d(y)
    prob_d = torch.empty(B, dtype=torch.float64, device=dev)
    avg_d = torch.empty((B, L), dtype=torch.float32, device=dev) if c["want_avg"] else None
    st_d = torch.empty((B, L), dtype=torch.float32, device=dev)
    pw_d = torch.empty((B, L), dtype=torch.float32, device=dev)
    lib = _lib.load()
    _lib.check(lib.camal_postprocess(E, _lib.ptr(cam_d), _lib.ptr(lg_d), _lib.ptr(x_d), _lib.ptr(x64_d), _lib.ptr(y_d), B, L,
                                     c["w_ma"], c["gate"], int(c["f64"]), p_on, 1000.0, 0.5, _lib.ptr(prob_d), _lib.ptr(avg_d),
                                     _lib.ptr(st_d), _lib.ptr(pw_d), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    prob, st, pw = prob_d.cpu().numpy(), st_d.cpu().numpy(), pw_d.cpu().numpy()

    outs = []
    for e in range(E):
        p = oracle.softmax(logits[e], np.float32)
        outs.append((cams[e], (p[:, 1] > p[:, 0]).astype(np.int32), p))
    yy = y
    if yy is None and c["gate"]:
        yy = outs[0][1].astype(np.int8)  # core.py:292-293: member 0's label gates every member
    with np.errstate(all="ignore"):
        oavg, oprob = oracle.ensemble_accumulate(outs, data, yy)
        ost = oracle.attention_sigmoid_module(data, oavg, w_ma=c["w_ma"]) if L >= c["w_ma"] else None
    r = {"prob_equal": bool(np.abs(prob - oprob[:, 0]).max() < 1e-6), "prob_max_diff": float(np.abs(prob - oprob[:, 0]).max())}
    if c["want_avg"]:
        r["avg_equal"] = bool(np.array_equal(avg_d.cpu().numpy(), oavg.astype(np.float32)))
        r["avg_max_diff"] = float(np.nanmax(np.abs(avg_d.cpu().numpy().astype(np.float64) - oavg)))
    if ost is not None:  # L < w: np.convolve's 'same' returns w values (the reference's shape bug, out of contract)
        r["state_equal"] = bool(np.array_equal(st, ost.astype(np.float32)) and np.array_equal(np.signbit(st), np.signbit(ost)))
        r["state_diffs"] = int((st != ost).sum())
        opw = oracle.power_estimate(ost, data, p_on, scale_mul=1000.0, scale_add=0.5).reshape(B, L)
        r["power_equal"] = bool(np.allclose(pw, opw, rtol=1e-6, atol=1e-3))
        r["power_max_diff"] = float(np.abs(pw - opw).max())
    return r


def check(c, r):
    return [k for k in ("prob_equal", "avg_equal", "state_equal", "power_equal") if k in r and not r[k]]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    outp = sys.argv[3] if len(sys.argv) > 3 else None
    rng = np.random.default_rng(seed)
    fails, t0 = [], time.time()
    for i in range(n):
        c = draw(rng)
        try:
            r = run_case(c)
            bad = check(c, r)
        except Exception as e:  # noqa: BLE001
            r, bad = {}, [f"{type(e).__name__}: {e}"]
        if bad:
            fails.append({"case": c, "result": r, "bad": bad})
            print("FAIL", i, json.dumps(c), bad, json.dumps(r), flush=True)
        if i % 50 == 49:
            print(f"{i + 1} cases, {len(fails)} failing, {time.time() - t0:.0f} s", flush=True)
    summary = {"cases": n, "seed": seed, "failures": len(fails), "comparison": "states and averaged CAM exact; probabilities 1e-6; power 1e-6 relative", "failing": fails[:20]}
    print(json.dumps({k: v for k, v in summary.items() if k != "failing"}))
    if outp:
        with open(outp, "w") as f:
            json.dump(summary, f, indent=1)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
