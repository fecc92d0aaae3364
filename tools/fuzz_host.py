#!/usr/bin/env python3
"""Randomised sweeps of the rows either side of the conv path (test infrastructure):

    python tools/fuzz_host.py [cases] [seed] [out.json]

  windows   camal_build_windows on drawn series (length, window, stride incl. stride > window, 1..3 appliances, NaN runs,
            float32 / float64, with and without a fitted NILMscaler) against oracle/windows_oracle: bit for bit;
  sums      camal_nilm_sums on drawn powers / states (NaN and inf in either power, states -1 / -0 / 0 / 1, labels 0 / 1 / 2)
            against float64 NumPy sums: counts exact, sums to 1e-12 relative (the order of the additions differs);
  pipeline  Ensemble.localize_host (chunks through pinned buffers, worker threads, result dtypes, out= arrays, y gate) and
            localize_host_sharded over two contexts on one GPU against ONE call of Ensemble.localize on all windows:
            bit for bit."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from camal_amd import metrics, synth, weights  # noqa: E402
from camal_amd.engine import Ensemble, localize_host_sharded  # noqa: E402
from camal_amd.windows import NILMscaler, build_windows  # noqa: E402
from oracle import windows_oracle  # noqa: E402


# ---- windows ---------------------------------------------------------------------------------------------------------
def draw_windows(rng):
    W = int(rng.choice([1, 2, 3, 7, 16, 63, 64, 65, 128, 255, 510, 600])) if rng.random() < 0.7 else int(rng.integers(1, 700))
    r = rng.random()
    S = W if r < 0.35 else int(rng.integers(1, max(2, int(1.6 * W) + 1)))
    T = int(W + rng.integers(0, 40000)) if rng.random() < 0.8 else int(W + rng.integers(0, 3 * W + 2))
    n_app = int(rng.integers(1, 4))
    return {"kind": "windows", "W": W, "S": S, "T": T, "n_app": n_app, "app": int(rng.integers(1, n_app + 1)),
            "nan_runs": int(rng.choice([0, 0, 3, 20, 200])), "f64": bool(rng.random() < 0.5),
            "scaler": [None, "StandardScaling", "MaxScaling", "MeanMaxScaling", 1000][int(rng.integers(0, 5))],
            "seed": int(rng.integers(0, 1 << 30))}


def run_windows(c):
    W, S, T = c["W"], c["S"], c["T"]
    stems = synth.house_series(max(T, 64), seed=c["seed"], n_app=c["n_app"], nan_runs=c["nan_runs"])[:, :T]
    series = stems.astype(np.float64 if c["f64"] else np.float32)
    Xd, kept = windows_oracle.nilm_dataset(series.astype(np.float64), W, S)
    sc = None
    if c["scaler"] is not None and len(Xd):
        sc = NILMscaler(power_scaling_type=c["scaler"], appliance_scaling_type="SameAsPower")
        sc.fit(Xd)
        if not np.isfinite(sc.power_affine()[1]) or sc.power_affine()[1] == 0:
            sc = None
    res = build_windows(series, W, S, scaler=sc, appliance=c["app"],
                        want=("x", "x64", "label", "app_power", "app_state", "index"))
    bad = []
    if res["count"] != len(kept):
        return [f"count {res['count']} != {len(kept)}"]
    s1, div = sc.power_affine() if sc is not None else (0.0, 1.0)
    ref64 = windows_oracle.scale_aggregate(Xd, s1, div) if len(Xd) else np.zeros((0, W))
    checks = {"index": (res["index"].cpu().numpy(), kept),
              "x64": (res["x64"].cpu().numpy(), ref64),
              "x": (res["x"].cpu().numpy(), ref64.astype(np.float32)),
              "label": (res["label"].cpu().numpy(), windows_oracle.weak_labels(Xd, c["app"])),
              "app_power": (res["app_power"].cpu().numpy(), Xd[:, c["app"], 0, :].astype(np.float32)),
              "app_state": (res["app_state"].cpu().numpy(), Xd[:, c["app"], 1, :].astype(np.int8))}
    for name, (got, ref) in checks.items():
        if not np.array_equal(got, ref):
            bad.append(name)
    return bad


# ---- sums ------------------------------------------------------------------------------------------------------------
def draw_sums(rng):
    return {"kind": "sums", "n": int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 65537, 300001])) if rng.random() < 0.6
            else int(rng.integers(1, 400000)), "f64": bool(rng.random() < 0.5), "specials": bool(rng.random() < 0.4),
            "reg": bool(rng.random() < 0.85), "cls": bool(rng.random() < 0.85), "seed": int(rng.integers(0, 1 << 30))}


def run_sums(c):
    rng = np.random.default_rng(c["seed"])
    n = c["n"]
    if not c["reg"] and not c["cls"]:
        c["reg"] = True
    y = (np.abs(rng.normal(300, 600, n)) * (rng.random(n) < 0.3)).astype(np.float64 if c["f64"] else np.float32)
    yh = (2000.0 * (rng.random(n) < 0.25)).astype(np.float32)
    if c["specials"]:
        for a in (y, yh):
            idx = rng.integers(0, n, size=min(n, 5))
            a[idx] = rng.choice([np.nan, np.inf, -np.inf], size=len(idx))
    ys = rng.choice([0, 1, 1, 0, 2] if c["specials"] else [0, 1], size=n).astype(np.int8)
    hs = rng.choice(np.array([0.0, 1.0, -1.0, -0.0], np.float32), size=n, p=[0.6, 0.3, 0.05, 0.05])
    sums, n_seen = metrics.nilm_sums_device(y=y if c["reg"] else None, y_hat=yh if c["reg"] else None,
                                            y_state=ys if c["cls"] else None, y_hat_state=hs if c["cls"] else None)
    sums = np.asarray(sums, np.float64)
    if n_seen != n:
        return [f"n {n_seen} != {n}"]
    ref = np.zeros(12)
    if c["reg"]:
        a = np.nan_to_num(y.astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
        b = np.nan_to_num(yh.astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
        d = b - a
        ref[:8] = [np.abs(d).sum(), (d * d).sum(), np.abs(a).sum(), (a * a).sum(), a.sum(), b.sum(), np.minimum(b, a).sum(),
                   np.maximum(b, a).sum()]
    if c["cls"]:
        hi = hs.astype(int)
        ref[8:] = [((ys == 1) & (hi == 1)).sum(), ((ys == 0) & (hi == 0)).sum(), ((ys == 0) & (hi == 1)).sum(),
                   ((ys == 1) & (hi == 0)).sum()]
    bad = []
    if not np.array_equal(sums[8:], ref[8:]):
        bad.append(f"counts {sums[8:].tolist()} != {ref[8:].tolist()}")
    if not np.allclose(sums[:8], ref[:8], rtol=1e-12, atol=1e-9):
        bad.append(f"sums off by {np.abs(sums[:8] - ref[:8]).max():.3g}")
    return bad


# ---- host pipeline ---------------------------------------------------------------------------------------------------
_ENS = {}


def ensembles(precision):
    if precision not in _ENS:
        x = synth.ukdale_windows(16, 256, seed=3)
        from camal_amd.calibrate import calibrated_state_dicts
        sds = calibrated_state_dicts([5, 9], seed=4, sample_x=x, device="cuda:0", precision="fp32")
        _ENS[precision] = [Ensemble(sds, device="cuda:0", precision=precision) for _ in range(2)]
    return _ENS[precision]


def draw_pipeline(rng):
    N = int(rng.integers(1, 70))
    return {"kind": "pipeline", "N": N, "L": int(4 * rng.integers(2, 130)) if rng.random() < 0.6 else int(rng.integers(5, 520)),
            "chunk": int(rng.integers(1, N + 4)), "f64": bool(rng.random() < 0.3), "signed": bool(rng.random() < 0.2),
            "use_y": bool(rng.random() < 0.3), "power": bool(rng.random() < 0.7), "w_ma": 4 if rng.random() < 0.2 else 5,
            "gate": int(rng.random() < 0.2), "states_dtype": ["float64", "float32", "int8"][int(rng.integers(0, 3))],
            "power_dtype": ["float64", "float32"][int(rng.integers(0, 2))], "out": bool(rng.random() < 0.4),
            "sharded": bool(rng.random() < 0.35), "precision": "bf16" if rng.random() < 0.3 else "fp32",
            "seed": int(rng.integers(0, 1 << 30))}


def run_pipeline(c):
    rng = np.random.default_rng(c["seed"])
    N, L = c["N"], c["L"]
    x = synth.ukdale_windows(N, max(L, 8), seed=c["seed"] % 100000)[:, :L]
    if c["signed"]:
        x = (x - np.float32(0.4)).astype(np.float32)
    data = x.astype(np.float64) if c["f64"] else x
    y = rng.integers(0, 2, N).astype(np.int8) if c["use_y"] else None
    ens = ensembles(c["precision"])
    p_on = 1800.0 if c["power"] else None
    one = ens[0].localize(x, x64=data if c["f64"] else None, y=y, w_ma=c["w_ma"], gate_mode=c["gate"], pred_f64=c["f64"],
                          p_on=p_on, scale_mul=1000.0, scale_add=0.25)
    torch.cuda.synchronize()
    sd, pd = np.dtype(c["states_dtype"]), np.dtype(c["power_dtype"])
    out = None
    if c["out"]:
        out = {"states": np.full((N, L), 7, sd), "prob": np.full((N,), -1.0), "power": np.full((N, L), -5, pd) if c["power"] else None}
    kw = dict(y=y, w_ma=c["w_ma"], gate_mode=c["gate"], p_on=p_on, scale_mul=1000.0, scale_add=0.25, chunk_windows=c["chunk"],
              out=out, states_dtype=sd, power_dtype=pd)
    if c["sharded"]:
        states, prob, power = localize_host_sharded(ens, data, **kw)
    else:
        states, prob, power = ens[0].localize_host(data, **kw)
    bad = []
    ref_s = one["state"].cpu().numpy()
    if states.dtype != sd:
        bad.append(f"states dtype {states.dtype}")
    exp = ref_s.astype(sd) if sd.kind == "f" else np.where(ref_s == 0, 0, ref_s).astype(sd)
    if not np.array_equal(states, exp):
        bad.append(f"states differ at {int((states != exp).sum())}")
    if sd.kind == "f" and not np.array_equal(np.signbit(states), np.signbit(ref_s)):
        bad.append("sign of zero")
    if not np.array_equal(prob.reshape(-1), one["prob"].cpu().numpy()):
        bad.append("prob")
    if c["power"]:
        if power.dtype != pd or not np.array_equal(power, one["power"].cpu().numpy().astype(pd)):
            bad.append("power")
    elif power is not None:
        bad.append("power returned without p_on")
    if c["out"] and (states is not out["states"] or prob is not out["prob"]):
        bad.append("out arrays not used")
    return bad


KINDS = {"windows": (draw_windows, run_windows), "sums": (draw_sums, run_sums), "pipeline": (draw_pipeline, run_pipeline)}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    outp = sys.argv[3] if len(sys.argv) > 3 else None
    rng = np.random.default_rng(seed)
    fails, counts, t0 = [], {k: 0 for k in KINDS}, time.time()
    for i in range(n):
        kind = list(KINDS)[i % 3]
        draw, run = KINDS[kind]
        c = draw(rng)
        counts[kind] += 1
        try:
            bad = run(c)
        except Exception as e:  # noqa: BLE001
            bad = [f"{type(e).__name__}: {e}"]
        if bad:
            fails.append({"case": c, "bad": bad})
            print("FAIL", i, json.dumps(c), bad, flush=True)
        if i % 30 == 29:
            print(f"{i + 1} cases, {len(fails)} failing, {time.time() - t0:.0f} s", flush=True)
    summary = {"cases": counts, "seed": seed, "failures": len(fails), "failing": fails[:20]}
    print(json.dumps({k: v for k, v in summary.items() if k != "failing"}))
    if outp:
        with open(outp, "w") as f:
            json.dump(summary, f, indent=1)
    for es in _ENS.values():
        for e in es:
            e.close()
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
