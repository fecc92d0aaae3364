"""GPU: how far inside the parity bounds the HIP path sits.  For every fixture the REFERENCE
produced (tests/golden/*.npz) this records the achieved errors — max row-normalised CAM error, max RELATIVE
probability error on p >= 1e-3 (and the absolute one), state agreement of the fp32 and bf16 paths, per-tensor gradient
errors of the training step — into ``gpurun_out/parity_margins.json`` (copied to profiles/parity_margins.json when it
is committed), and asserts the bounds the margins allow: fp32 probabilities to 1e-4 RELATIVE on p >= 1e-3 (the other
tests check 1e-3 absolute), fp32 CAM rows to 1e-4 of the row's maximum (measured: about 1e-6 for both), gradients to
1e-3 of the tensor's maximum (measured: 3e-6 .. 2.6e-4)."""
import json
import os

import numpy as np
import pytest

from tests.helpers import (REL_TOL_FP32, STATE_AGREE_MIN, case_state_dicts, golden_cases, is_bn_fed_bias, load_case,
                           rel_err, train_case_inputs)

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TIGHT = 1e-4  # what the fp32 path is held to here: RELATIVE probability error and CAM error, 10x inside the north_star bound
OUT = os.environ.get("CAMAL_MARGINS_OUT", os.path.join(ROOT, "gpurun_out", "parity_margins.json"))


def _softmax(lg):
    z = lg - lg.max(-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(-1, keepdims=True)


def _rel_prob_err(p, ref, floor=1e-3):
    """max |p - ref| / ref over the entries with ref >= floor (relative error is undefined towards p = 0)."""
    m = ref >= floor
    return float((np.abs(p - ref)[m] / ref[m]).max()) if m.any() else 0.0


def test_record_parity_margins():
    import torch
    from camal_amd import synth
    from camal_amd.engine import Ensemble
    assert torch.cuda.is_available()
    rec = {"what": "achieved errors of the HIP path against the reference's own outputs (tests/golden), per fixture",
           "bounds": {"cam_rel_to_row_max": REL_TOL_FP32, "prob_rel_on_p>=1e-3": REL_TOL_FP32,
                      "state_agreement_min": STATE_AGREE_MIN, "grad_rel_to_tensor_max": 1e-3}, "inference": {}, "training": {}}

    # ---- inference fixtures with full CAM rows -----------------------------------------------------------------
    for name in golden_cases("ensemble_") + golden_cases("pair_"):
        c = load_case(name)
        data = c["data"]
        sds = case_state_dicts(c)
        r = {}
        for prec in ("fp32", "bf16"):
            ens = Ensemble(sds, device="cuda:0", precision=prec)
            res = ens.localize(data.astype(np.float32), x64=data if data.dtype == np.float64 else None, w_ma=int(c["w_ma"]),
                               gate_mode=int(bool(c["gated"])), pred_f64=data.dtype == np.float64, want_members=True)
            mcam, mlog = res["member_cam"].cpu().numpy(), res["member_logits"].cpu().numpy()
            probs = _softmax(mlog.astype(np.float64))
            prob, state = res["prob"].cpu().numpy(), res["state"].cpu().numpy()
            refp = c["prob_detect"][:, 0]
            safe = np.abs(refp - 0.5) > (2e-3 if prec == "fp32" else 5e-2)
            r[prec] = {
                "cam_max_rel_to_row_max": max(rel_err(mcam[e], c["cams"][e]) for e in range(len(sds))),
                "member_prob_max_rel": _rel_prob_err(probs, c["probs"].astype(np.float64)),
                "member_prob_max_abs": float(np.abs(probs - c["probs"]).max()),
                "ensemble_prob_max_rel": _rel_prob_err(prob, refp) if not bool(c["gated"]) else None,
                "ensemble_prob_max_abs": float(np.abs(prob - refp).max()),
                "state_agreement": float((state[safe] == c["states"][safe]).mean()) if safe.any() else None,
                "windows_counted": int(safe.sum()), "windows": int(len(safe)),
            }
            ens.close()
        f = r["fp32"]
        # measured 4e-7 .. 1.4e-6 (profiles/parity_margins.json): asserted at a tenth of the north_star's 1e-3
        assert f["cam_max_rel_to_row_max"] < TIGHT and f["member_prob_max_rel"] < TIGHT, (name, f)
        assert f["state_agreement"] is None or f["state_agreement"] >= STATE_AGREE_MIN, (name, f)
        rec["inference"][name] = r

    # ---- the full-size fixtures (strided CAM samples) ------------------------------------------------------------
    for name, gen in (("cfg0_l1024", synth.ukdale_windows), ("week_l10080", synth.refit_windows)):
        c = load_case(name)
        sds = case_state_dicts(c)
        x = gen(int(c["n"]), int(c["length"]), seed=int(c["data_seed"]))
        st = int(c["cam_stride"])
        refp = c["prob_detect"][:, 0]
        r = {}
        for prec in ("fp32", "bf16"):
            ens = Ensemble(sds, device="cuda:0", precision=prec)
            res = ens.localize(x, want_members=True)
            mcam, mlog = res["member_cam"].cpu().numpy(), res["member_logits"].cpu().numpy()
            prob, state = res["prob"].cpu().numpy(), res["state"].cpu().numpy()
            cam_err = max(float((np.abs(mcam[e][:, ::st] - c["cams_strided"][e]).max(axis=-1) / np.abs(c["cam_row_max"][e])).max())
                          for e in range(len(sds)))
            safe = np.abs(refp - 0.5) > 2e-3  # the fp32 criterion for both precisions: every counted window, every timestamp
            r[prec] = {"cam_max_rel_to_row_max": cam_err, "ensemble_prob_max_rel": _rel_prob_err(prob, refp),
                       "ensemble_prob_max_abs": float(np.abs(prob - refp).max()),
                       "state_agreement": float((state[safe] == c["states"][safe]).mean()),
                       "windows_counted": int(safe.sum()), "windows": int(len(safe))}
            if "probs" in c:
                p1 = 1.0 / (1.0 + np.exp(mlog[..., 0].astype(np.float64) - mlog[..., 1]))
                r[prec]["member_prob_max_rel"] = _rel_prob_err(p1, c["probs"][..., 1].astype(np.float64))
            ens.close()
        assert r["fp32"]["cam_max_rel_to_row_max"] < TIGHT and r["fp32"]["ensemble_prob_max_rel"] < TIGHT, (name, r)
        assert r["fp32"]["state_agreement"] >= STATE_AGREE_MIN and r["bf16"]["state_agreement"] >= STATE_AGREE_MIN, (name, r)
        rec["inference"][name] = r

    # ---- training fixtures: every gradient tensor against the torch-autograd oracle on the fixture's batch ---------
    from oracle import torch_train
    from tests.test_gpu_train import _engine, _grads_dict
    for name in golden_cases("train_"):
        c = load_case(name)
        k, x, y, sd0 = train_case_inputs(c)
        B = int(c["batch"])
        m, eng = _engine(sd0, k, B, x.shape[1])
        loss, _ = eng.forward_backward(x[:B], y[:B])
        grads = _grads_dict(eng, m)
        loss_ref, _nc, g_ref, _sd1 = torch_train.loss_and_grads(sd0, x[:B], y[:B])
        gmax = max(np.abs(g).max() for g in g_ref.values())
        errs = {kk: float(np.abs(grads[kk] - g_ref[kk]).max() / np.abs(g_ref[kk]).max()) for kk in eng.keys
                if not is_bn_fed_bias(kk) and np.abs(g_ref[kk]).max() >= 1e-5 * gmax}
        worst = max(errs, key=errs.get)
        rec["training"][name] = {"loss_abs_err_vs_reference_fixture": abs(float(loss.item()) - float(c["step1_loss"])),
                                 "loss_abs_err_vs_oracle": abs(float(loss.item()) - float(loss_ref)),
                                 "grad_max_rel_to_tensor_max": errs[worst], "worst_tensor": worst, "tensors": len(errs)}
        assert errs[worst] < 1e-3, (name, worst, errs[worst])
        eng.close()

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
    worst_cam = max(v["fp32"]["cam_max_rel_to_row_max"] for v in rec["inference"].values())
    worst_p = max(v["fp32"].get("member_prob_max_rel", 0.0) or 0.0 for v in rec["inference"].values())
    print(f"parity margins: worst fp32 CAM error {worst_cam:.2e} of the row maximum, worst relative probability error {worst_p:.2e} "
          f"(bounds {REL_TOL_FP32:g}); written to {OUT}")
