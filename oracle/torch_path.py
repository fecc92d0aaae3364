"""CPU restatement of the reference's per-window predict loop with PyTorch-CPU ops.

TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/camal_oracle.py's header): imported by
``bench.py``'s ``cpu_baseline`` leg and by tests, never by the product path.

The reference's arithmetic on this path IS torch's (Conv1d / BatchNorm1d / Linear / softmax,
``camal_resnet.py:73-86``, ``class_activation_map.py:54-74``) followed by NumPy on the host
(``core.py:305-324,415-461``).  This module restates that module graph with
``torch.nn.functional`` calls (no reference source, no nn.Module copy) and keeps the
reference's control flow: one batch-1 forward per (member, window), the hooked feature map
moved to NumPy, a NumPy dot for the CAM — so that timing it on the host cores gives the
"reference CPU path" figure BASELINE.md §3 asks for.  Checked against the NumPy oracle and the
golden fixtures in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import camal_oracle as oracle


def _t(sd):
    return {k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}


def _conv_block(sd, p, x):
    w = sd[p + ".layers.0.weight"]
    k = w.shape[-1]
    if oracle.is_resnet_variant(sd):
        x = F.pad(x, (k // 2, k // 2))  # padding="same" (zeros), classifiers/resnet.py:62
    else:
        x = F.pad(x, (k // 2, k // 2), mode="replicate")  # padding_mode='replicate', camal_resnet.py:83
    z = F.conv1d(x, w, sd[p + ".layers.0.bias"])
    z = F.batch_norm(z, sd[p + ".layers.1.running_mean"], sd[p + ".layers.1.running_var"],
                     sd[p + ".layers.1.weight"], sd[p + ".layers.1.bias"], training=False, eps=1e-5)
    return F.relu(z)


def features(sd, x):
    """[N,1,L] -> block-3 feature map [N,128,L] (camal_resnet.py:21-22,56-59)."""
    for b in range(3):
        h = x
        for i in range(3):
            h = _conv_block(sd, f"layers.{b}.layers.{i}", h)
        rk = f"layers.{b}.residual.0.weight"
        if rk in sd:
            r = F.conv1d(x, sd[rk], sd[f"layers.{b}.residual.0.bias"])
            r = F.batch_norm(r, sd[f"layers.{b}.residual.1.running_mean"], sd[f"layers.{b}.residual.1.running_var"],
                             sd[f"layers.{b}.residual.1.weight"], sd[f"layers.{b}.residual.1.bias"],
                             training=False, eps=1e-5)
            h = h + r
        x = h
    return x


def cam_run(sd, instance):
    """One CAM.run call (class_activation_map.py:51-74) with label 1."""
    with torch.no_grad():
        xt = torch.tensor(np.asarray(instance).reshape(1, 1, -1)).float()
        f = features(sd, xt)
        logits = F.linear(f.mean(dim=-1), sd["linear.weight"], sd["linear.bias"])
        probs = F.softmax(logits, dim=1).squeeze()
        feats = f.numpy()  # SaveFeatures hook: feature map to host NumPy (:11)
        w = sd["linear.weight"].numpy()
        cam = w[1].dot(feats.reshape(feats.shape[1], feats.shape[2]))
        label = int(torch.topk(probs, 1)[1])
    return cam, label, probs.numpy()


def predict_loop(list_sd, data, w_ma=5):
    """CamAL.predict, device='cpu' path (core.py:286-332) -> (states, prob_detect)."""
    sds = [_t(sd) for sd in list_sd]
    data = np.asarray(data)
    prediction = np.zeros_like(data)
    prob_detect = np.zeros((len(data), 1))
    for sd in sds:
        for idx in range(len(data)):
            cam, y_pred, proba = cam_run(sd, data[idx])
            prob_detect[idx][0] += proba[1]
            if y_pred > 0:
                clip_cam = np.clip(cam, a_min=0, a_max=None)
                clip_cam = np.nan_to_num(clip_cam.astype(np.float32), nan=0.0, neginf=0.0, posinf=0.0)
                if clip_cam.max() > 0:
                    prediction[idx] += (clip_cam / clip_cam.max()).ravel()
    prob_detect = prob_detect / len(sds)
    prediction = prediction / len(sds)
    prediction = prediction * np.round(prob_detect)
    return oracle.attention_sigmoid_module(data, prediction, w_ma=w_ma), prob_detect


def predict_batched(list_sd, data, w_ma=5, batch=128):
    """The same algorithm with ONE forward per member over batches of windows instead of a batch-1 forward per
    (member, window) — BASELINE.md §3's second CPU figure (how much of the loop's time is loop overhead).  The
    arithmetic per window is predict_loop's; the batched conv may differ from the batch-1 conv in the last bit."""
    sds = [_t(sd) for sd in list_sd]
    data = np.asarray(data)
    prediction = np.zeros_like(data)
    prob_detect = np.zeros((len(data), 1))
    with torch.no_grad():
        for sd in sds:
            w = sd["linear.weight"].numpy()
            for s in range(0, len(data), batch):
                xt = torch.tensor(data[s:s + batch][:, None, :]).float()
                f = features(sd, xt)
                probs = F.softmax(F.linear(f.mean(dim=-1), sd["linear.weight"], sd["linear.bias"]), dim=1).numpy()
                cams = np.einsum("c,ncl->nl", w[1], f.numpy())
                for i in range(len(cams)):
                    prob_detect[s + i][0] += probs[i, 1]
                    if probs[i, 1] > probs[i, 0]:
                        clip_cam = np.nan_to_num(np.clip(cams[i], a_min=0, a_max=None).astype(np.float32), nan=0.0,
                                                 neginf=0.0, posinf=0.0)
                        if clip_cam.max() > 0:
                            prediction[s + i] += clip_cam / clip_cam.max()
    prob_detect = prob_detect / len(sds)
    prediction = prediction / len(sds)
    prediction = prediction * np.round(prob_detect)
    return oracle.attention_sigmoid_module(data, prediction, w_ma=w_ma), prob_detect
