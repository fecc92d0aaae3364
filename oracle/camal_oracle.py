"""CPU restatement (NumPy) of CamAL's detection + CAM-localization path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``camal_amd``) never imports anything from ``oracle/``
and fails loudly when the HIP library is missing.

Pinning: the reference repository holds no tests / golden vectors for this path
(SURVEY.md §4).  This restatement is pinned against outputs of the reference
itself, imported in the dev container by ``tests/golden/make_golden.py``
(fixtures committed under ``tests/golden/``; ``tests/test_oracle_golden.py``
re-checks the restatement against them on every run).

Every function cites the reference lines (relative to the reference root) it
restates.  All arithmetic is float32 unless ``dtype`` says otherwise, like the
reference (torch fp32 on device, NumPy on host).
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5  # torch.nn.BatchNorm1d default used by camal_resnet.py:53,84

# ----------------------------------------------------------------------------
# ResNet forward  (src/models/camal/classifiers/camal_resnet.py)
# ----------------------------------------------------------------------------


def conv1d_replicate(x, w, b, dtype=np.float32):
    """Conv1d(stride 1, padding=k//2, padding_mode='replicate') — camal_resnet.py:82-83.

    z[n,co,t] = b[co] + sum_ci sum_j w[co,ci,j] * x[n,ci,clamp(t+j-k//2, 0, L-1)]
    x [N,Cin,L], w [Cout,Cin,k] (k odd), b [Cout] -> [N,Cout,L].
    The clamp applies to THIS layer's input (quirk 4 in SURVEY.md §2).
    """
    x = np.asarray(x, dtype=dtype)
    w = np.asarray(w, dtype=dtype)
    n, cin, length = x.shape
    cout, cin_w, k = w.shape
    assert cin == cin_w
    idx = np.clip(np.arange(length)[:, None] + np.arange(k)[None, :] - k // 2, 0, length - 1)
    w2 = w.reshape(cout, cin * k).T  # [(ci,j), co]
    out = np.empty((n, cout, length), dtype=dtype)
    for i in range(n):  # window at a time keeps the im2col buffer small
        cols = x[i][:, idx]  # [Cin, L, k]
        cols = cols.transpose(1, 0, 2).reshape(length, cin * k)
        out[i] = (cols @ w2).T
    if b is not None:
        out += np.asarray(b, dtype=dtype)[None, :, None]
    return out


def conv1d_same_zero(x, w, b, dtype=np.float32):
    """Conv1d(stride 1, padding="same") with zeros outside the window — classifiers/resnet.py:58-62
    (the ``ResNet`` variant; k odd, so "same" = k//2 zeros on either side)."""
    x = np.asarray(x, dtype=dtype)
    k = np.asarray(w).shape[2]
    xp = np.pad(x, ((0, 0), (0, 0), (k // 2, k // 2)))
    w = np.asarray(w, dtype=dtype)
    n, cin, lp = xp.shape
    length = lp - (k - 1)
    cout = w.shape[0]
    idx = np.arange(length)[:, None] + np.arange(k)[None, :]
    w2 = w.reshape(cout, cin * k).T
    out = np.empty((n, cout, length), dtype=dtype)
    for i in range(n):
        cols = xp[i][:, idx].transpose(1, 0, 2).reshape(length, cin * k)
        out[i] = (cols @ w2).T
    if b is not None:
        out += np.asarray(b, dtype=dtype)[None, :, None]
    return out


def is_resnet_variant(sd):
    """The two classifiers share every state-dict key; the second ConvBlock of a block has kernel 7 in
    CamALResNet (camal_resnet.py:31) and 5 in ResNet (classifiers/resnet.py:31)."""
    return int(np.asarray(sd["layers.0.layers.1.layers.0.weight"]).shape[2]) == 5


def conv1d_1x1(x, w, b, dtype=np.float32):
    """Conv1d(kernel_size=1), no padding — the residual branch, camal_resnet.py:52."""
    x = np.asarray(x, dtype=dtype)
    w2 = np.asarray(w, dtype=dtype)[:, :, 0]
    out = np.einsum("oc,ncl->nol", w2, x).astype(dtype)
    return out + np.asarray(b, dtype=dtype)[None, :, None]


def batchnorm_eval(z, gamma, beta, mean, var, dtype=np.float32):
    """BatchNorm1d in eval mode — camal_resnet.py:53,84 (always BN: SURVEY quirk 1)."""
    g = np.asarray(gamma, dtype=dtype)[None, :, None]
    bt = np.asarray(beta, dtype=dtype)[None, :, None]
    m = np.asarray(mean, dtype=dtype)[None, :, None]
    v = np.asarray(var, dtype=dtype)[None, :, None]
    return ((z - m) / np.sqrt(v + dtype(BN_EPS)) * g + bt).astype(dtype)


def _conv_block(sd, prefix, x, dtype):
    """ConvBlock = replicate-pad conv -> BN -> ReLU — camal_resnet.py:81-89
    (ResNet variant: zero-pad conv -> BN -> ReLU, classifiers/resnet.py:53-68)."""
    conv = conv1d_same_zero if is_resnet_variant(sd) else conv1d_replicate
    z = conv(x, sd[prefix + ".layers.0.weight"], sd[prefix + ".layers.0.bias"], dtype)
    z = batchnorm_eval(
        z,
        sd[prefix + ".layers.1.weight"],
        sd[prefix + ".layers.1.bias"],
        sd[prefix + ".layers.1.running_mean"],
        sd[prefix + ".layers.1.running_var"],
        dtype,
    )
    return np.maximum(z, 0)


def _resnet_block(sd, b, x, dtype):
    """ResNetBlock — camal_resnet.py:26-59.

    Three ConvBlocks with kernels [k,7,3]; blocks whose in/out channels differ add
    BN(Conv1x1(x)) with NO ReLU after the add (:56-59); block 3 has no skip at all.
    """
    h = x
    for i in range(3):
        h = _conv_block(sd, f"layers.{b}.layers.{i}", h, dtype)
    rkey = f"layers.{b}.residual.0.weight"
    if rkey in sd:
        r = conv1d_1x1(x, sd[rkey], sd[f"layers.{b}.residual.0.bias"], dtype)
        r = batchnorm_eval(
            r,
            sd[f"layers.{b}.residual.1.weight"],
            sd[f"layers.{b}.residual.1.bias"],
            sd[f"layers.{b}.residual.1.running_mean"],
            sd[f"layers.{b}.residual.1.running_var"],
            dtype,
        )
        h = h + r
    return h


def resnet_features(sd, x, dtype=np.float32):
    """Output of the last ResNetBlock, [N,128,L] — what SaveFeatures hooks
    (class_activation_map.py:8-12 on core.py:403's ``layers[2]``)."""
    x = np.asarray(x, dtype=dtype)
    if x.ndim == 2:
        x = x[:, None, :]
    for b in range(3):
        x = _resnet_block(sd, b, x, dtype)
    return x


def resnet_forward(sd, x, dtype=np.float32, return_features=False):
    """CamALResNet.forward — camal_resnet.py:21-23: Linear(mean_t(F)). -> logits [N,2]."""
    f = resnet_features(sd, x, dtype)
    pooled = f.mean(axis=-1, dtype=dtype)
    logits = pooled @ np.asarray(sd["linear.weight"], dtype=dtype).T + np.asarray(sd["linear.bias"], dtype=dtype)
    logits = logits.astype(dtype)
    if return_features:
        return logits, f
    return logits


def softmax(logits, dtype=np.float32):
    """F.softmax(prediction, dim=1) — class_activation_map.py:59."""
    z = logits - logits.max(axis=-1, keepdims=True)
    e = np.exp(z.astype(dtype))
    return (e / e.sum(axis=-1, keepdims=True)).astype(dtype)


# ----------------------------------------------------------------------------
# CAM  (src/helpers/class_activation_map.py:34-74)
# ----------------------------------------------------------------------------


def cam_run(sd, instance, returned_cam_for_label=1, dtype=np.float32):
    """CAM.run for one window — class_activation_map.py:34-38,51-74.

    Returns (cam[L] float32, label_pred int, proba[2] float32);
    cam[t] = sum_c W_fc[label,c] * F[c,t] (:42-48, bias not included).
    label_pred = top-1 of the probabilities (:64); ties resolve to index 0 here
    (the reference's tie behaviour is unpinned).
    """
    inst = np.asarray(instance).reshape(1, 1, -1)
    logits, feats = resnet_forward(sd, inst, dtype, return_features=True)
    probs = softmax(logits, dtype)[0]
    w = np.asarray(sd["linear.weight"], dtype=dtype)
    cam = w[returned_cam_for_label] @ feats[0]
    label = int(probs[1] > probs[0])
    return cam.astype(np.float32), label, probs.astype(np.float32)


def cam_batch(sd, data, returned_cam_for_label=1, dtype=np.float32, chunk=16):
    """Batched equivalent of calling ``cam_run`` window by window (same math)."""
    data = np.asarray(data)
    n, length = data.shape
    cams = np.empty((n, length), np.float32)
    labels = np.empty((n,), np.int32)
    probs = np.empty((n, 2), np.float32)
    w = np.asarray(sd["linear.weight"], dtype=dtype)
    for s in range(0, n, chunk):
        logits, feats = resnet_forward(sd, data[s : s + chunk, None, :], dtype, return_features=True)
        p = softmax(logits, dtype)
        probs[s : s + chunk] = p
        labels[s : s + chunk] = p[:, 1] > p[:, 0]
        cams[s : s + chunk] = np.einsum("c,ncl->nl", w[returned_cam_for_label], feats)
    return cams, labels, probs


# ----------------------------------------------------------------------------
# CamAL.predict  (src/models/camal/core.py:252-332, 415-461)
# ----------------------------------------------------------------------------


def moving_average(x, w):
    """core.py:439-449 — np.convolve(x, ones(w), 'same') / w (zero-padded edges)."""
    return np.convolve(x, np.ones(w), "same") / w


def rectified_sigmoid(z):
    """core.py:451-461 — 2*sigmoid(z) - 1."""
    return 2 * (1.0 / (1.0 + np.exp(-z))) - 1


def attention_sigmoid_module(data, soft_label, w_ma=5):
    """core.py:415-437 — moving average, rectified sigmoid of (ma*data), np.round."""
    soft_label = np.apply_along_axis(lambda r: moving_average(r, w_ma), axis=1, arr=soft_label)
    soft_label = rectified_sigmoid(soft_label * data)
    return np.round(soft_label)


def predict_subsequence_class(sd, data, dtype=np.float32, batch=128):
    """core.py:463-482 — batched argmax(logits) of ONE member (GPU-path pre-filter)."""
    data = np.asarray(data)
    out = []
    for s in range(0, len(data), batch):
        logits = resnet_forward(sd, data[s : s + batch, None, :], dtype)
        out.append(np.argmax(logits, axis=1))
    return np.concatenate(out).astype(np.int8)


def ensemble_accumulate(member_outputs, data, y=None):
    """The member/window double loop of CamAL.predict — core.py:286-324.

    member_outputs: list over members of (cams[N,L] f32, labels[N], probs[N,2]).
    Returns (avg_cam[N,L] (dtype of data, already vote-gated), prob_detect[N,1] f64).
    """
    data = np.asarray(data)
    prediction = np.zeros_like(data)
    prob_detect = np.zeros((len(data), 1))
    for cams, labels, probs in member_outputs:
        for idx in range(len(data)):
            if y is not None and y[idx] < 1:  # core.py:297-300
                continue
            prob_detect[idx][0] += probs[idx][1]  # :305
            if labels[idx] > 0:  # :308
                clip_cam = np.clip(cams[idx], a_min=0, a_max=None)  # :310
                clip_cam = np.nan_to_num(clip_cam.astype(np.float32), nan=0.0, neginf=0.0, posinf=0.0)
                if clip_cam.max() > 0:  # :315
                    clip_cam = clip_cam / clip_cam.max()
                    prediction[idx] += clip_cam.ravel()
    n_members = len(member_outputs)
    prob_detect = prob_detect / n_members  # :320
    prediction = prediction / n_members  # :321
    prediction = prediction * np.round(prob_detect)  # :324 (round half to even)
    return prediction, prob_detect


def camal_predict(list_sd, data, y=None, return_prob=False, w_ma=5, gated=False,
                  dtype=np.float32, return_intermediate=False):
    """CamAL.predict — core.py:252-332.

    ``gated=False`` is the reference's ``device == 'cpu'`` path (the canonical parity
    target).  ``gated=True`` restates the ``device != 'cpu'`` path:
    ``y`` is computed once from the FIRST member (core.py:292-293) and windows with
    ``y < 1`` are skipped for every member.
    """
    data = np.asarray(data)
    assert data.ndim <= 2
    if data.ndim == 1:
        data = data[None, :]
    if gated and y is None:
        y = predict_subsequence_class(list_sd[0], data, dtype)
    outs = [cam_batch(sd, data, 1, dtype) for sd in list_sd]
    avg_cam, prob_detect = ensemble_accumulate(outs, data, y)
    states = attention_sigmoid_module(data, avg_cam, w_ma=w_ma)
    if return_intermediate:
        return states, prob_detect, avg_cam, outs
    if return_prob:
        return states, prob_detect
    return states


def camal_predict_loop(list_sd, data, w_ma=5, dtype=np.float32):
    """Same result as ``camal_predict`` but with the reference's exact control flow:
    one batch-1 forward + CAM per (member, window) — core.py:290-317.  This is the
    variant timed as the CPU baseline."""
    data = np.asarray(data)
    prediction = np.zeros_like(data)
    prob_detect = np.zeros((len(data), 1))
    for sd in list_sd:
        for idx in range(len(data)):
            cam, y_pred, proba = cam_run(sd, data[idx], 1, dtype)
            prob_detect[idx][0] += proba[1]
            if y_pred > 0:
                clip_cam = np.clip(cam, a_min=0, a_max=None)
                clip_cam = np.nan_to_num(clip_cam.astype(np.float32), nan=0.0, neginf=0.0, posinf=0.0)
                if clip_cam.max() > 0:
                    prediction[idx] += (clip_cam / clip_cam.max()).ravel()
    prob_detect = prob_detect / len(list_sd)
    prediction = prediction / len(list_sd)
    prediction = prediction * np.round(prob_detect)
    return attention_sigmoid_module(data, prediction, w_ma=w_ma), prob_detect


# ----------------------------------------------------------------------------
# CamAL.test power step  (core.py:201-218; scaler: torch_dataset.py:131-156)
# ----------------------------------------------------------------------------


def power_estimate(states, data_scaled, appliance_mean_on_power, scale_mul=1.0, scale_add=0.0):
    """y_hat = int(state) * P_on, clipped to the inverse-scaled aggregate.

    core.py:202 casts the states to int; :216 multiplies by the mean on-power; :218
    clips to ``tmp_agg = data*stat2 + stat1`` (NILMscaler.inverse_transform,
    torch_dataset.py:144 for an integer ``power_scaling``: stat1=0, stat2=scaling).
    """
    y_hat_state = np.asarray(states).ravel().astype(int)
    agg = (np.asarray(data_scaled).ravel() * scale_mul + scale_add)
    y_hat = y_hat_state * appliance_mean_on_power
    y_hat = np.asarray(y_hat, dtype=np.result_type(y_hat, agg))
    over = y_hat > agg
    y_hat[over] = agg[over]
    return y_hat.reshape(np.asarray(states).shape)


# ----------------------------------------------------------------------------
# BN folding (not in the reference: algebra used by the HIP path, checked here)
# ----------------------------------------------------------------------------


def fold_bn(w, b, gamma, beta, mean, var):
    """(conv w,b) + BN(eval) -> (w', b') with y = conv(x; w', b')   (SURVEY §9)."""
    s = np.asarray(gamma, np.float64) / np.sqrt(np.asarray(var, np.float64) + BN_EPS)
    w2 = np.asarray(w, np.float64) * s[:, None, None]
    b2 = (np.asarray(b, np.float64) - np.asarray(mean, np.float64)) * s + np.asarray(beta, np.float64)
    return w2.astype(np.float32), b2.astype(np.float32)
