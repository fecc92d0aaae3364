"""Metrics with the reference's key names (``src/helpers/other.py:86-159``).

NumPy restatements of the sklearn calls the reference wraps, for binary labels, for host arrays; and
the device route ``CamAL.test`` takes for the localization metrics: the sums behind
``NILMmetrics`` come from ``camal_nilm_sums`` (one HBM pass over states and powers that stay on the
device), the formulas on top of the twelve sums are shared with the host route.
"""
from __future__ import annotations

import numpy as np


def _confusion(y, yh):
    y = np.asarray(y).astype(np.int64).ravel()
    yh = np.asarray(yh).astype(np.int64).ravel()
    tp = int(np.sum((y == 1) & (yh == 1)))
    tn = int(np.sum((y == 0) & (yh == 0)))
    fp = int(np.sum((y == 0) & (yh == 1)))
    fn = int(np.sum((y == 1) & (yh == 0)))
    return tp, tn, fp, fn


def _div(a, b):
    return float(a) / float(b) if b else 0.0


def _binary_scores(y, yh):
    return _scores_from_counts(*_confusion(y, yh))


def _scores_from_counts(tp, tn, fp, fn):
    n = tp + tn + fp + fn
    prec, rec = _div(tp, tp + fp), _div(tp, tp + fn)
    spec = _div(tn, tn + fp)
    f1 = _div(2 * tp, 2 * tp + fp + fn)
    f1_neg = _div(2 * tn, 2 * tn + fn + fp)
    present = [v for v, has in ((rec, tp + fn), (spec, tn + fp)) if has]
    return {"ACCURACY": _div(tp + tn, n), "BALANCED_ACCURACY": float(np.mean(present)) if present else 0.0,
            "PRECISION": prec, "RECALL": rec, "F1_SCORE": f1, "F1_SCORE_MACRO": (f1 + f1_neg) / 2}


def roc_auc(y, score):
    y = np.asarray(y).ravel().astype(bool)
    s = np.asarray(score, np.float64).ravel()
    n1, n0 = int(y.sum()), int((~y).sum())
    if n1 == 0 or n0 == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(s, kind="mergesort")
    ranks = np.empty(len(s), np.float64)
    sv = s[order]
    i = 0
    while i < len(sv):  # average ranks over ties
        j = i
        while j + 1 < len(sv) and sv[j + 1] == sv[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[y].sum() - n1 * (n1 + 1) / 2.0) / (n1 * n0))


def average_precision(y, score):
    y = np.asarray(y).ravel().astype(np.int64)
    s = np.asarray(score, np.float64).ravel()
    order = np.argsort(-s, kind="mergesort")
    y, s = y[order], s[order]
    distinct = np.r_[np.where(np.diff(s))[0], len(s) - 1]
    tps = np.cumsum(y)[distinct]
    fps = 1 + distinct - tps
    precision = tps / np.maximum(tps + fps, 1)
    recall = tps / max(int(y.sum()), 1)
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


class Classifmetrics:
    """other.py:86-109."""

    def __init__(self, round_to=5):
        self.round_to = round_to

    def __call__(self, y, y_hat):
        y_hat = np.asarray(y_hat, np.float64)
        m = _binary_scores(y, y_hat.round())
        out = {k: round(m[k], self.round_to) for k in
               ("ACCURACY", "BALANCED_ACCURACY", "PRECISION", "RECALL", "F1_SCORE", "F1_SCORE_MACRO")}
        out["ROC_AUC_SCORE"] = round(roc_auc(y, y_hat), self.round_to)
        out["AP"] = round(average_precision(y, y_hat), self.round_to)
        return out


def nilm_sums_device(y=None, y_hat=None, y_state=None, y_hat_state=None):
    """The twelve sums of ``camal_nilm_sums`` (include/camal_hip.h) for device or host inputs.
    ``y_hat`` / ``y_hat_state``: float32 tensors as ``Ensemble.localize`` returns them (``power``, ``state``);
    ``y``: true power, float32 or float64; ``y_state``: true labels (converted to int8).  -> (sums[12], n)"""
    import ctypes as C

    import torch

    from . import _lib
    lib = _lib.load()
    ref = y_hat if y_hat is not None else y_hat_state
    dev = ref.device if torch.is_tensor(ref) and ref.is_cuda else torch.device("cuda", torch.cuda.current_device())

    def put(a, dtype):
        if a is None:
            return None
        t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=dev, dtype=dtype).contiguous().reshape(-1)

    y32 = y64 = None
    if y is not None:
        is32 = (y.dtype == torch.float32) if torch.is_tensor(y) else (np.asarray(y).dtype == np.float32)
        if is32:
            y32 = put(y, torch.float32)
        else:
            y64 = put(y, torch.float64)
    yh = put(y_hat, torch.float32)
    ys = put(np.asarray(y_state).astype(np.int64) if y_state is not None and not torch.is_tensor(y_state) else y_state,
             torch.int8)
    yhs = put(y_hat_state, torch.float32)
    n = int((yh if yh is not None else yhs).numel())
    with torch.cuda.device(dev):
        ws = torch.empty(int(lib.camal_nilm_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        sums = torch.empty(12, dtype=torch.float64, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.camal_nilm_sums(_lib.ptr(y32), _lib.ptr(y64), _lib.ptr(yh), _lib.ptr(ys), _lib.ptr(yhs), n,
                                       _lib.ptr(sums), _lib.ptr(ws), ws.numel(), stream))
    return sums.cpu().numpy(), n


class NILMmetrics:
    """other.py:112-159."""

    def __init__(self, round_to=3):
        self.round_to = round_to

    def from_sums(self, sums, n, regression=True, classification=True):
        """Same keys and rounding from the twelve sums of ``camal_nilm_sums``."""
        r = self.round_to
        s = [float(v) for v in sums]
        metrics = {}
        if regression:
            mse = s[1] / n
            metrics["MAE"] = round(s[0] / n, r)
            metrics["MSE"] = round(mse, r)
            metrics["RMSE"] = round(float(np.sqrt(mse)), r)
            with np.errstate(divide="ignore", invalid="ignore"):
                metrics["TECA"] = round(float(1 - np.float64(s[0]) / (2 * np.float64(s[2]))), r)
                metrics["NDE"] = round(float(np.float64(s[1]) / np.float64(s[3])), r)
                metrics["SAE"] = round(float(np.abs(np.float64(s[5]) - np.float64(s[4])) / np.float64(s[4])), r)
                metrics["MR"] = round(float(np.float64(s[6]) / np.float64(s[7])), r)
        if classification:
            m = _scores_from_counts(int(round(s[8])), int(round(s[9])), int(round(s[10])), int(round(s[11])))
            for k in ("ACCURACY", "BALANCED_ACCURACY", "PRECISION", "RECALL", "F1_SCORE"):
                metrics[k] = round(m[k], r)
        return metrics

    def __call__(self, y=None, y_hat=None, y_state=None, y_hat_state=None):
        r = self.round_to
        metrics = {}
        if y is not None:
            assert y_hat is not None, "Target y_hat not provided, please provide y_hat to compute regression metrics."
            y = np.nan_to_num(np.asarray(y).astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
            y_hat = np.nan_to_num(np.asarray(y_hat).astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
            mse = float(np.mean((y - y_hat) ** 2))
            metrics["MAE"] = round(float(np.mean(np.abs(y - y_hat))), r)
            metrics["MSE"] = round(mse, r)
            metrics["RMSE"] = round(float(np.sqrt(mse)), r)
            metrics["TECA"] = round(float(1 - (np.sum(np.abs(y_hat - y)) / (2 * np.sum(np.abs(y))))), r)
            metrics["NDE"] = round(float(np.sum((y_hat - y) ** 2) / np.sum(y ** 2)), r)
            metrics["SAE"] = round(float(np.abs(np.sum(y_hat) - np.sum(y)) / np.sum(y)), r)
            metrics["MR"] = round(float(np.sum(np.minimum(y_hat, y)) / np.sum(np.maximum(y_hat, y))), r)
        if y_state is not None:
            assert y_hat_state is not None, "Target y_hat_state not provided, please pass y_hat_state to compute classification metrics."
            ys = np.nan_to_num(np.asarray(y_state).astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
            yh = np.round(np.nan_to_num(np.asarray(y_hat_state).astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0))
            m = _binary_scores(ys, yh)
            for k in ("ACCURACY", "BALANCED_ACCURACY", "PRECISION", "RECALL", "F1_SCORE"):
                metrics[k] = round(m[k], r)
        return metrics
