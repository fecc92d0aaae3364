"""Seeded synthetic inputs shaped like the reference's datasets.

The reference ships no data (``data/*/README.md`` only give download links), so every
test / bench uses these generators.  Values are aggregate power in kW, i.e. after the
reference's ``/1000`` scaling (``configs/config_expes.yaml:15``,
``src/helpers/torch_dataset.py:63-65``).
"""
from __future__ import annotations

import numpy as np

# (low kW, high kW, min minutes, max minutes)
_UKDALE_PULSES = [(2.0, 3.0, 2, 5), (0.8, 2.5, 30, 90)]       # kettle-like, dishwasher-like
_REFIT_PULSES = [(2.0, 3.0, 2, 6), (0.5, 2.5, 20, 120), (0.1, 0.3, 10, 40)]
_IDEAL_PULSES = [(7.0, 9.0, 3, 12), (1.5, 3.0, 2, 6)]          # shower-like, kettle-like


def _windows(n, length, seed, pulses, clip_hi, p_pulse, dtype):
    rng = np.random.default_rng(seed)
    x = np.abs(rng.normal(0.15, 0.05, size=(n, length)))
    labels = np.zeros(n, dtype=np.int64)
    for i in range(n):
        n_p = rng.poisson(p_pulse)
        for _ in range(n_p):
            lo, hi, dmin, dmax = pulses[rng.integers(len(pulses))]
            amp = rng.uniform(lo, hi)
            dur = int(rng.integers(dmin, dmax + 1))
            start = int(rng.integers(0, max(1, length - 1)))
            x[i, start:start + dur] += amp
            labels[i] = 1
    x[x < 0.005] = 0.0  # values < 5 W zeroed (data_processing.py:485,510)
    np.clip(x, 0.0, clip_hi, out=x)
    return x.astype(dtype), labels


def ukdale_windows(n=64, length=1024, seed=0, dtype=np.float32, return_labels=False):
    """BASELINE config 1/2: base load |N(0.15,0.05)| kW + rectangular pulses, clip [0, 6]."""
    x, y = _windows(n, length, seed, _UKDALE_PULSES, 6.0, 1.2, dtype)
    return (x, y) if return_labels else x


def refit_windows(n=64, length=1440, seed=0, dtype=np.float32, return_labels=False):
    """BASELINE config 3: REFIT meter-day windows @1 min, clip [0, 10] kW."""
    x, y = _windows(n, length, seed, _REFIT_PULSES, 10.0, 2.0, dtype)
    return (x, y) if return_labels else x


def ideal_windows(n=128, length=510, seed=0, dtype=np.float32, return_labels=True):
    """BASELINE config 4: IDEAL-shaped training windows (no cutoff), ~50/50 labels."""
    x, y = _windows(n, length, seed, _IDEAL_PULSES, 1e9, 0.7, dtype)
    return (x, y) if return_labels else x


def metric_arrays(seed, n=4000):
    """Flat arrays shaped like what ``CamAL.test`` hands the metrics (core.py:199-231): true / estimated power in W
    (the estimate holds a few NaN / +-inf entries, which the metrics zero), true states 0/1, estimated states as
    the attention module emits them for unsigned inputs (-0, 0, 1; the reference's sklearn calls refuse -1), window labels and detection scores with ties.
    -> (y, y_hat, y_state, y_hat_state, y_clf, score)"""
    rng = np.random.default_rng(seed)
    y_state = (rng.random(n) < 0.3).astype(np.int64)
    y = y_state * rng.uniform(500.0, 2500.0, n)
    flip = rng.random(n) < 0.15
    hat_on = np.where(flip, 1 - y_state, y_state)
    y_hat_state = hat_on.astype(np.float32)
    neg = rng.random(n) < 0.02
    y_hat_state[neg & (hat_on == 0)] = -0.0
    y_hat = np.minimum(np.maximum(y_hat_state, 0) * 2000.0, rng.uniform(100.0, 6000.0, n)).astype(np.float32)
    bad = rng.choice(n, 6, replace=False)
    y_hat[bad[:2]] = np.nan
    y_hat[bad[2:4]] = np.inf
    y_hat[bad[4:]] = -np.inf
    m = n // 16
    y_clf = (rng.random(m) < 0.4).astype(np.int64)
    score = np.round(np.clip(0.5 + 0.35 * (2 * y_clf - 1) * rng.random(m) + 0.2 * rng.normal(size=m), 0, 1), 2)
    return y, y_hat, y_state, y_hat_state, y_clf, score


def house_series(T=20000, seed=0, n_app=1, dtype=np.float64, nan_runs=6):
    """A long series as ``_get_stems`` lays it out (data_processing.py:366-367): row 0 aggregate power in W, then
    (power, status) per appliance, with a few runs of NaN (meter gaps) that the window loop must drop."""
    rng = np.random.default_rng(seed)
    rows = np.zeros((1 + 2 * n_app, T))
    rows[0] = np.abs(rng.normal(150.0, 50.0, T))
    for a in range(n_app):
        p = np.zeros(T)
        for _ in range(max(1, T // 1500)):
            start = int(rng.integers(0, T - 50))
            dur = int(rng.integers(3, 40))
            p[start:start + dur] = rng.uniform(800.0, 2500.0)
        rows[1 + 2 * a] = p
        rows[2 + 2 * a] = (p > 0).astype(float)
        rows[0] += p
    for _ in range(nan_runs):
        start = int(rng.integers(0, T - 30))
        r = int(rng.integers(0, 1 + 2 * n_app))
        rows[r, start:start + int(rng.integers(1, 25))] = np.nan
    return rows.astype(dtype)
