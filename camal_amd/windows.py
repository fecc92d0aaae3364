"""Window builder, scaler and weak labels in front of ``CamAL.predict`` / ``test``.

``NILMscaler`` mirrors the reference's scaler (``src/helpers/torch_dataset.py:22-156``) for host 4-D arrays
(its statistics are what the device path needs); ``build_windows`` does, on the device and in one call of
``camal_build_windows``, what the reference does with the window loop of ``*_DataBuilder.get_nilm_dataset``
(``src/helpers/data_processing.py:377-400``), ``NILMscaler.transform`` on the aggregate channel and
``nilmdataset_to_clfdataset`` (``data_processing.py:17-33``): the windows ``CamAL.predict`` is fed never exist
as a host array.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

_TYPES = ["StandardScaling", "MinMax", "MeanScaling", "MeanMaxScaling", "MaxScaling"]


class NILMscaler:
    """torch_dataset.py:22-156 (temperature scaling and the tensor helpers are out of the path)."""

    def __init__(self, power_scaling_type="StandardScaling", appliance_scaling_type=None):
        if not isinstance(power_scaling_type, int):
            assert power_scaling_type in _TYPES
        if not isinstance(appliance_scaling_type, int):
            # like the reference (:37-38), None does not pass: NILMscaler() without an appliance scaling raises
            assert appliance_scaling_type in _TYPES + ["SameAsPower"]
        self.power_scaling_type = power_scaling_type
        self.appliance_scaling_type = appliance_scaling_type
        self.n_appliance = 0
        self.is_fitted = False
        self.power_stat1 = 0
        self.power_stat2 = 0
        self.appliance_stat1 = []
        self.appliance_stat2 = []

    def __call__(self, data_train, data_test):
        return self.fit_transform(data_train.copy()), self.transform(data_test.copy())

    @staticmethod
    def _stats(kind, v, own):
        """(stat1, stat2) as torch_dataset.py:63-99 computes them.  'MinMaxScaling' is the name fit() tests for
        the aggregate (:69) while the constructor only admits 'MinMax' (:35): that branch is dead there and
        the statistics stay 0; the appliance branch tests 'MinMax' (:88) and is live."""
        if isinstance(kind, int):
            return 0, kind
        if kind == "StandardScaling":
            return v.mean(), v.std()
        if kind == "MinMax":
            return (v.min(), v.max()) if own == "appliance" else (0, 0)
        if kind == "MeanScaling":
            return (0, v.mean()) if own == "power" else None
        if kind == "MeanMaxScaling":
            return v.mean(), v.max()
        if kind == "MaxScaling":
            return 0, v.max()
        return None

    def fit(self, data):
        self.power_stat1, self.power_stat2 = self._stats(self.power_scaling_type, data[:, 0, 0, :], "power")
        if self.appliance_scaling_type is not None:
            self.n_appliance = data.shape[1] - 1
            if self.appliance_scaling_type == "SameAsPower":
                st = (self.power_stat1, self.power_stat2)
            else:
                st = self._stats(self.appliance_scaling_type, data[:, 1, 0, :], "appliance")
            if st is not None:
                self.appliance_stat1.append(st[0])
                self.appliance_stat2.append(st[1])
        self.is_fitted = True

    def _minmax(self, which):
        if which == "power":
            return self.power_scaling_type == "MinMax"
        return self.appliance_scaling_type == "MinMax" or (
            self.appliance_scaling_type == "SameAsPower" and self.power_scaling_type == "MinMax")

    def power_affine(self):
        """(stat1, divisor): scaled = (x - stat1) / divisor for the aggregate channel (:108-111)."""
        s1, s2 = self.power_stat1, self.power_stat2
        return float(s1), float(s2 - s1) if self._minmax("power") else float(s2)

    def transform(self, data):
        assert self.is_fitted, "Not fitted yet."
        s1, div = self.power_stat1, (self.power_stat2 - self.power_stat1) if self._minmax("power") else self.power_stat2
        data[:, 0, 0, :] = (data[:, 0, 0, :] - s1) / div
        if self.appliance_scaling_type is not None:
            for n_app in range(1, self.n_appliance + 1):
                a1, a2 = self.appliance_stat1[n_app - 1], self.appliance_stat2[n_app - 1]
                data[:, n_app, 0, :] = (data[:, n_app, 0, :] - a1) / ((a2 - a1) if self._minmax("appliance") else a2)
        return data

    def fit_transform(self, data):
        self.fit(data)
        return self.transform(data)

    def inverse_transform(self, data):
        r = data.copy()
        assert len(r.shape) < 5, "Data containing too many dimensions (>5)."
        flag = len(r.shape) < 4
        if flag:
            assert len(data.shape) == 3, "At lest non batched data (3D array) to inverse transform."
            r = np.expand_dims(r, axis=0)
        s1, s2 = self.power_stat1, self.power_stat2
        r[:, 0, 0, :] = r[:, 0, 0, :] * ((s2 - s1) if self._minmax("power") else s2) + s1
        if self.appliance_scaling_type is not None:
            for n_app in range(1, self.n_appliance + 1):
                a1, a2 = self.appliance_stat1[n_app - 1], self.appliance_stat2[n_app - 1]
                r[:, n_app, 0, :] = r[:, n_app, 0, :] * ((a2 - a1) if self._minmax("appliance") else a2) + a1
        return r[0] if flag else r


    def _channel_inverse(self, data, stats):
        """data [n, C, L] (or [C, L]): channel c -> x * mul_c + add_c for the (mul, add) pairs of ``stats``; NumPy or torch."""
        try:
            import torch
            is_t = torch.is_tensor(data)
        except ImportError:
            is_t = False
        r = data.clone() if is_t else data.copy()
        if len(r.shape) == 2:
            r = r.unsqueeze(0) if is_t else np.expand_dims(r, axis=0)
        for c, (mul, add) in enumerate(stats):
            r[:, c, :] = r[:, c, :] * mul + add
        return r

    def inverse_transform_appliance(self, data):
        """torch_dataset.py:158-176: appliance power channels [n, M, L] back to watts."""
        assert self.appliance_scaling_type is not None, "Original data : no scaling apply on appliance channels!"
        mm = self._minmax("appliance")
        return self._channel_inverse(data, [((a2 - a1) if mm else a2, a1)
                                            for a1, a2 in zip(self.appliance_stat1[:self.n_appliance], self.appliance_stat2)])

    def inverse_transform_agg_power(self, data):
        """torch_dataset.py:178-193: channel 0 of [n, C, L] back to watts with the aggregate's statistics (the reference
        picks the MinMax form from the APPLIANCE scaling type there, :189: kept)."""
        s1, s2 = self.power_stat1, self.power_stat2
        return self._channel_inverse(data, [((s2 - s1) if self._minmax("appliance") else s2, s1)])


def n_windows(T, window_size, window_stride):
    """data_processing.py:377-380."""
    if T < window_size:
        return 0
    return T // window_stride if window_size == window_stride else 1 + (T - window_size) // window_stride


def build_windows(series, window_size, window_stride=None, scaler=None, appliance=1, device="cuda", want=("x",)):
    """series [n_rows][T] (NumPy or torch; float32 or float64): row 0 the aggregate, then (power, status) per
    appliance.  -> dict of device tensors trimmed to the surviving windows: always ``x`` [n][W] float32 (scaled
    by ``scaler``'s aggregate statistics, identity without one) and ``count``; on request (``want``): ``x64``,
    ``label`` (int8 weak labels of ``appliance``), ``app_power``, ``app_state``, ``index``."""
    import torch

    from . import _lib
    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.CamalHipError(f"camal_amd builds windows on MI355X only (device={device!r}); there is no CPU path.")
    dev = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
    t = series if torch.is_tensor(series) else torch.from_numpy(np.ascontiguousarray(series))
    if t.dtype not in (torch.float32, torch.float64):
        t = t.double()
    t = t.to(dev).contiguous()
    assert t.dim() == 2, "series is [n_rows][T]"
    n_rows, T = t.shape
    W = int(window_size)
    S = int(window_stride) if window_stride is not None else W
    n = int(lib.camal_windows_count(T, W, S))
    if n < 1:
        raise ValueError("the series is shorter than one window")
    s1, div = scaler.power_affine() if scaler is not None else (0.0, 1.0)
    with torch.cuda.device(dev):
        out = {"x": torch.empty((n, W), dtype=torch.float32, device=dev)}
        if "x64" in want:
            out["x64"] = torch.empty((n, W), dtype=torch.float64, device=dev)
        if "label" in want:
            out["label"] = torch.empty(n, dtype=torch.int8, device=dev)
        if "app_power" in want:
            out["app_power"] = torch.empty((n, W), dtype=torch.float32, device=dev)
        if "app_state" in want:
            out["app_state"] = torch.empty((n, W), dtype=torch.int8, device=dev)
        if "index" in want:
            out["index"] = torch.empty(n, dtype=torch.int32, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = torch.empty(int(lib.camal_windows_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.camal_build_windows(
            _lib.ptr(t), int(t.dtype == torch.float64), n_rows, T, W, S, int(appliance), float(s1), float(div),
            _lib.ptr(out["x"]), _lib.ptr(out.get("x64")), _lib.ptr(out.get("label")), _lib.ptr(out.get("app_power")),
            _lib.ptr(out.get("app_state")), _lib.ptr(out.get("index")), _lib.ptr(count), _lib.ptr(ws), ws.numel(), stream))
        k = int(count.item())
    res = {name: v[:k] for name, v in out.items()}
    res["count"] = k
    return res
