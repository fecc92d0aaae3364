"""Device-side ensemble: owns one ``camal_ctx`` (packed weights + workspace) per GPU.

Torch is used only as the device-memory / stream provider: tensors' ``data_ptr()`` and the
current HIP stream are handed to the C ABI (include/camal_hip.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, weights


def folded_blob(sd) -> np.ndarray:
    """State dict (reference key names) -> the float32 blob ``camal_set_member_weights`` takes.

    BatchNorm (eval) is folded into each conv here, in float64; order documented
    in include/camal_hip.h.
    """
    sd = weights.to_numpy_state_dict(sd)
    k = weights.kernel_size_of(sd)
    mid = mid_channels_of(sd)
    parts = []
    for prefix, _ci, _co, _k in weights.conv_layers(k, mid_channels=mid, arch=weights.arch_of(sd)):
        w, b = weights.fold_conv_bn(
            sd[prefix + ".layers.0.weight"], sd[prefix + ".layers.0.bias"],
            sd[prefix + ".layers.1.weight"], sd[prefix + ".layers.1.bias"],
            sd[prefix + ".layers.1.running_mean"], sd[prefix + ".layers.1.running_var"])
        parts += [w.ravel(), b.ravel()]
    for prefix, _ci, _co in weights.residual_layers(mid_channels=mid):
        w, b = weights.fold_conv_bn(
            sd[prefix + ".0.weight"], sd[prefix + ".0.bias"],
            sd[prefix + ".1.weight"], sd[prefix + ".1.bias"],
            sd[prefix + ".1.running_mean"], sd[prefix + ".1.running_var"])
        parts += [w.ravel(), b.ravel()]
    parts += [np.asarray(sd["linear.weight"], np.float32).ravel(), np.asarray(sd["linear.bias"], np.float32).ravel()]
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)


SUPPORTED_MID_CHANNELS = (32, 64, 128)


def mid_channels_of(sd) -> int:
    """mid_channels of a state dict (camal_resnet.py:8,14-16), checked against what the library has kernels for:
    in_channels = 1, nb_class = 2, mid_channels 32 / 64 / 128 (64 is what CamAL.eval() always builds, core.py:386-390)."""
    w0 = np.asarray(sd["layers.0.layers.0.layers.0.weight"])
    mid = int(w0.shape[0])
    if w0.shape[1] != 1 or tuple(np.asarray(sd["linear.weight"]).shape) != (2, 2 * mid) or mid not in SUPPORTED_MID_CHANNELS:
        raise NotImplementedError(
            "the HIP path implements in_channels=1, nb_class=2 and mid_channels in (32, 64, 128); "
            f"got first conv {tuple(w0.shape)}, linear {tuple(np.asarray(sd['linear.weight']).shape)}")
    return mid


def _device_index(device) -> int:
    d = torch.device(device)
    if d.type != "cuda":
        raise _lib.CamalHipError(
            f"camal_amd runs on MI355X only (device={device!r}); there is no CPU path. "
            "Use device='cuda' / 'cuda:N'.")
    return d.index if d.index is not None else torch.cuda.current_device()


class Ensemble:
    """E ResNet members resident on one GPU."""

    def __init__(self, state_dicts, device="cuda", precision="fp32", chunk_windows=None):
        self.lib = _lib.load()
        self.dev_index = _device_index(device)
        self.device = torch.device("cuda", self.dev_index)
        self.precision = {"fp32": _lib.FP32, "bf16": _lib.BF16}[precision]
        self.precision_name = precision
        sds = [weights.to_numpy_state_dict(sd) for sd in state_dicts]
        self.kernel_sizes = [weights.kernel_size_of(sd) for sd in sds]
        archs = {weights.arch_of(sd) for sd in sds}
        if len(archs) != 1:
            raise ValueError("the members of one ensemble share one classifier architecture")
        self.arch = archs.pop()
        arch_id = {weights.ARCH_CAMAL: _lib.ARCH_CAMAL, weights.ARCH_RESNET: _lib.ARCH_RESNET}[self.arch]
        mids = {mid_channels_of(sd) for sd in sds}
        if len(mids) != 1:
            raise ValueError("the members of one ensemble share one mid_channels")
        self.mid_channels = mids.pop()
        self.E = len(sds)
        ks = (C.c_int * self.E)(*self.kernel_sizes)
        h = C.c_void_p()
        _lib.check(self.lib.camal_create_widths(self.dev_index, self.E, ks, self.precision, arch_id, self.mid_channels,
                                                C.byref(h)))
        self._h = h
        for e, sd in enumerate(sds):
            blob = folded_blob(sd)
            assert blob.size == self.lib.camal_member_blob_floats_widths(self.kernel_sizes[e], arch_id, self.mid_channels)
            _lib.check(self.lib.camal_set_member_weights(
                self._h, e, blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size))
        if chunk_windows:
            self.set_chunk_windows(chunk_windows)

    def close(self):
        pipe = getattr(self, "_pipe", None)
        if pipe is not None:
            pipe[2].shutdown(wait=True)
            self._pipe, self._pipe_key = None, None
        if getattr(self, "_h", None):
            self.lib.camal_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_fused(self, fused=True):
        """True (default): fused-block kernels for blocks 2/3; False: one kernel per ConvBlock."""
        _lib.check(self.lib.camal_set_fused(self._h, int(bool(fused))))

    def reserve(self, B, L):
        """Allocate the workspace for calls of up to B windows of L steps now (later calls then allocate nothing)."""
        _lib.check(self.lib.camal_reserve(self._h, int(B), int(L)))

    def set_chunk_windows(self, n):
        _lib.check(self.lib.camal_set_chunk_windows(self._h, int(n)))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _as_x(self, x):
        x = torch.as_tensor(x)
        if x.dim() == 1:
            x = x[None, :]
        if x.dim() == 3 and x.shape[1] == 1:
            x = x[:, 0, :]
        if x.dim() != 2:
            raise ValueError(f"expected [B,L] or [B,1,L] windows, got shape {tuple(x.shape)}")
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    # --- CamALResNet.forward + CAM.run for a batch (camal_forward_cam) -------------------
    def forward_cam(self, x, member=-1, want_cam=True, cam_label=1, tap_block=2):
        """-> (logits [E,B,2] float32, cam [E,B,L] float32 or None) as device tensors.  tap_block: the ResNetBlock the
        CAM is taken from (2 = the last one; 1 = the second, ``CAM(last_conv_layer=layers[1])``)."""
        x = self._as_x(x)
        B, L = x.shape
        eo = self.E if member < 0 else 1
        with torch.cuda.device(self.device):
            logits = torch.empty((eo, B, 2), dtype=torch.float32, device=self.device)
            cam = torch.empty((eo, B, L), dtype=torch.float32, device=self.device) if want_cam else None
            _lib.check(self.lib.camal_forward_cam_tap(self._h, _lib.ptr(x), B, L, member, cam_label, int(tap_block),
                                                      _lib.ptr(logits), _lib.ptr(cam), self._stream()))
        return logits, cam

    # --- CamAL.predict body + power step (camal_localize) ---------------------------------
    def localize(self, x, x64=None, y=None, w_ma=5, gate_mode=0, pred_f64=False, p_on=None,
                 scale_mul=1.0, scale_add=0.0, want_avg_cam=False, want_members=False, out=None):
        """Returns a dict of device tensors: prob [B] f64, state [B,L] f32, and optionally
        power / avg_cam / member_logits / member_cam."""
        x = self._as_x(x)
        B, L = x.shape
        dev = self.device
        with torch.cuda.device(dev):
            if x64 is not None:
                x64 = torch.as_tensor(x64).reshape(B, L).to(device=dev, dtype=torch.float64).contiguous()
            if y is not None:
                y = torch.as_tensor(np.asarray(y).astype(np.int8) if not torch.is_tensor(y) else y)
                y = y.to(device=dev, dtype=torch.int8).contiguous()
                if y.numel() != B:
                    raise ValueError("y must hold one label per window")
            o = out if out is not None else {}
            prob = o.get("prob") if "prob" in o else torch.empty((B,), dtype=torch.float64, device=dev)
            state = o.get("state") if "state" in o else torch.empty((B, L), dtype=torch.float32, device=dev)
            power = None
            if p_on is not None:
                power = o.get("power") if "power" in o else torch.empty((B, L), dtype=torch.float32, device=dev)
            avg = torch.empty((B, L), dtype=torch.float32, device=dev) if want_avg_cam else None
            mlog = torch.empty((self.E, B, 2), dtype=torch.float32, device=dev) if want_members else None
            mcam = torch.empty((self.E, B, L), dtype=torch.float32, device=dev) if want_members else None
            _lib.check(self.lib.camal_localize(
                self._h, _lib.ptr(x), _lib.ptr(x64), _lib.ptr(y), B, L, int(w_ma), int(gate_mode), int(bool(pred_f64)),
                float(p_on if p_on is not None else 0.0), float(scale_mul), float(scale_add),
                _lib.ptr(prob), _lib.ptr(avg), _lib.ptr(state), _lib.ptr(power), _lib.ptr(mlog), _lib.ptr(mcam),
                self._stream()))
        res = {"prob": prob, "state": state}
        if power is not None:
            res["power"] = power
        if avg is not None:
            res["avg_cam"] = avg
        if want_members:
            res["member_logits"], res["member_cam"] = mlog, mcam
        return res

    # --- the same with the states as one-byte codes (camal_localize_i8): what the host pipeline ships over PCIe ------
    def localize_i8(self, x, prob, state8, power=None, x64=None, y=None, w_ma=5, gate_mode=0, pred_f64=False, p_on=None,
                    scale_mul=1.0, scale_add=0.0):
        """``camal_localize_i8`` on device tensors the caller owns (x [B,L] float32, prob [B] float64, state8 [B,L] int8,
        power [B,L] float32 or None): allocates nothing."""
        B, L = x.shape
        with torch.cuda.device(self.device):
            _lib.check(self.lib.camal_localize_i8(
                self._h, _lib.ptr(x), _lib.ptr(x64), _lib.ptr(y), B, L, int(w_ma), int(gate_mode), int(bool(pred_f64)),
                float(p_on if p_on is not None else 0.0), float(scale_mul), float(scale_add),
                _lib.ptr(prob), _lib.ptr(state8), _lib.ptr(power), self._stream()))

    # --- NumPy in / NumPy out for N >> what stays resident: chunks through pinned buffers, three streams ---------
    NSETS = 3  # buffer sets of the host pipeline: one chunk computing, one on its way out, one being widened on the host

    def stream_chunk_windows(self, L):
        """Windows per chunk of ``localize_host`` when the caller does not say: about 6 M timesteps (4096 day
        windows, 585 week windows): a chunk's kernels run for tens of milliseconds, its copies for well under one."""
        return max(64, min(8192, (6 << 20) // max(int(L), 1)))

    def _host_pipeline(self, chunk, L, is64, want_power):
        """NSETS sets of pinned host / device buffers for ``localize_host``, kept for the next call of the same shape
        (pinning memory costs more than a chunk's work), the three streams, the worker threads that widen results into
        the caller's arrays; the context's workspace is sized once here (camal_reserve) so that no call allocates."""
        key = (chunk, L, bool(is64), bool(want_power))
        if getattr(self, "_pipe_key", None) == key:
            return self._pipe
        from concurrent.futures import ThreadPoolExecutor
        dev = self.device
        sets = []
        with torch.cuda.device(dev):
            for _ in range(self.NSETS):
                b = {"in32": torch.empty((chunk, L), dtype=torch.float32).pin_memory(),
                     "d_in32": torch.empty((chunk, L), dtype=torch.float32, device=dev),
                     "d_state8": torch.empty((chunk, L), dtype=torch.int8, device=dev),
                     "d_prob": torch.empty((chunk,), dtype=torch.float64, device=dev),
                     "state8": torch.empty((chunk, L), dtype=torch.int8).pin_memory(),
                     "prob": torch.empty((chunk,), dtype=torch.float64).pin_memory(),
                     "h2d": torch.cuda.Event(), "comp": torch.cuda.Event(), "d2h": torch.cuda.Event(), "jobs": []}
                if is64:
                    b["in64"] = torch.empty((chunk, L), dtype=torch.float64).pin_memory()
                    b["d_in64"] = torch.empty((chunk, L), dtype=torch.float64, device=dev)
                if want_power:
                    b["d_power"] = torch.empty((chunk, L), dtype=torch.float32, device=dev)
                    b["power32"] = torch.empty((chunk, L), dtype=torch.float32).pin_memory()
                sets.append(b)
            streams = (torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev))
        self.reserve(chunk, L)
        old = getattr(self, "_pipe", None)
        if old is not None:
            old[2].shutdown(wait=True)
        pool = ThreadPoolExecutor(max_workers=self.host_workers, thread_name_prefix="camal-widen")
        self._pipe_key, self._pipe = key, (sets, streams, pool)
        return self._pipe

    host_workers = 4  # threads that widen a chunk's int8 states / float32 power into the caller's arrays

    @staticmethod
    def _widen_states(dst, codes):
        """int8 state codes (0, 1, -1, -128 = -0.0: camal_localize_i8) -> the caller's array, any dtype.  float dtypes get
        np.round's -0.0 back (core.py:437); integer dtypes have no -0: it becomes 0."""
        np.copyto(dst, codes, casting="unsafe")
        if codes.size and codes.min() == -128:  # only with negative (signed-scaled) inputs
            neg = codes == -128
            dst[neg] = -0.0 if np.issubdtype(dst.dtype, np.floating) else 0

    def localize_host(self, data, y=None, w_ma=5, gate_mode=0, p_on=None, scale_mul=1.0, scale_add=0.0, chunk_windows=None,
                      out=None, states_dtype=np.float64, power_dtype=np.float64):
        """``localize`` for host arrays of any length: ``data`` [N,L] NumPy (float32, or anything else = the float64
        path of the reference, core.py:286) -> (states [N,L], prob [N] float64, power [N,L] or None) as NumPy arrays.

        The windows go through the device in chunks: chunk i+1's host-to-device copy and chunk i-1's device-to-host
        copy run on their own streams under chunk i's kernels, through pinned buffers, so the device holds NSETS
        chunks, not N windows (configs[2]: 10 M day windows).  What crosses PCIe is compact — float32 in, ONE byte
        per state (camal_localize_i8) and float32 power out — and worker threads widen it into the result arrays
        while the next chunks run.  The reference's contract is float64 states (``zeros_like(data)`` of a float64
        array, core.py:286): that is the default; ``states_dtype`` / ``power_dtype`` (e.g. int8 / float32: 5 instead
        of 16 bytes per timestamp of host memory) and ``out`` = {"states", "prob", "power"} arrays to fill in place
        (np.memmap included) keep the host side bounded at that scale.  Same numbers as one call of ``localize`` on
        everything: windows are independent.  Not re-entrant: one call at a time per Ensemble (the pinned buffers and
        streams are the Ensemble's; several GPUs = several Ensembles, ``localize_host_sharded``)."""
        data = np.asarray(data)
        N, L = data.shape
        is64 = data.dtype != np.float32
        want_power = p_on is not None
        chunk = int(chunk_windows or self.stream_chunk_windows(L))
        chunk = max(1, min(chunk, N))
        sets, (s_h2d, s_comp, s_d2h), pool = self._host_pipeline(chunk, L, is64, want_power)
        out = out or {}
        states = out["states"] if out.get("states") is not None else np.empty((N, L), dtype=states_dtype)
        prob = out["prob"] if out.get("prob") is not None else np.empty((N,), dtype=np.float64)
        power = None
        if want_power:
            power = out["power"] if out.get("power") is not None else np.empty((N, L), dtype=power_dtype)
        if states.shape != (N, L) or prob.shape[0] != N or (want_power and power.shape != (N, L)):
            raise ValueError("out arrays must be states [N,L], prob [N] (or [N,1]), power [N,L]")
        if not prob.flags.c_contiguous:
            raise ValueError("out['prob'] must be contiguous")
        prob_flat = prob.reshape(-1)
        if y is not None:
            y = np.asarray(y).astype(np.int8).ravel()
            if y.size != N:
                raise ValueError("y must hold one label per window")
        nw = max(1, int(self.host_workers))

        def widen(b, s, r0, r1):  # rows [r0, r1) of the chunk at window s: pinned buffers -> the result arrays
            b["d2h"].synchronize()
            self._widen_states(states[s + r0:s + r1], b["state8"][r0:r1].numpy())
            if want_power:
                np.copyto(power[s + r0:s + r1], b["power32"][r0:r1].numpy(), casting="same_kind")
            if r0 == 0:
                n = b["n"]
                prob_flat[s:s + n] = b["prob"][:n].numpy()

        def settle(b):  # the chunk that used this set NSETS steps ago has left its pinned buffers
            for f in b["jobs"]:
                f.result()
            b["jobs"] = []

        cur = torch.cuda.current_stream(self.device)
        for b in sets:
            settle(b)
        with torch.cuda.device(self.device):
            s_h2d.wait_stream(cur)
            s_comp.wait_stream(cur)
            for i, s in enumerate(range(0, N, chunk)):
                n = min(chunk, N - s)
                b = sets[i % self.NSETS]
                settle(b)
                # host: stage the chunk in pinned memory (the float32 model input; the float64 original beside it)
                b["in32"][:n].numpy()[...] = data[s:s + n]
                if is64:
                    b["in64"][:n].numpy()[...] = data[s:s + n]
                with torch.cuda.stream(s_h2d):
                    s_h2d.wait_event(b["comp"])  # the kernels of the chunk that used this set have read the device inputs
                    b["d_in32"][:n].copy_(b["in32"][:n], non_blocking=True)
                    if is64:
                        b["d_in64"][:n].copy_(b["in64"][:n], non_blocking=True)
                    b["h2d"].record(s_h2d)
                with torch.cuda.stream(s_comp):
                    s_comp.wait_event(b["h2d"])
                    s_comp.wait_event(b["d2h"])  # that chunk's results have left the device outputs
                    yd = None if y is None else torch.from_numpy(y[s:s + n]).to(self.device)
                    self.localize_i8(b["d_in32"][:n], b["d_prob"][:n], b["d_state8"][:n],
                                     power=b["d_power"][:n] if want_power else None, x64=b["d_in64"][:n] if is64 else None,
                                     y=yd, w_ma=w_ma, gate_mode=gate_mode, pred_f64=is64, p_on=p_on, scale_mul=scale_mul,
                                     scale_add=scale_add)
                    b["comp"].record(s_comp)
                with torch.cuda.stream(s_d2h):
                    s_d2h.wait_event(b["comp"])
                    b["state8"][:n].copy_(b["d_state8"][:n], non_blocking=True)
                    b["prob"][:n].copy_(b["d_prob"][:n], non_blocking=True)
                    if want_power:
                        b["power32"][:n].copy_(b["d_power"][:n], non_blocking=True)
                    b["d2h"].record(s_d2h)
                b["n"] = n
                per = -(-n // nw)
                b["jobs"] = [pool.submit(widen, b, s, r0, min(n, r0 + per)) for r0 in range(0, n, per)]
            for b in sets:
                settle(b)
            cur.wait_stream(s_comp)
        return states, prob, power

    # --- CamAL.test at any N: windows, true powers and true states stream through the device, twelve sums come back ---
    def test_sums_host(self, data, y=None, y_state=None, w_ma=5, gate_mode=0, p_on=None, scale_mul=1.0, scale_add=0.0,
                       y_mul=1.0, y_add=0.0, chunk_windows=None):
        """The sums behind ``NILMmetrics`` as ``CamAL.test`` needs them (core.py:199-231), for host arrays of any length:
        ``data`` [N,L] (the scaled aggregate), ``y`` [N,L] true appliance power in the SCALED units of the data set
        (``y * y_mul + y_add`` = NILMscaler.inverse_transform of that channel, applied chunk by chunk while staging; None:
        no regression sums), ``y_state`` [N,L] true on/off labels.  Chunks go host -> device through pinned buffers under
        the previous chunk's kernels; states and powers never leave the device, each chunk returns twelve doubles and its
        windows' probabilities, added up on the host in chunk order (deterministic).
        -> (sums [12] float64 as camal_nilm_sums, number of timesteps, prob [N] float64)."""
        data = np.asarray(data)
        N, L = data.shape
        is64 = data.dtype != np.float32
        want_power = p_on is not None and y is not None
        chunk = max(1, min(int(chunk_windows or self.stream_chunk_windows(L)), N))
        y64 = y is not None and np.asarray(y).dtype != np.float32
        key = ("test", chunk, L, is64, y is not None, y64, y_state is not None)
        dev = self.device
        if getattr(self, "_tpipe_key", None) != key:
            sets = []
            with torch.cuda.device(dev):
                for _ in range(2):
                    b = {"in32": torch.empty((chunk, L), dtype=torch.float32).pin_memory(),
                         "d_in32": torch.empty((chunk, L), dtype=torch.float32, device=dev),
                         "d_state": torch.empty((chunk, L), dtype=torch.float32, device=dev),
                         "d_power": torch.empty((chunk, L), dtype=torch.float32, device=dev),
                         "d_prob": torch.empty((chunk,), dtype=torch.float64, device=dev),
                         "prob": torch.empty((chunk,), dtype=torch.float64).pin_memory(),
                         "d_sums": torch.empty(12, dtype=torch.float64, device=dev),
                         "sums": torch.empty(12, dtype=torch.float64).pin_memory(),
                         "ws": torch.empty(int(self.lib.camal_nilm_workspace_bytes(chunk * L)), dtype=torch.uint8, device=dev),
                         "h2d": torch.cuda.Event(), "done": torch.cuda.Event(), "n": 0, "s": 0}
                    if is64:
                        b["in64"] = torch.empty((chunk, L), dtype=torch.float64).pin_memory()
                        b["d_in64"] = torch.empty((chunk, L), dtype=torch.float64, device=dev)
                    if y is not None:
                        ydt = torch.float64 if y64 else torch.float32
                        b["y"] = torch.empty((chunk, L), dtype=ydt).pin_memory()
                        b["d_y"] = torch.empty((chunk, L), dtype=ydt, device=dev)
                    if y_state is not None:
                        b["ys"] = torch.empty((chunk, L), dtype=torch.int8).pin_memory()
                        b["d_ys"] = torch.empty((chunk, L), dtype=torch.int8, device=dev)
                    sets.append(b)
                streams = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
            self.reserve(chunk, L)
            self._tpipe_key, self._tpipe = key, (sets, streams)
        sets, (s_h2d, s_comp) = self._tpipe
        total = np.zeros(12, dtype=np.float64)
        prob = np.empty((N,), dtype=np.float64)

        def collect(b):
            if b["n"] == 0:
                return
            b["done"].synchronize()
            total[...] += b["sums"].numpy()
            prob[b["s"]:b["s"] + b["n"]] = b["prob"][:b["n"]].numpy()
            b["n"] = 0

        cur = torch.cuda.current_stream(dev)
        for b in sets:
            b["n"] = 0
        with torch.cuda.device(dev):
            s_h2d.wait_stream(cur)
            s_comp.wait_stream(cur)
            for i, s in enumerate(range(0, N, chunk)):
                n = min(chunk, N - s)
                b = sets[i % 2]
                collect(b)  # the chunk that used this set two steps ago is done: its buffers are free
                b["in32"][:n].numpy()[...] = data[s:s + n]
                if is64:
                    b["in64"][:n].numpy()[...] = data[s:s + n]
                if y is not None:
                    yp = b["y"][:n].numpy()
                    np.multiply(y[s:s + n], y_mul, out=yp, casting="unsafe")
                    if y_add != 0.0:
                        yp += y_add
                if y_state is not None:
                    b["ys"][:n].numpy()[...] = np.asarray(y_state[s:s + n]).astype(np.int64, copy=False)
                with torch.cuda.stream(s_h2d):
                    for k in ("in32", "in64", "y", "ys"):
                        if k in b:
                            b["d_" + ("in32" if k == "in32" else "in64" if k == "in64" else k)][:n].copy_(b[k][:n], non_blocking=True)
                    b["h2d"].record(s_h2d)
                with torch.cuda.stream(s_comp):
                    s_comp.wait_event(b["h2d"])
                    out = {"prob": b["d_prob"][:n], "state": b["d_state"][:n]}
                    if want_power:
                        out["power"] = b["d_power"][:n]
                    self.localize(b["d_in32"][:n], x64=b["d_in64"][:n] if is64 else None, w_ma=w_ma, gate_mode=gate_mode,
                                  pred_f64=is64, p_on=p_on if want_power else None, scale_mul=scale_mul, scale_add=scale_add, out=out)
                    yd = b["d_y"][:n] if want_power else None
                    _lib.check(self.lib.camal_nilm_sums(
                        _lib.ptr(yd) if (yd is not None and not y64) else None, _lib.ptr(yd) if (yd is not None and y64) else None,
                        _lib.ptr(b["d_power"][:n]) if want_power else None, _lib.ptr(b["d_ys"][:n]) if y_state is not None else None,
                        _lib.ptr(b["d_state"][:n]) if y_state is not None else None, n * L, _lib.ptr(b["d_sums"]), _lib.ptr(b["ws"]),
                        b["ws"].numel(), self._stream()))
                    b["sums"].copy_(b["d_sums"], non_blocking=True)
                    b["prob"][:n].copy_(b["d_prob"][:n], non_blocking=True)
                    b["done"].record(s_comp)  # collect() waits for it on the host before this set is staged again
                b["n"], b["s"] = n, s
            nchunks = (N + chunk - 1) // chunk
            for k in range(2):  # oldest first: the sums are added in chunk order
                collect(sets[(nchunks + k) % 2])
            cur.wait_stream(s_comp)
        return total, N * L, prob

    # --- profiling ---------------------------------------------------------------------------
    def set_profiling(self, enabled=True):
        _lib.check(self.lib.camal_set_profiling(self._h, int(bool(enabled))))

    def get_profile(self):
        """{class: (launches, total_ms)} since profiling was enabled (synchronises the events)."""
        n = (C.c_int * len(_lib.PROF_CLASSES))()
        ms = (C.c_double * len(_lib.PROF_CLASSES))()
        _lib.check(self.lib.camal_get_profile(self._h, n, ms))
        return {name: (int(n[i]), float(ms[i])) for i, name in enumerate(_lib.PROF_CLASSES)}


def localize_host_sharded(ensembles, data, out=None, states_dtype=np.float64, power_dtype=np.float64, **kw):
    """``Ensemble.localize_host`` over several GPUs from ONE process (the reference's all_gpu idea for inference: a plain
    ``python script.py`` uses every GPU): ``ensembles`` hold the same members on different devices; the windows are cut
    into contiguous blocks (parallel.shard_bounds), one host thread per GPU runs its block through its own pinned
    buffers and streams, and every thread writes into its own slice of the shared result arrays.  No collective, no
    exchange.  -> (states, prob, power) as ``localize_host``."""
    import threading

    from . import parallel
    data = np.asarray(data)
    N, L = data.shape
    want_power = kw.get("p_on") is not None
    out = out or {}
    states = out["states"] if out.get("states") is not None else np.empty((N, L), dtype=states_dtype)
    prob = out["prob"] if out.get("prob") is not None else np.empty((N,), dtype=np.float64)
    power = (out["power"] if out.get("power") is not None else np.empty((N, L), dtype=power_dtype)) if want_power else None
    y = kw.pop("y", None)
    if y is not None:
        y = np.asarray(y).ravel()
        if y.size != N:
            raise ValueError("y must hold one label per window")
    world = len(ensembles)
    errors = [None] * world

    def run(r):
        try:
            s, e = parallel.shard_bounds(N, world, r)
            if e > s:
                o = {"states": states[s:e], "prob": prob[s:e], "power": power[s:e] if want_power else None}
                ensembles[r].localize_host(data[s:e], y=None if y is None else y[s:e], out=o, **kw)
        except BaseException as ex:  # noqa: BLE001 — re-raised in the caller's thread
            errors[r] = ex

    threads = [threading.Thread(target=run, args=(r,), name=f"camal-gpu{r}") for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for ex in errors:
        if ex is not None:
            raise ex
    return states, prob, power
