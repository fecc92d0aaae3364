"""camal_build_windows_static — the entry point that allocates nothing (static device scratch) — against
camal_build_windows with a caller workspace, through the C ABI: every output bit for bit, on two calls in a row,
the second on a smaller input (the scratch keeps the first call's contents), and the size limit's error."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(lib, _lib, torch, series, W, S, app, s1, div, static):
    dev = series.device
    n_rows, T = series.shape
    n = int(lib.camal_windows_count(T, W, S))
    # outputs pre-filled with a sentinel: rows past `count` must stay untouched by both entry points
    out = {"x": torch.full((n, W), -7.0, dtype=torch.float32, device=dev),
           "x64": torch.full((n, W), -7.0, dtype=torch.float64, device=dev),
           "label": torch.full((n,), 5, dtype=torch.int8, device=dev),
           "app_power": torch.full((n, W), -7.0, dtype=torch.float32, device=dev),
           "app_state": torch.full((n, W), 5, dtype=torch.int8, device=dev),
           "index": torch.full((n,), -3, dtype=torch.int32, device=dev)}
    count = torch.full((1,), -1, dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    common = (_lib.ptr(series), int(series.dtype == torch.float64), n_rows, T, W, S, app, float(s1), float(div),
              _lib.ptr(out["x"]), _lib.ptr(out["x64"]), _lib.ptr(out["label"]), _lib.ptr(out["app_power"]),
              _lib.ptr(out["app_state"]), _lib.ptr(out["index"]), _lib.ptr(count))
    if static:
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(dev)[0]
        _lib.check(lib.camal_build_windows_static(*common, stream))
        torch.cuda.synchronize()
        if static == "no_alloc":  # not the first call on this device (that one resolves the symbol's address)
            assert torch.cuda.mem_get_info(dev)[0] == free0, "camal_build_windows_static changed the device's free memory"
    else:
        ws = torch.empty(int(lib.camal_windows_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        _lib.check(lib.camal_build_windows(*common, _lib.ptr(ws), ws.numel(), stream))
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items()}
    res["count"] = int(count.item())
    return res


def test_static_window_builder_equals_workspace_variant():
    import torch

    from camal_amd import _lib, synth
    lib = _lib.load()
    assert lib.camal_windows_static_scratch_bytes() == (5 << 20) + 4096
    big = torch.from_numpy(synth.house_series(70000, seed=9, n_app=2, nan_runs=40)).cuda()          # 4372 windows, 5 scan blocks
    small = torch.from_numpy(synth.house_series(6000, seed=4, n_app=2, nan_runs=6).astype(np.float32)).cuda()
    # two calls in a row on the static scratch, the second on a smaller input
    got_big = _run(lib, _lib, torch, big, 64, 16, 2, 0.25, 1.5, static=True)
    got_small = _run(lib, _lib, torch, small, 100, 50, 1, 0.0, 1000.0, static="no_alloc")
    ref_big = _run(lib, _lib, torch, big, 64, 16, 2, 0.25, 1.5, static=False)
    ref_small = _run(lib, _lib, torch, small, 100, 50, 1, 0.0, 1000.0, static=False)
    for got, ref in ((got_big, ref_big), (got_small, ref_small)):
        assert 0 < got["count"] == ref["count"] < len(ref["index"])  # some windows dropped for NaN
        for k in ("x", "x64", "label", "app_power", "app_state", "index"):
            assert np.array_equal(got[k], ref[k]), k
    # and once more on the big input after the small one
    again = _run(lib, _lib, torch, big, 64, 16, 2, 0.25, 1.5, static="no_alloc")
    for k in ("x", "x64", "label", "app_power", "app_state", "index"):
        assert np.array_equal(again[k], ref_big[k]), k


def test_static_window_builder_rejects_oversize():
    import torch

    from camal_amd import _lib
    lib = _lib.load()
    T = (1 << 20) + 9  # W = stride = 1: one window per sample, more than CAMAL_WINDOWS_STATIC_MAX
    series = torch.zeros((1, T), dtype=torch.float32, device="cuda")
    x = torch.empty((T, 1), dtype=torch.float32, device="cuda")
    count = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = lib.camal_build_windows_static(_lib.ptr(series), 0, 1, T, 1, 1, 1, 0.0, 1.0, _lib.ptr(x), None, None, None, None,
                                        None, _lib.ptr(count), None)
    assert rc == -1 and b"exceed the static scratch" in lib.camal_last_error()
    # exactly the maximum is accepted
    rc = lib.camal_build_windows_static(_lib.ptr(series), 0, 1, 1 << 20, 1, 1, 1, 0.0, 1.0, _lib.ptr(x), None, None, None,
                                        None, None, _lib.ptr(count), None)
    assert rc == 0, lib.camal_last_error()
    torch.cuda.synchronize()
    assert int(count.item()) == 1 << 20
    # host memory is not device memory of one GPU: refused, nothing launched
    host = torch.zeros((1, 64), dtype=torch.float32).pin_memory()
    rc = lib.camal_build_windows_static(_lib.ptr(host), 0, 1, 64, 8, 8, 1, 0.0, 1.0, _lib.ptr(x), None, None, None, None,
                                        None, _lib.ptr(count), None)
    assert rc != 0
