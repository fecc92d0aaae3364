"""GPU parity of camal_build_windows (window loop + NaN filter + aggregate scaling + weak labels on the device)
against oracle/windows_oracle.py and the reference's fixture: bit for bit."""
import numpy as np
import pytest

from tests.helpers import load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_build_windows_matches_reference_fixture(dtype):
    from camal_amd import synth
    from camal_amd.windows import NILMscaler, build_windows
    from oracle import windows_oracle
    c = load_case("windows_ref")
    for seed, W, S, kind in zip(c["seeds"], c["W"], c["S"], c["kinds"]):
        seed, W, S = int(seed), int(W), int(S)
        kind = int(kind) if str(kind).isdigit() else str(kind)
        stems = synth.house_series(6000, seed=seed)
        X, kept = windows_oracle.nilm_dataset(stems, W, S)
        sc = NILMscaler(power_scaling_type=kind, appliance_scaling_type="SameAsPower")
        sc.fit(X)
        series = stems.astype(dtype)
        res = build_windows(series, W, S, scaler=sc, want=("x", "x64", "label", "app_power", "app_state", "index"))
        assert res["count"] == len(kept)
        assert np.array_equal(res["index"].cpu().numpy(), kept)
        assert np.array_equal(res["label"].cpu().numpy(), c[f"y_{seed}"])
        got64 = res["x64"].cpu().numpy()
        if dtype == np.float64:  # the reference's own numbers
            assert np.array_equal(got64[::3], c[f"x_{seed}"])
        s1, div = sc.power_affine()
        Xd, _ = windows_oracle.nilm_dataset(series.astype(np.float64), W, S)
        ref64 = windows_oracle.scale_aggregate(Xd, s1, div)
        assert np.array_equal(got64, ref64)
        assert np.array_equal(res["x"].cpu().numpy(), ref64.astype(np.float32))
        assert np.array_equal(res["app_power"].cpu().numpy(), Xd[:, 1, 0, :].astype(np.float32))
        assert np.array_equal(res["app_state"].cpu().numpy(), Xd[:, 1, 1, :].astype(np.int8))


def test_build_windows_many_windows_and_no_scaler():
    """more windows than one scan block (1024), overlapping, two appliances, labels of the second."""
    from camal_amd import synth
    from camal_amd.windows import build_windows
    from oracle import windows_oracle
    stems = synth.house_series(70000, seed=9, n_app=2, nan_runs=40)
    W, S = 64, 16
    X, kept = windows_oracle.nilm_dataset(stems, W, S)
    assert len(kept) > 3000
    res = build_windows(stems, W, S, appliance=2, want=("x", "label", "index"))
    assert res["count"] == len(kept) and np.array_equal(res["index"].cpu().numpy(), kept)
    assert np.array_equal(res["x"].cpu().numpy(), X[:, 0, 0, :].astype(np.float32))
    assert np.array_equal(res["label"].cpu().numpy(), windows_oracle.weak_labels(X, 2))
    with pytest.raises(ValueError):
        build_windows(stems[:, :10], 64)
