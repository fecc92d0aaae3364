"""CPU: the oracle (oracle/camal_oracle.py) against outputs of the reference itself.

The .npz files hold what the reference's CamALResNet / CAM.run / CamAL.predict returned
in the dev container (tests/golden/make_golden.py).  This pins the oracle; the GPU
parity tests then compare the HIP path with the oracle and with the same fixtures.
"""
import numpy as np
import pytest

from oracle import camal_oracle as oracle
from tests.helpers import REL_TOL_FP32, case_state_dicts, golden_cases, load_case, rel_err, resnetvar_state_dicts

CASES = golden_cases("ensemble_") + golden_cases("pair_")


def test_fixtures_present():
    assert len(CASES) >= 6


@pytest.mark.parametrize("name", CASES)
def test_members_match_reference(name):
    c = load_case(name)
    sds = case_state_dicts(c)
    for e, sd in enumerate(sds):
        cams, labels, probs = oracle.cam_batch(sd, c["data"], 1)
        assert rel_err(cams, c["cams"][e]) < REL_TOL_FP32
        assert np.abs(probs - c["probs"][e]).max() < REL_TOL_FP32
        # labels may only differ where the probabilities tie to within rounding
        diff = labels != c["labels"][e]
        assert np.all(np.abs(c["probs"][e][diff, 1] - 0.5) < 1e-4)
        logits = oracle.resnet_forward(sd, c["data"][:, None, :])
        assert rel_err(logits.T, c["logits"][e].T) < REL_TOL_FP32


@pytest.mark.parametrize("name", CASES)
def test_predict_matches_reference(name):
    c = load_case(name)
    sds = case_state_dicts(c)
    states, prob, avg_cam, _ = oracle.camal_predict(
        sds, c["data"], w_ma=int(c["w_ma"]), gated=bool(c["gated"]), return_intermediate=True)
    assert states.dtype == c["states"].dtype and states.shape == c["states"].shape
    assert np.abs(prob - c["prob_detect"]).max() < REL_TOL_FP32
    assert np.abs(avg_cam - c["avg_cam"]).max() < REL_TOL_FP32
    assert (states == c["states"]).mean() >= 0.999
    # the signed case must contain the reference's negative states
    if "signed" in name:
        assert (c["states"] < 0).any()


@pytest.mark.parametrize("name", golden_cases("resnetvar_"))
def test_resnet_variant_members_match_reference(name):
    """classifiers/resnet.py's ResNet (kernels 7/5/3, zero "same" padding): NumPy and torch restatements against
    the reference class's own logits and CAM.run rows."""
    import torch
    from oracle import torch_path
    c = load_case(name)
    for e, sd in enumerate(resnetvar_state_dicts(c)):
        assert oracle.is_resnet_variant(sd)
        cams, labels, probs = oracle.cam_batch(sd, c["data"], 1)
        assert rel_err(cams, c["cams"][e]) < REL_TOL_FP32
        assert np.abs(probs - c["probs"][e]).max() < REL_TOL_FP32
        diff = labels != c["labels"][e]  # only where the probabilities tie to within rounding
        assert np.all(np.abs(c["probs"][e][diff, 1] - 0.5) < 1e-4)
        logits = oracle.resnet_forward(sd, c["data"][:, None, :])
        assert rel_err(logits.T, c["logits"][e].T) < REL_TOL_FP32
        cam0, lab0, p0 = torch_path.cam_run({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}, c["data"][0])
        assert rel_err(cam0[None], c["cams"][e][:1]) < REL_TOL_FP32 and np.abs(p0 - c["probs"][e][0]).max() < REL_TOL_FP32


@pytest.mark.parametrize("name", golden_cases("width"))
def test_other_widths_match_reference(name):
    """CamALResNet(mid_channels=32 / 128) (camal_resnet.py:5-19): the oracle against the reference class's own logits
    and CAM.run rows."""
    from tests.helpers import width_state_dicts
    c = load_case(name)
    for e, sd in enumerate(width_state_dicts(c)):
        assert sd["linear.weight"].shape == (2, 2 * int(c["mid"]))
        cams, labels, probs = oracle.cam_batch(sd, c["data"], 1)
        assert rel_err(cams, c["cams"][e]) < REL_TOL_FP32
        assert np.abs(probs - c["probs"][e]).max() < REL_TOL_FP32
        diff = labels != c["labels"][e]
        assert np.all(np.abs(c["probs"][e][diff, 1] - 0.5) < 1e-4)
        logits = oracle.resnet_forward(sd, c["data"][:, None, :])
        assert rel_err(logits.T, c["logits"][e].T) < REL_TOL_FP32


def test_resnet_variant_fixtures_present():
    assert len(golden_cases("resnetvar_")) >= 2


def test_loop_variant_equals_batched():
    c = load_case("pair_l128_wma4")
    sds = case_state_dicts(c)
    s1, p1 = oracle.camal_predict(sds, c["data"], w_ma=4, return_prob=True)
    s2, p2 = oracle.camal_predict_loop(sds, c["data"], w_ma=4)
    assert np.array_equal(s1, s2) and np.allclose(p1, p2, atol=1e-6)


def test_moving_average_taps():
    # w=5 -> m in [t-2,t+2]; w=4 -> m in [t-2,t+1]; zeros outside [0,L)
    x = np.arange(1.0, 11.0)
    for w, lo, hi in ((5, -2, 2), (4, -2, 1), (3, -1, 1), (1, 0, 0)):
        ref = np.array([sum(x[m] for m in range(t + lo, t + hi + 1) if 0 <= m < len(x)) / w
                        for t in range(len(x))])
        assert np.allclose(oracle.moving_average(x, w), ref)


def test_state_threshold_is_ln3():
    # 2*sigmoid(z)-1 rounds (half to even) to 1 iff z > ln 3
    z = np.array([np.log(3.0) - 1e-9, np.log(3.0) + 1e-9, -np.log(3.0) - 1e-9, 0.0])
    assert list(np.round(oracle.rectified_sigmoid(z))) == [0.0, 1.0, -1.0, 0.0]


def test_power_clip():
    states = np.array([[0.0, 1.0, 1.0, 1.0]])
    x = np.array([[0.5, 0.1, 3.0, 2.0]])  # kW, scaler multiplies by 1000
    p = oracle.power_estimate(states, x, 2000.0, scale_mul=1000.0)
    assert np.allclose(p, [[0.0, 100.0, 2000.0, 2000.0]])


def test_fold_bn_equivalence():
    from camal_amd import weights
    sd = weights.make_state_dict(7, seed=3)
    x = np.random.default_rng(0).normal(size=(2, 64, 50)).astype(np.float32)
    p = "layers.0.layers.1"
    z = oracle.conv1d_replicate(x, sd[p + ".layers.0.weight"], sd[p + ".layers.0.bias"])
    y = oracle.batchnorm_eval(z, sd[p + ".layers.1.weight"], sd[p + ".layers.1.bias"],
                              sd[p + ".layers.1.running_mean"], sd[p + ".layers.1.running_var"])
    w2, b2 = oracle.fold_bn(sd[p + ".layers.0.weight"], sd[p + ".layers.0.bias"], sd[p + ".layers.1.weight"],
                            sd[p + ".layers.1.bias"], sd[p + ".layers.1.running_mean"], sd[p + ".layers.1.running_var"])
    y2 = oracle.conv1d_replicate(x, w2, b2)
    assert np.abs(y - y2).max() < 1e-5 * np.abs(y).max() + 1e-5


@pytest.mark.parametrize("name", ["pair_l128_wma4", "ensemble_l64_degenerate"])
def test_torch_path_matches_reference(name):
    """oracle/torch_path.py (the CPU-baseline restatement) against the reference's outputs."""
    from oracle import torch_path
    c = load_case(name)
    states, prob = torch_path.predict_loop(case_state_dicts(c), c["data"], w_ma=int(c["w_ma"]))
    assert np.abs(prob - c["prob_detect"]).max() < 1e-6
    assert (states == c["states"]).mean() >= 0.999


def test_window_oracle_and_scaler_mirror_match_the_reference_fixture():
    """tests/golden/windows_ref.npz: the reference's NILMscaler.fit_transform / nilmdataset_to_clfdataset outputs
    on the 4-D arrays oracle/windows_oracle.py builds (make_golden.py::windows_case)."""
    from camal_amd import synth
    from camal_amd.windows import NILMscaler, n_windows
    from oracle import windows_oracle
    from tests.helpers import load_case
    c = load_case("windows_ref")
    for seed, W, S, kind in zip(c["seeds"], c["W"], c["S"], c["kinds"]):
        seed, W, S = int(seed), int(W), int(S)
        kind = int(kind) if str(kind).isdigit() else str(kind)
        stems = synth.house_series(6000, seed=seed)
        X, kept = windows_oracle.nilm_dataset(stems, W, S)
        assert np.array_equal(kept, c[f"kept_{seed}"]) and len(kept) < n_windows(6000, W, S)  # NaN windows dropped
        sc = NILMscaler(power_scaling_type=kind, appliance_scaling_type="SameAsPower")
        Xs = sc.fit_transform(X.copy())
        assert [float(sc.power_stat1), float(sc.power_stat2)] == c[f"stat_{seed}"].tolist()
        assert np.array_equal(Xs[::3, 0, 0, :], c[f"x_{seed}"])            # bit for bit (float64)
        assert np.array_equal(Xs[::7, 1, 0, ::5], c[f"app_{seed}"])
        s1, div = sc.power_affine()
        assert np.array_equal(windows_oracle.scale_aggregate(X, s1, div)[::3], c[f"x_{seed}"])
        assert np.array_equal(windows_oracle.weak_labels(X), c[f"y_{seed}"])
        assert float(np.abs(sc.inverse_transform(Xs)[:, 0, 0, :] - X[:, 0, 0, :]).max()) == float(c[f"inv_err_{seed}"])
    with pytest.raises(AssertionError):  # the reference's default constructor fails its own assert (:37-38)
        NILMscaler()


def test_cfg0_full_size_matches_reference():
    """BASELINE configs[0] at its own size (E=5, 64 x 1024; tests/golden/cfg0_l1024.npz = the reference's CAM.run
    and CamAL.predict): the torch restatement on all 64 windows (it is also the CPU baseline bench.py times), the
    NumPy oracle on every 6th window."""
    from camal_amd import synth
    from oracle import torch_path
    c = load_case("cfg0_l1024")
    sds = case_state_dicts(c)
    x = synth.ukdale_windows(int(c["n"]), int(c["length"]), seed=int(c["data_seed"]))
    states, prob = torch_path.predict_loop(sds, x)
    assert np.abs(prob - c["prob_detect"]).max() < 1e-6
    assert (states == c["states"]).mean() >= 0.9999 and set(np.unique(states)) <= {0.0, 1.0}
    sb, pb = torch_path.predict_batched(sds, x)  # the batched variant of the baseline: same states
    assert np.abs(pb - c["prob_detect"]).max() < 1e-5 and (sb == c["states"]).mean() >= 0.999
    idx = np.arange(0, len(x), 6)
    s2, p2, avg, outs = oracle.camal_predict(sds, x[idx], return_intermediate=True)
    st = int(c["cam_stride"])
    assert np.abs(p2 - c["prob_detect"][idx]).max() < REL_TOL_FP32
    safe = np.abs(c["prob_detect"][idx, 0] - 0.5) > 2e-3
    assert (s2[safe] == c["states"][idx][safe]).mean() >= 0.999
    assert np.abs(avg[safe][:, ::st] - c["avg_cam_strided"][idx][safe]).max() < REL_TOL_FP32
    for e in range(len(sds)):
        cams, labels, probs = outs[e]
        assert np.abs(cams[:, ::st] - c["cams_strided"][e][idx]).max() < REL_TOL_FP32 * np.abs(c["cam_row_max"][e][idx]).max()
        assert np.abs(probs - c["probs"][e][idx]).max() < REL_TOL_FP32


def test_week_windows_match_reference():
    """BASELINE configs[4]'s length (L = 10080, a week at 1 min) on 6 REFIT-shaped windows and two members (k = 5, 25;
    tests/golden/week_l10080.npz = the reference's CAM.run and CamAL.predict): the torch restatement and the NumPy
    oracle on the first two windows (the oracle's plain loops take their time at this length)."""
    from camal_amd import synth
    from oracle import torch_path
    c = load_case("week_l10080")
    sds = case_state_dicts(c)
    x = synth.refit_windows(int(c["n"]), int(c["length"]), seed=int(c["data_seed"]))
    states, prob = torch_path.predict_loop(sds, x)
    assert np.abs(prob - c["prob_detect"]).max() < 1e-6
    assert (states == c["states"]).mean() >= 0.9999 and (c["states"] != 0).sum() > 100
    idx = np.arange(2)
    s2, p2, avg, outs = oracle.camal_predict(sds, x[idx], return_intermediate=True)
    st = int(c["cam_stride"])
    assert np.abs(p2 - c["prob_detect"][idx]).max() < REL_TOL_FP32
    assert (s2 == c["states"][idx]).mean() >= 0.999
    assert np.abs(avg[:, ::st] - c["avg_cam_strided"][idx]).max() < REL_TOL_FP32
    for e in range(len(sds)):
        cams, labels, probs = outs[e]
        assert np.abs(cams[:, ::st] - c["cams_strided"][e][idx]).max() < REL_TOL_FP32 * np.abs(c["cam_row_max"][e][idx]).max()
        assert np.abs(probs - c["probs"][e][idx]).max() < REL_TOL_FP32

