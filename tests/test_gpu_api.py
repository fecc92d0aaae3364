"""GPU tests of the drop-in Python surface (CamAL / CamALResNet / CAM) against the fixtures
the reference itself produced — these read like calls into the reference's own classes."""
import pickle

import numpy as np
import pytest

from tests.helpers import REL_TOL_FP32, STATE_AGREE_MIN, case_state_dicts, load_case, rel_err

pytestmark = pytest.mark.gpu


def _fitted_camal(case, **kw):
    import torch
    from camal_amd import CamAL
    logs = {"ensemble_training_logs": {}, "camal_predictors_id": [], "is_fitted": True}
    for e, (k, sd) in enumerate(zip(case["members"], case_state_dicts(case))):
        name = f"Predictor_{e}"
        logs["ensemble_training_logs"][name] = {
            "kernel_size": int(k), "best_model_state_dict": {kk: torch.from_numpy(np.array(v)) for kk, v in sd.items()}}
        logs["camal_predictors_id"].append(name)
    camal = CamAL(n_predictors=len(case["members"]), device="cuda:0", **kw)
    camal.camal_logs = logs
    camal.is_fitted = True
    return camal


@pytest.mark.parametrize("name", ["ensemble_l256", "ensemble_l333_f64", "ensemble_l200_signed", "pair_l128_wma4"])
def test_camal_predict_matches_reference(name):
    c = load_case(name)
    camal = _fitted_camal(c)
    states, prob = camal.predict(c["data"], return_prob=True, w_ma=int(c["w_ma"]))
    assert states.shape == c["states"].shape and states.dtype == np.float64
    assert prob.shape == c["prob_detect"].shape and prob.dtype == np.float64
    assert np.abs(prob - c["prob_detect"]).max() < REL_TOL_FP32
    safe = np.abs(c["prob_detect"][:, 0] - 0.5) > 2e-3
    assert (states[safe] == c["states"][safe]).mean() >= STATE_AGREE_MIN
    only_states = camal(c["data"][0])  # 1-D input + __call__ (core.py:280-281,334-344)
    assert only_states.shape == (1, c["data"].shape[1])


def test_camal_prefilter_is_the_reference_device_path():
    c = load_case("ensemble_l160_gated")
    camal = _fitted_camal(c)
    camal.prefilter = True
    states, prob = camal.predict(c["data"], return_prob=True)
    assert np.abs(prob - c["prob_detect"]).max() < REL_TOL_FP32
    assert (prob == 0).any()  # gated windows keep prob 0 for every member (quirk 7)
    safe = np.abs(c["prob_detect"][:, 0] - 0.5) > 2e-3
    assert (states[safe] == c["states"][safe]).mean() >= STATE_AGREE_MIN
    # explicit y gate (core.py:297-300)
    y = np.array([1, 0] * (len(c["data"]) // 2), np.int8)
    camal.prefilter = False
    s2, p2 = camal.predict(c["data"], y=y, return_prob=True)
    assert (p2[y == 0] == 0).all() and (s2[y == 0] == 0).all()


def test_cam_run_matches_reference():
    import torch
    from camal_amd import CAM, CamALResNet
    c = load_case("ensemble_l64_degenerate")
    sd = case_state_dicts(c)[1]
    m = CamALResNet(kernel_size=int(c["members"][1]))
    m.to("cuda:0")
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.eval()
    cam = CAM(model=m, device="cuda:0", last_conv_layer=m._modules["layers"][2], fc_layer_name=m._modules["linear"])
    for i in range(len(c["data"])):
        got, label, proba = cam.run(instance=c["data"][i], returned_cam_for_label=1)
        assert got.dtype == np.float32 and got.shape == (c["data"].shape[1],)
        assert rel_err(got, c["cams"][1][i]) < REL_TOL_FP32
        assert np.abs(proba - c["probs"][1][i]).max() < REL_TOL_FP32
        if abs(c["probs"][1][i][1] - 0.5) > 1e-3:
            assert label == c["labels"][1][i]
    logits = m(torch.from_numpy(c["data"][:, None, :]).float())
    assert tuple(logits.shape) == (len(c["data"]), 2)
    assert rel_err(logits.cpu().numpy().T, c["logits"][1].T) < REL_TOL_FP32


def test_camal_load_and_test(tmp_path):
    """camal_logs pickle round trip (core.py:181-183,242-250) and CamAL.test (core.py:191-240)."""
    from oracle import camal_oracle as oracle
    c = load_case("ensemble_l256")
    camal = _fitted_camal(c)
    path = tmp_path / "camal_logs.pkl"
    with open(path, "wb") as f:
        pickle.dump(camal.camal_logs, f)
    from camal_amd import CamAL
    fresh = CamAL(n_predictors=5, device="cuda:0")
    fresh.load(str(path))
    assert fresh.is_fitted

    n, L = c["data"].shape

    class Scaler:  # NILMscaler with an integer power_scaling (torch_dataset.py:63-65,144)
        power_scaling_type = 1000
        power_stat1, power_stat2 = 0, 1000

        def inverse_transform(self, d):
            r = d.copy()
            r[:, 0, 0, :] = r[:, 0, 0, :] * 1000
            return r

    data4 = np.zeros((n, 2, 2, L))
    data4[:, 0, 0, :] = c["data"]
    data4[:, 1, 1, :] = (c["states"] > 0)           # "true" states
    data4[:, 1, 0, :] = (c["states"] > 0) * 1800.0   # true appliance power in W
    loc, clf = fresh.test(data4, appliance_mean_on_power=2000.0, scaler=Scaler())
    assert loc["F1_SCORE"] > 0.99 and "MAE" in loc and "TECA" in loc
    assert set(clf) >= {"ACCURACY", "ROC_AUC_SCORE", "AP"}
    ref_power = oracle.power_estimate(c["states"], c["data"], 2000.0, scale_mul=1000.0)
    y = data4[:, 1, 0, :].ravel()
    assert abs(loc["MAE"] - round(float(np.mean(np.abs(y - ref_power.ravel()))), 3)) < 0.5
    loc2, _ = fresh.test(data4)
    assert "MAE" not in loc2 and "F1_SCORE" in loc2


def test_device_metric_sums_match_the_reference_fixture():
    """camal_nilm_sums (one pass over device arrays) -> NILMmetrics.from_sums == the reference's NILMmetrics."""
    import torch
    from camal_amd import synth
    from camal_amd.metrics import NILMmetrics, nilm_sums_device
    c = load_case("metrics_ref")
    for i, seed in enumerate(c["seeds"]):
        y, y_hat, y_state, y_hat_state, _, _ = synth.metric_arrays(int(seed))
        for yy in (y, y.astype(np.float32)):
            sums, n = nilm_sums_device(y=yy, y_hat=torch.from_numpy(y_hat).cuda(), y_state=y_state,
                                       y_hat_state=torch.from_numpy(y_hat_state).cuda())
            assert n == len(y)
            ok = np.isfinite(y_hat)
            d = np.where(ok, y_hat, 0).astype(np.float64) - yy.astype(np.float64)
            assert abs(sums[0] - np.abs(d).sum()) <= 1e-12 * np.abs(d).sum()
            assert sums[8:].tolist() == [float(((y_state == a) & (np.abs(y_hat_state) == b)).sum())
                                         for a, b in ((1, 1), (0, 0), (0, 1), (1, 0))]
        m = NILMmetrics().from_sums(*nilm_sums_device(y=y, y_hat=torch.from_numpy(y_hat).cuda(), y_state=y_state,
                                                      y_hat_state=torch.from_numpy(y_hat_state).cuda()))
        assert [m[k] for k in c["loc_keys"]] == c["loc"][i].tolist()
    # states only (CamAL.test without a mean on-power), and a size that is not a multiple of anything
    y, y_hat, y_state, y_hat_state, _, _ = synth.metric_arrays(7, n=100003)
    sums, n = nilm_sums_device(y_state=y_state, y_hat_state=torch.from_numpy(y_hat_state).cuda())
    assert n == 100003 and sums[:8].tolist() == [0.0] * 8 and sums[8:].sum() == n
    m = NILMmetrics().from_sums(sums, n, regression=False)
    assert m == NILMmetrics()(y_state=y_state, y_hat_state=y_hat_state)


def test_attention_module_and_subsequence_class():
    from oracle import camal_oracle as oracle
    c = load_case("ensemble_l256")
    camal = _fitted_camal(c)
    camal.eval()
    got = camal.attention_sigmoid_module(c["data"], c["avg_cam"], w_ma=5)
    ref = oracle.attention_sigmoid_module(c["data"], c["avg_cam"], w_ma=5)
    assert (got == ref).mean() >= STATE_AGREE_MIN
    y = camal._predict_subsequence_class(camal.list_resnet_predictors[0], c["data"])
    firm = np.abs(c["probs"][0][:, 1] - 0.5) > 1e-3
    assert y.dtype == np.int8 and (y[firm] == c["labels"][0][firm]).all()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_streaming_predict_equals_one_shot(dtype):
    """CamAL.predict's host pipeline (chunks through pinned buffers, copies on their own streams under the kernels):
    N = 3 x chunk + 7 windows give bit for bit what one resident call gives, states, probabilities and power, with
    a per-window gate, for float32 and float64 callers; a second call reuses the buffers."""
    import torch
    from camal_amd import synth, weights
    from camal_amd.engine import Ensemble
    chunk, L = 40, 200
    N = 3 * chunk + 7
    x = synth.ukdale_windows(N, L, seed=31).astype(dtype)
    if dtype == np.float64:
        x = x - 0.7  # signed input: -1 / -0 states on the float64 path
    sds = [weights.make_state_dict(k, seed=12, head_gain=3.0) for k in (5, 9, 25)]
    y = (np.arange(N) % 3 != 0).astype(np.int8)
    ens = Ensemble(sds, device="cuda:0", precision="fp32")
    is64 = dtype == np.float64
    ref = ens.localize(x.astype(np.float32), x64=x if is64 else None, y=y, pred_f64=is64, p_on=2000.0, scale_mul=1000.0)
    torch.cuda.synchronize()
    for _ in range(2):
        states, prob, power = ens.localize_host(x, y=y, p_on=2000.0, scale_mul=1000.0, chunk_windows=chunk)
        assert states.dtype == np.float64 and states.shape == (N, L)
        assert np.array_equal(prob, ref["prob"].cpu().numpy())
        rs = ref["state"].cpu().numpy()
        assert np.array_equal(states, rs.astype(np.float64)) and np.array_equal(np.signbit(states), np.signbit(rs))
        assert np.array_equal(power, ref["power"].cpu().numpy().astype(np.float64))
    s1, _p, _ = ens.localize_host(x[:5], chunk_windows=chunk)  # fewer windows than a chunk, no power, no gate
    r1 = ens.localize(x[:5].astype(np.float32), x64=x[:5] if is64 else None, pred_f64=is64)
    assert np.array_equal(s1, r1["state"].cpu().numpy().astype(np.float64))
    ens.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cam_hook_on_the_second_block(precision):
    """CAM(last_conv_layer = model.layers[1]) (class_activation_map.py:25-31): tests/golden/camtap_l120.npz holds the
    reference's rows for label 1 and for the predicted label; the first block is refused (its 64 channels do not match
    the head's 128 weights: the reference's dot product fails there too)."""
    from camal_amd import CAM, CamALResNet, weights
    from camal_amd.engine import Ensemble
    c = load_case("camtap_l120")
    sd = weights.make_state_dict(int(c["k"]), seed=int(c["seed"]), mid_channels=int(c["mid"]), head_gain=float(c["head_gain"]),
                                 head_bias=tuple(c["head_bias"]))
    assert 0 < c["labels"].sum() < len(c["labels"])  # both classes are predicted: the label-0 rows are exercised
    if precision == "fp32":
        m = CamALResNet(kernel_size=int(c["k"]))
        m.to("cuda:0")
        m.load_state_dict(sd)
        m.eval()
        cam = CAM(m, "cuda:0", last_conv_layer=m._modules["layers"][1], fc_layer_name=m._modules["linear"])
        rows1, labels, probs = cam.run_batch(c["data"], 1)
        assert rel_err(rows1, c["cams_label1"]) < REL_TOL_FP32
        assert np.abs(probs - c["probs"]).max() < REL_TOL_FP32 and np.array_equal(labels, c["labels"])
        rowsp, _, _ = cam.run_batch(c["data"], None)
        assert rel_err(rowsp, c["cams_predicted"]) < REL_TOL_FP32
        one, lab, _ = cam.run(instance=c["data"][2], returned_cam_for_label=1)
        assert rel_err(one[None], c["cams_label1"][2:3]) < REL_TOL_FP32 and lab == int(c["labels"][2])
        with pytest.raises(ValueError):
            CAM(m, "cuda:0", last_conv_layer=m._modules["layers"][0], fc_layer_name=m._modules["linear"])
        # the last block through the same entry point is what it always was
        last = CAM(m, "cuda:0", last_conv_layer=m._modules["layers"][2], fc_layer_name=m._modules["linear"])
        assert last.tap_block == 2
    else:
        ens = Ensemble([sd], device="cuda:0", precision="bf16")  # fused blocks: the tap sits between block 2 and block 3
        _, cam = ens.forward_cam(c["data"], cam_label=1, tap_block=1)
        assert rel_err(cam[0].cpu().numpy(), c["cams_label1"]) < 3e-2
        ens.close()


def test_streaming_predict_compact_dtypes_and_caller_arrays():
    """The host pipeline at configs[2]'s scale keeps the host side bounded: int8 states + float32 power (5 instead of 16
    bytes per timestamp), written into arrays the caller brings (``out``; np.memmap works the same way).  Same
    decisions as the one-shot resident call; -0.0 survives in float dtypes and becomes 0 in integer ones."""
    import torch
    from camal_amd import synth, weights
    from camal_amd.engine import Ensemble
    chunk, L = 48, 160
    N = 5 * chunk + 11  # more chunks than buffer sets: every set is reused while its widening jobs are checked
    x = (synth.ukdale_windows(N, L, seed=33) * 2.0 - 2.5).astype(np.float32)  # signed: -1 and -0 states occur
    sds = [weights.make_state_dict(k, seed=13, head_gain=3.0) for k in (7, 15)]
    ens = Ensemble(sds, device="cuda:0", precision="fp32")
    ref = ens.localize(x, p_on=2000.0, scale_mul=1000.0)
    torch.cuda.synchronize()
    rs, rp, rw = ref["state"].cpu().numpy(), ref["prob"].cpu().numpy(), ref["power"].cpu().numpy()
    assert (rs == -1).any() and (np.signbit(rs) & (rs == 0)).any()
    out = {"states": np.full((N, L), 99, np.int8), "prob": np.full((N, 1), -1.0), "power": np.full((N, L), -1.0, np.float32)}
    s8, p, w = ens.localize_host(x, p_on=2000.0, scale_mul=1000.0, chunk_windows=chunk, out=out)
    assert s8 is out["states"] and p is out["prob"] and w is out["power"]
    assert np.array_equal(s8, rs.astype(np.int8)) and np.array_equal(p[:, 0], rp) and np.array_equal(w, rw)
    s32, _, w64 = ens.localize_host(x, p_on=2000.0, scale_mul=1000.0, chunk_windows=chunk, states_dtype=np.float32)
    assert s32.dtype == np.float32 and np.array_equal(s32, rs) and np.array_equal(np.signbit(s32), np.signbit(rs))
    assert w64.dtype == np.float64 and np.array_equal(w64, rw.astype(np.float64))
    with pytest.raises(ValueError):
        ens.localize_host(x, out={"states": np.empty((N, L + 1))})
    # the tiled post-processing kernel's byte stores: a length that is no multiple of 4, a 4-step moving average, float64 data
    x2 = x[:50, :131].astype(np.float64)
    r2 = ens.localize(x2.astype(np.float32), x64=x2, pred_f64=True, w_ma=4, p_on=2000.0, scale_mul=1000.0)
    s2, p2, w2 = ens.localize_host(x2, w_ma=4, p_on=2000.0, scale_mul=1000.0, chunk_windows=16)
    rs2 = r2["state"].cpu().numpy()
    assert np.array_equal(s2, rs2.astype(np.float64)) and np.array_equal(np.signbit(s2), np.signbit(rs2))
    assert np.array_equal(p2, r2["prob"].cpu().numpy()) and np.array_equal(w2, r2["power"].cpu().numpy().astype(np.float64))
    # one window (fewer rows than worker threads) and none at all
    s1, p1, _ = ens.localize_host(x2[:1], w_ma=4)
    assert np.array_equal(s1, rs2[:1].astype(np.float64)) and p1.shape == (1,)
    s0, p0, w0 = ens.localize_host(x2[:0], p_on=2000.0)
    assert s0.shape == (0, 131) and p0.shape == (0,) and w0.shape == (0, 131)
    ens.close()


def test_predict_over_several_gpus_from_one_process():
    """CamAL.devices: one Ensemble + host thread per GPU over contiguous window blocks, results in disjoint slices of
    the output arrays, no collective (the reference's single-process all_gpu idea, torch_trainer.py:82-90).
    This box has one GPU: two contexts on cuda:0 stand in for two GPUs; the result must equal the single-device
    run bit for bit, ragged split and y gate included."""
    from camal_amd import synth
    c = load_case("ensemble_l256")
    x = synth.ukdale_windows(37, c["data"].shape[1], seed=41).astype(c["data"].dtype)
    y = (np.arange(37) % 4 != 1).astype(np.int8)
    one = _fitted_camal(c)
    s1, p1 = one.predict(x, return_prob=True)
    s1y, p1y = one.predict(x, y=y, return_prob=True)
    two = _fitted_camal(c)
    two.devices = ["cuda:0", "cuda:0"]
    two.stream_windows = 8
    s2, p2 = two.predict(x, return_prob=True)
    assert len(two._ensembles) == 2 and two._ensembles[0] is not two._ensembles[1]
    assert s2.dtype == np.float64 and np.array_equal(s1, s2) and np.array_equal(p1, p2)
    s2y, p2y = two.predict(x, y=y, return_prob=True)
    assert np.array_equal(s1y, s2y) and np.array_equal(p1y, p2y)
    assert np.array_equal(two.predict(x[:1]), s1[:1])  # fewer windows than devices: the first device alone
    three = _fitted_camal(c)
    three.all_devices = True  # every visible GPU (one here)
    three.states_dtype = np.int8
    s3 = three.predict(x)
    assert s3.dtype == np.int8 and np.array_equal(s3, s1.astype(np.int8))


def test_camal_test_streams_at_any_size():
    """CamAL.test (core.py:191-240) with the windows, true powers and true states streamed through the device in chunks
    (Ensemble.test_sums_host): twelve sums per chunk, nothing of size [N, L] ever comes back.  Against the host route —
    predict + the reference's NILMmetrics formulas on host arrays, inverse scaling through NILMscaler.inverse_transform —
    for a scaler that scales the appliance channel too, with chunks smaller than the test set and ragged."""
    from camal_amd import synth
    from camal_amd.metrics import NILMmetrics
    from camal_amd.windows import NILMscaler
    c = load_case("ensemble_l256")
    n, L = 45, c["data"].shape[1]
    x = synth.ukdale_windows(n, L, seed=51).astype(np.float64)  # kW
    camal = _fitted_camal(c)
    states, prob = camal.predict(x, return_prob=True)
    rng = np.random.default_rng(3)
    true_state = ((states > 0) ^ (rng.random(states.shape) < 0.1)).astype(np.float64)
    true_state[::4] = 0.0  # windows without the appliance: both weak labels occur (ROC AUC needs them)
    data4 = np.zeros((n, 2, 2, L))
    data4[:, 0, 0, :] = x * 1000.0                                  # W
    data4[:, 1, 0, :] = true_state * rng.uniform(1500, 2500, size=states.shape)
    data4[:, 1, 1, :] = true_state
    sc = NILMscaler(power_scaling_type=1000, appliance_scaling_type="SameAsPower")
    scaled = sc.fit_transform(data4.copy())
    assert np.allclose(scaled[:, 0, 0, :], x)
    # host route: the reference's flow on host arrays
    y = sc.inverse_transform(scaled)[:, 1, 0, :].ravel()
    y_hat = np.minimum(states.astype(int) * 2000.0, scaled[:, 0, 0, :] * 1000.0).ravel()
    want = NILMmetrics()(y, y_hat, scaled[:, 1, 1, :].ravel().astype(int), states.ravel().astype(int))
    for chunk in (None, 7, 16):
        camal.stream_windows = chunk
        loc, clf = camal.test(scaled, appliance_mean_on_power=2000.0, scaler=sc)
        assert loc == want, (chunk, loc, want)
        assert clf["ACCURACY"] >= 0.0 and "AP" in clf
    camal.stream_windows = 7
    loc2, _ = camal.test(scaled)
    assert loc2 == NILMmetrics()(y_state=scaled[:, 1, 1, :].ravel().astype(int), y_hat_state=states.ravel().astype(int))


def test_localize_sequence_is_capturable_in_a_hip_graph():
    """After camal_reserve the whole camal_localize launch sequence (conv kernels, head, header, post-processing, the one
    memset) allocates nothing and synchronises nothing: it can be captured in a HIP graph and replayed on new inputs in
    the same buffers, bit-equal to the eager call.  (Measured on 4096 x 1440 x E = 5: 29.4 ms replayed, 29.3 ms eager:
    the ~90 launches of a step are not what limits it.)"""
    import torch
    from camal_amd import synth, weights
    from camal_amd.engine import Ensemble
    sds = [weights.make_state_dict(k, seed=14, head_gain=3.0) for k in (5, 9)]
    B, L = 40, 500
    xa, xb = (torch.from_numpy(synth.ukdale_windows(B, L, seed=s)).cuda() for s in (61, 62))
    ens = Ensemble(sds, device="cuda:0", precision="bf16")
    ens.reserve(B, L)
    x = xa.clone()
    out = {"prob": torch.empty(B, dtype=torch.float64, device="cuda"), "state": torch.empty(B, L, dtype=torch.float32, device="cuda"),
           "power": torch.empty(B, L, dtype=torch.float32, device="cuda")}

    def step():
        ens.localize(x, p_on=2000.0, scale_mul=1000.0, out=out)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()  # warm-up outside the capture: kernel attributes are set on first use
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for src in (xb, xa):
        x.copy_(src)
        g.replay()
        torch.cuda.synchronize()
        got = {k: v.clone() for k, v in out.items()}
        step()
        torch.cuda.synchronize()
        assert all(torch.equal(got[k], out[k]) for k in got)
    ens.close()


@pytest.mark.parametrize("mid", [32, 128])
def test_camal_train_predict_test_on_other_widths(mid):
    """CamAL(mid_channels=...) end to end with the reference's default kernel list (core.py:47-56): train, the eval() that
    follows it, predict, test.  The reference's eval() rebuilds default-width models (core.py:386-390) and fails in
    load_state_dict for any other width; here the width is read from the checkpoint.  mid_channels = 128 with k = 25 also
    needs the four-wave data-gradient tile (256 -> 128 channels)."""
    from camal_amd import CamAL, synth
    L = 96
    x, y = synth.ideal_windows(64, L, seed=3)
    xv, yv = synth.ideal_windows(12, L, seed=4)
    xt, yt = synth.ideal_windows(12, L, seed=5)
    m = CamAL(n_predictors=3, device="cuda:0", mid_channels=mid)
    m.train((x, y), (xv, yv), (xt, yt), list_kernel_sizes=[5, 7, 9, 15, 25], n_try_by_kernel=1, epochs=2, batch_size=32)
    assert all(p.mid_channels == mid for p in m.list_resnet_predictors)
    states, prob = m.predict(xt, return_prob=True)
    assert states.shape == xt.shape and prob.shape == (len(xt), 1) and np.isfinite(prob).all()
    m.precision = "bf16"
    m.eval()
    _, prob16 = m.predict(xt, return_prob=True)
    assert np.abs(prob16 - prob).max() < 5e-2
    data4 = np.zeros((len(xt), 2, 2, L))
    data4[:, 0, 0, :] = xt
    data4[:, 1, 0, :] = xt * 0.5
    data4[:, 1, 1, :] = xt > 1.0
    loc, clf = m.test(data4)
    assert "F1_SCORE" in loc and "ACCURACY" in clf
