"""GPU tests at BASELINE.json's full sizes.  The oracle cannot run 4096 x 1024 windows in seconds, so
these use what the domain offers: windows are independent (a sample of them against the oracle,
permutation equivariance bit for bit, invariance to the internal chunking) and the week-long window of
configs[4] at a small batch against the oracle."""
import numpy as np
import pytest

from tests.helpers import REL_TOL_FP32, STATE_AGREE_MIN, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def test_config1_sample_vs_oracle_and_permutation(torch_cuda):
    """configs[1]: single ResNet k=7, 4096 x 1024, fp32."""
    torch = torch_cuda
    from camal_amd import synth
    from camal_amd.calibrate import calibrated_state_dicts
    from camal_amd.engine import Ensemble
    from oracle import camal_oracle as oracle
    B, L = 4096, 1024
    x = synth.ukdale_windows(B, L, seed=77)
    sds = calibrated_state_dicts([7], seed=3, sample_x=x[:64], device="cuda:0", precision="fp32")
    ens = Ensemble(sds, device="cuda:0", precision="fp32")
    xd = torch.from_numpy(x).cuda()
    res = ens.localize(xd, p_on=2000.0, scale_mul=1000.0)
    prob, state, power = res["prob"].clone(), res["state"].clone(), res["power"].clone()
    # a sample of windows, spread over the internal chunks and post-processing groups
    idx = np.array([0, 255, 256, 1777, 4000, 4095])
    states_ref, prob_ref = oracle.camal_predict(sds, x[idx], return_prob=True)
    assert np.abs(prob[idx].cpu().numpy() - prob_ref[:, 0]).max() < REL_TOL_FP32
    safe = np.abs(prob_ref[:, 0] - 0.5) > 2e-3
    assert safe.any()
    assert (state[idx].cpu().numpy()[safe] == states_ref[safe]).mean() >= STATE_AGREE_MIN
    ref_power = oracle.power_estimate(state[idx].cpu().numpy(), x[idx], 2000.0, scale_mul=1000.0)
    assert np.allclose(power[idx].cpu().numpy(), ref_power, rtol=1e-6, atol=1e-3)
    # windows are independent: a permuted batch gives the permuted result, bit for bit
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
    res2 = ens.localize(xd[perm].contiguous(), p_on=2000.0, scale_mul=1000.0)
    assert torch.equal(res2["prob"], prob[perm])
    assert torch.equal(res2["state"], state[perm])
    assert torch.equal(res2["power"], power[perm])
    # and the internal chunk size is invisible
    ens.set_chunk_windows(96)
    res3 = ens.localize(xd, p_on=2000.0, scale_mul=1000.0)
    assert torch.equal(res3["prob"], prob) and torch.equal(res3["state"], state)
    assert 0.02 < float((prob > 0.5).double().mean()) < 0.98  # the votes are mixed: the test means something
    ens.close()


def test_config2_bf16_ensemble_fused_vs_fp32(torch_cuda):
    """configs[2] shape (E=5, L=1440) on 1024 windows: bf16 fused-block path against the fp32 per-layer path
    on windows whose fp32 vote is firm, permutation equivariance bit for bit."""
    torch = torch_cuda
    from camal_amd import synth
    from camal_amd.calibrate import calibrated_state_dicts
    from camal_amd.engine import Ensemble
    B, L, ks = 1024, 1440, [5, 7, 9, 15, 25]
    x = synth.refit_windows(B, L, seed=78)
    sds = calibrated_state_dicts(ks, seed=4, sample_x=x[:64], device="cuda:0", precision="fp32")
    xd = torch.from_numpy(x).cuda()
    e32 = Ensemble(sds, device="cuda:0", precision="fp32")
    r32 = e32.localize(xd, want_members=True)
    p32, s32 = r32["prob"].cpu().numpy(), r32["state"].cpu().numpy()
    lg32 = r32["member_logits"].cpu().numpy()
    e32.close()
    e16 = Ensemble(sds, device="cuda:0", precision="bf16")
    r16 = e16.localize(xd)
    p16, s16 = r16["prob"].clone(), r16["state"].clone()
    assert np.abs(p16.cpu().numpy() - p32).max() < 2e-2
    # firm: the ensemble vote and every member's own vote are away from 0.5 in fp32
    m1 = 1.0 / (1.0 + np.exp(lg32[..., 0] - lg32[..., 1]))
    firm = (np.abs(p32 - 0.5) > 5e-2) & (np.abs(m1 - 0.5) > 5e-2).all(axis=0)
    assert firm.sum() >= B // 4
    assert (s16.cpu().numpy()[firm] == s32[firm]).mean() >= STATE_AGREE_MIN
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(2)).cuda()
    r16p = e16.localize(xd[perm].contiguous())
    assert torch.equal(r16p["prob"], p16[perm]) and torch.equal(r16p["state"], s16[perm])
    e16.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_config4_week_window_vs_oracle(torch_cuda, precision):
    """configs[4]: L = 10080 (a week at 1 min), many time tiles per window, + power."""
    from camal_amd import synth
    from camal_amd.calibrate import calibrated_state_dicts
    from camal_amd.engine import Ensemble
    from oracle import camal_oracle as oracle
    B, L, ks = 3, 10080, [5, 25]
    x = synth.refit_windows(B, L, seed=79)
    sds = calibrated_state_dicts(ks, seed=5, sample_x=x, device="cuda:0", precision="fp32")
    ens = Ensemble(sds, device="cuda:0", precision=precision)
    res = ens.localize(x, p_on=1500.0, scale_mul=1000.0, want_members=True)
    states_ref, prob_ref, _avg, outs = oracle.camal_predict(sds, x, return_intermediate=True)
    tol_p, tol_c = (REL_TOL_FP32, REL_TOL_FP32) if precision == "fp32" else (2e-2, 3e-2)
    assert np.abs(res["prob"].cpu().numpy() - prob_ref[:, 0]).max() < tol_p
    cam = res["member_cam"].cpu().numpy()
    for e in range(len(ks)):
        assert rel_err(cam[e], outs[e][0]) < tol_c
    safe = np.abs(prob_ref[:, 0] - 0.5) > (2e-3 if precision == "fp32" else 5e-2)
    if precision == "fp32" and safe.any():
        assert (res["state"].cpu().numpy()[safe] == states_ref[safe]).mean() >= STATE_AGREE_MIN
    state = res["state"].cpu().numpy()
    ref_power = oracle.power_estimate(state, x, 1500.0, scale_mul=1000.0)
    assert np.allclose(res["power"].cpu().numpy(), ref_power, rtol=1e-6, atol=1e-3)
    ens.close()


def test_config0_full_size_vs_reference_fixture(torch_cuda):
    """configs[0] at its own size (E=5, k=5,7,9,15,25, 64 x 1024) against what the REFERENCE returned for these
    windows and weights (tests/golden/cfg0_l1024.npz): fp32 probabilities, CAM samples, averaged CAM, states; then
    the bf16 path's states over ALL timestamps of every window whose reference vote is >= 2e-3 away from 0.5."""
    from camal_amd import synth
    from camal_amd.engine import Ensemble
    from tests.helpers import case_state_dicts, load_case
    c = load_case("cfg0_l1024")
    sds = case_state_dicts(c)
    x = synth.ukdale_windows(int(c["n"]), int(c["length"]), seed=int(c["data_seed"]))
    st = int(c["cam_stride"])
    ens = Ensemble(sds, device="cuda:0", precision="fp32")
    res = ens.localize(x, want_avg_cam=True, want_members=True)
    prob, state = res["prob"].cpu().numpy(), res["state"].cpu().numpy()
    assert np.abs(prob - c["prob_detect"][:, 0]).max() < REL_TOL_FP32
    mcam, mlog = res["member_cam"].cpu().numpy(), res["member_logits"].cpu().numpy()
    for e in range(len(sds)):
        err = np.abs(mcam[e][:, ::st] - c["cams_strided"][e]).max(axis=-1) / np.abs(c["cam_row_max"][e])
        assert err.max() < REL_TOL_FP32, (e, err.max())
        p1 = 1.0 / (1.0 + np.exp(mlog[e][:, 0].astype(np.float64) - mlog[e][:, 1]))
        assert np.abs(p1 - c["probs"][e][:, 1]).max() < REL_TOL_FP32
    safe = np.abs(c["prob_detect"][:, 0] - 0.5) > 2e-3
    assert safe.sum() >= 60
    assert np.abs(res["avg_cam"].cpu().numpy()[safe][:, ::st] - c["avg_cam_strided"][safe]).max() < REL_TOL_FP32
    agree32 = (state[safe] == c["states"][safe]).mean()
    assert agree32 >= STATE_AGREE_MIN, agree32
    ens.close()
    e16 = Ensemble(sds, device="cuda:0", precision="bf16")
    r16 = e16.localize(x)
    p16, s16 = r16["prob"].cpu().numpy(), r16["state"].cpu().numpy()
    assert np.abs(p16 - c["prob_detect"][:, 0]).max() < 2e-2
    agree16 = (s16[safe] == c["states"][safe]).mean()  # every counted window, every timestamp
    print(f"cfg0 state agreement vs the reference: fp32 {agree32:.6f}, bf16 {agree16:.6f} over {int(safe.sum())} windows")
    assert agree16 >= STATE_AGREE_MIN, agree16
    e16.close()


def test_config4_week_windows_vs_reference_fixture(torch_cuda):
    """configs[4]'s length (L = 10080) against what the REFERENCE returned (tests/golden/week_l10080.npz: 6 REFIT-shaped
    windows, members k = 5 and 25): fp32 probabilities, CAM samples, averaged CAM and states; bf16 states on every window
    (none of them votes within 0.2 of the threshold)."""
    from camal_amd import synth
    from camal_amd.engine import Ensemble
    from tests.helpers import case_state_dicts, load_case
    c = load_case("week_l10080")
    sds = case_state_dicts(c)
    x = synth.refit_windows(int(c["n"]), int(c["length"]), seed=int(c["data_seed"]))
    st = int(c["cam_stride"])
    ens = Ensemble(sds, device="cuda:0", precision="fp32")
    res = ens.localize(x, want_avg_cam=True, want_members=True)
    prob, state = res["prob"].cpu().numpy(), res["state"].cpu().numpy()
    assert np.abs(prob - c["prob_detect"][:, 0]).max() < REL_TOL_FP32
    mcam = res["member_cam"].cpu().numpy()
    for e in range(len(sds)):
        err = np.abs(mcam[e][:, ::st] - c["cams_strided"][e]).max(axis=-1) / np.abs(c["cam_row_max"][e])
        assert err.max() < REL_TOL_FP32, (e, err.max())
    assert np.abs(c["prob_detect"][:, 0] - 0.5).min() > 0.2
    assert np.abs(res["avg_cam"].cpu().numpy()[:, ::st] - c["avg_cam_strided"]).max() < REL_TOL_FP32
    assert (state == c["states"]).mean() >= STATE_AGREE_MIN and (c["states"] != 0).sum() > 100
    ens.close()
    e16 = Ensemble(sds, device="cuda:0", precision="bf16")
    r16 = e16.localize(x)
    assert np.abs(r16["prob"].cpu().numpy() - c["prob_detect"][:, 0]).max() < 2e-2
    assert (r16["state"].cpu().numpy() == c["states"]).mean() >= STATE_AGREE_MIN
    e16.close()


def test_config2_fp32_ensemble_sample_vs_oracle_and_bf16_all_windows(torch_cuda):
    """configs[2] shape (E=5, 1024 x 1440): the fp32 ensemble against the ORACLE on a spread sample of windows (not a
    self-comparison), then the bf16 path against fp32 over ALL timestamps of ALL windows whose fp32 vote is >= 2e-3
    from 0.5 (the fp32 criterion), with the flip rate printed."""
    torch = torch_cuda
    from camal_amd import synth
    from camal_amd.calibrate import calibrated_state_dicts
    from camal_amd.engine import Ensemble
    from oracle import camal_oracle as oracle
    B, L, ks = 1024, 1440, [5, 7, 9, 15, 25]
    x = synth.refit_windows(B, L, seed=81)
    sds = calibrated_state_dicts(ks, seed=6, sample_x=x[:64], device="cuda:0", precision="fp32")
    xd = torch.from_numpy(x).cuda()
    e32 = Ensemble(sds, device="cuda:0", precision="fp32")
    r32 = e32.localize(xd, p_on=2000.0, scale_mul=1000.0)
    p32, s32 = r32["prob"].cpu().numpy(), r32["state"].cpu().numpy()
    idx = np.array([0, 1, 511, 512, 777, 1023])
    states_ref, prob_ref = oracle.camal_predict(sds, x[idx], return_prob=True)
    assert np.abs(p32[idx] - prob_ref[:, 0]).max() < REL_TOL_FP32
    safe = np.abs(prob_ref[:, 0] - 0.5) > 2e-3
    assert safe.sum() >= 4
    assert (s32[idx][safe] == states_ref[safe]).mean() >= STATE_AGREE_MIN
    ref_power = oracle.power_estimate(s32[idx], x[idx], 2000.0, scale_mul=1000.0)
    assert np.allclose(r32["power"].cpu().numpy()[idx], ref_power, rtol=1e-6, atol=1e-3)
    e32.close()
    e16 = Ensemble(sds, device="cuda:0", precision="bf16")
    r16 = e16.localize(xd)
    p16, s16 = r16["prob"].cpu().numpy(), r16["state"].cpu().numpy()
    counted = np.abs(p32 - 0.5) >= 2e-3
    agree = (s16[counted] == s32[counted]).mean()
    flips = int(((p16 > 0.5) != (p32 > 0.5))[counted].sum())
    print(f"configs[2] shape: bf16 vs fp32 states agree on {agree:.6f} of all timestamps of {int(counted.sum())}/{B} windows, "
          f"{flips} vote flips, max |dp| {np.abs(p16 - p32).max():.2e}")
    assert counted.sum() >= 0.95 * B
    assert agree >= STATE_AGREE_MIN, agree
    e16.close()
