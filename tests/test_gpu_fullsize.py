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
