"""GPU parity tests: the HIP path (through the C ABI) against the golden fixtures produced by
the reference itself and against the CPU oracle on seeded inputs.

Tolerances (north_star): fp32 path — detection probabilities and CAMs within 1e-3 relative
(CAM error normalised by the row's max |CAM|), on/off states identical on >= 99.9 % of
timestamps.  bf16 path — states identical on >= 99.9 % outside windows whose ensemble vote
sits within bf16 noise of the 0.5 threshold; probabilities within 2e-2 absolute.
"""
import numpy as np
import pytest

from tests.helpers import (REL_TOL_FP32, STATE_AGREE_MIN, case_state_dicts, golden_cases, load_case, rel_err,
                           resnetvar_state_dicts)

pytestmark = pytest.mark.gpu

CASES = golden_cases("ensemble_") + golden_cases("pair_")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _softmax(lg):
    z = lg - lg.max(-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(-1, keepdims=True)


@pytest.mark.parametrize("name", CASES)
def test_fp32_members_vs_reference(torch_cuda, name):
    from camal_amd.engine import Ensemble
    c = load_case(name)
    ens = Ensemble(case_state_dicts(c), device="cuda:0", precision="fp32")
    logits, cam = ens.forward_cam(c["data"].astype(np.float32))
    logits, cam = logits.cpu().numpy(), cam.cpu().numpy()
    for e in range(ens.E):
        assert rel_err(cam[e], c["cams"][e]) < REL_TOL_FP32
        assert rel_err(logits[e].T, c["logits"][e].T) < REL_TOL_FP32
        assert np.abs(_softmax(logits[e]) - c["probs"][e]).max() < REL_TOL_FP32
    ens.close()


@pytest.mark.parametrize("name", CASES)
def test_fp32_localize_vs_reference(torch_cuda, name):
    from camal_amd.engine import Ensemble
    c = load_case(name)
    data = c["data"]
    ens = Ensemble(case_state_dicts(c), device="cuda:0", precision="fp32")
    res = ens.localize(data.astype(np.float32), x64=data if data.dtype == np.float64 else None,
                       w_ma=int(c["w_ma"]), gate_mode=int(bool(c["gated"])), pred_f64=data.dtype == np.float64,
                       want_avg_cam=True)
    prob = res["prob"].cpu().numpy()
    state = res["state"].cpu().numpy()
    avg = res["avg_cam"].cpu().numpy()
    assert np.abs(prob - c["prob_detect"][:, 0]).max() < REL_TOL_FP32
    # windows whose ensemble probability sits on the 0.5 vote threshold may legitimately flip
    safe = np.abs(c["prob_detect"][:, 0] - 0.5) > 2e-3
    assert safe.sum() >= len(safe) // 2
    assert np.abs(avg[safe] - c["avg_cam"][safe]).max() < REL_TOL_FP32
    assert (state[safe] == c["states"][safe]).mean() >= STATE_AGREE_MIN
    # sign of zero follows np.round as well (-0.0 for small negative arguments)
    assert (np.signbit(state[safe]) == np.signbit(c["states"][safe])).mean() >= STATE_AGREE_MIN
    ens.close()


@pytest.mark.parametrize("L", [1, 2, 7, 31, 129, 1000])
def test_fp32_vs_oracle_ragged_lengths(torch_cuda, L):
    """Edge lengths: L < k (every tap clamps), L not a multiple of the 128/256-step tiles."""
    from camal_amd import synth, weights
    from camal_amd.engine import Ensemble
    from oracle import camal_oracle as oracle
    sds = [weights.make_state_dict(k, seed=5, head_gain=3.0) for k in (5, 25)]
    x = synth.ukdale_windows(3, max(L, 8), seed=L)[:, :L]
    ens = Ensemble(sds, device="cuda:0", precision="fp32")
    logits, cam = ens.forward_cam(x)
    for e, sd in enumerate(sds):
        cams, _labels, probs = oracle.cam_batch(sd, x, 1)
        assert rel_err(cam[e].cpu().numpy(), cams) < REL_TOL_FP32
        assert np.abs(_softmax(logits[e].cpu().numpy()) - probs).max() < REL_TOL_FP32
    ens.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_extreme_kernel_sizes_vs_oracle(torch_cuda, precision):
    """The C ABI admits odd first-kernel sizes 1..63: k = 1 (a 1x1 first conv), 31 and 63 (fused block tiles no
    longer fit the LDS for the largest: per-layer kernels, weight ring past the last out-channel tile)."""
    from camal_amd import synth, weights
    from camal_amd.engine import Ensemble
    from oracle import camal_oracle as oracle
    ks = (1, 31, 63)
    sds = [weights.make_state_dict(k, seed=8, head_gain=3.0) for k in ks]
    x = synth.ukdale_windows(3, 300, seed=21)
    ens = Ensemble(sds, device="cuda:0", precision=precision)
    logits, cam = ens.forward_cam(x)
    tol = REL_TOL_FP32 if precision == "fp32" else 5e-2
    for e, sd in enumerate(sds):
        cams, _labels, probs = oracle.cam_batch(sd, x, 1)
        assert rel_err(cam[e].cpu().numpy(), cams) < tol, (ks[e], rel_err(cam[e].cpu().numpy(), cams))
        assert np.abs(_softmax(logits[e].cpu().numpy()) - probs).max() < tol
    ens.close()


def test_chunking_is_invisible(torch_cuda):
    """B larger than the internal chunk gives the same numbers as one pass."""
    from camal_amd import synth, weights
    from camal_amd.engine import Ensemble
    sds = [weights.make_state_dict(k, seed=9, head_gain=3.0) for k in (7, 9)]
    x = synth.ukdale_windows(11, 200, seed=3)
    a = Ensemble(sds, device="cuda:0", precision="fp32")
    b = Ensemble(sds, device="cuda:0", precision="fp32", chunk_windows=4)
    ra = a.localize(x, p_on=2000.0, scale_mul=1000.0)
    rb = b.localize(x, p_on=2000.0, scale_mul=1000.0)
    for k in ("prob", "state", "power"):
        assert torch_cuda.equal(ra[k], rb[k])
    la, ca = a.forward_cam(x)
    lb, cb = b.forward_cam(x)
    assert torch_cuda.equal(la, lb) and torch_cuda.equal(ca, cb)


def test_power_and_cam_label0(torch_cuda):
    from camal_amd import synth, weights
    from camal_amd.engine import Ensemble
    from oracle import camal_oracle as oracle
    sds = [weights.make_state_dict(7, seed=2, head_gain=3.0)]
    x = synth.ukdale_windows(4, 300, seed=8)
    ens = Ensemble(sds, device="cuda:0", precision="fp32")
    res = ens.localize(x, p_on=2000.0, scale_mul=1000.0)
    state = res["state"].cpu().numpy()
    ref = oracle.power_estimate(state, x, 2000.0, scale_mul=1000.0)
    assert np.allclose(res["power"].cpu().numpy(), ref, rtol=1e-6, atol=1e-3)
    _, cam0 = ens.forward_cam(x, cam_label=0)
    ref0, _, _ = oracle.cam_batch(sds[0], x, 0)
    assert rel_err(cam0[0].cpu().numpy(), ref0) < REL_TOL_FP32


@pytest.mark.parametrize("name", ["ensemble_l256", "ensemble_l333_f64"])
def test_bf16_localize_vs_reference(torch_cuda, name):
    from camal_amd.engine import Ensemble
    c = load_case(name)
    data = c["data"]
    ens = Ensemble(case_state_dicts(c), device="cuda:0", precision="bf16")
    res = ens.localize(data.astype(np.float32), x64=data if data.dtype == np.float64 else None,
                       w_ma=int(c["w_ma"]), pred_f64=data.dtype == np.float64, want_members=True)
    prob = res["prob"].cpu().numpy()
    state = res["state"].cpu().numpy()
    assert np.abs(prob - c["prob_detect"][:, 0]).max() < 2e-2
    # members whose own vote is not within bf16 noise of 0.5 must agree on the label
    probs_m = _softmax(res["member_logits"].cpu().numpy())
    firm = np.abs(c["probs"][..., 1] - 0.5) > 5e-2
    assert ((probs_m[..., 1] > 0.5) == (c["labels"] > 0))[firm].all()
    safe = (np.abs(c["prob_detect"][:, 0] - 0.5) > 5e-2) & firm.all(axis=0)
    assert safe.sum() >= len(safe) // 2
    assert (state[safe] == c["states"][safe]).mean() >= STATE_AGREE_MIN
    ens.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("L", [5, 120, 121, 248, 249, 700])
def test_fused_blocks_equal_per_layer_kernels(torch_cuda, precision, L):
    """block_kernel (default) against one conv_kernel per ConvBlock on the same inputs, at lengths
    around the fused tile sizes (120 valid columns fp32, 248 bf16) so that first/last-tile edge
    fix-ups, single-tile windows and ragged last tiles are all hit."""
    from camal_amd import synth, weights
    from camal_amd.engine import Ensemble
    sds = [weights.make_state_dict(k, seed=17, head_gain=3.0) for k in (5, 25)]
    x = synth.ukdale_windows(5, max(L, 8), seed=100 + L)[:, :L]
    a = Ensemble(sds, device="cuda:0", precision=precision)
    b = Ensemble(sds, device="cuda:0", precision=precision)
    a.set_fused(True)
    b.set_fused(False)
    la, ca = a.forward_cam(x)
    lb, cb = b.forward_cam(x)
    ca, cb, la, lb = ca.cpu().numpy(), cb.cpu().numpy(), la.cpu().numpy(), lb.cpu().numpy()
    if precision == "fp32":
        # same products, same k-order per element: CAM rows agree to rounding of the 4-way channel-tile sum
        assert rel_err(ca, cb) < 1e-5
        assert rel_err(la.reshape(-1, 2).T, lb.reshape(-1, 2).T) < 1e-5
    else:
        # bf16: the per-layer path rounds the block-2 skip to bf16 in HBM, the fused one keeps it in fp32
        assert rel_err(ca, cb) < 3e-2
    a.close()
    b.close()


@pytest.mark.parametrize("ks", [(3, 9), (7, 15), (31, 5), "resnet"])
def test_fused_bf16_weight_stream_over_many_tiles(torch_cuda, ks):
    """More tiles than workgroups (300 windows x 3 tiles on 256 CUs): every workgroup of the fused bf16 kernels walks
    several tiles, so the endless weight ring wraps from the last layer of a tile into the first layer of the next
    (camal_kernels.hpp: WRing), for first kernel sizes that leave the ring on either slot phase; against the
    per-layer kernels, whose ring restarts with every layer."""
    from camal_amd import synth, weights
    from camal_amd.engine import Ensemble
    if ks == "resnet":  # classifiers/resnet.py's blocks (7 / 5 / 3, zero padding): the other instantiation of the kernels
        sds, ks = [weights.make_state_dict(7, seed=sd, arch=weights.ARCH_RESNET, head_gain=3.0) for sd in (23, 24)], (7, 7)
    else:
        sds = [weights.make_state_dict(k, seed=23, head_gain=3.0) for k in ks]
    x = synth.ukdale_windows(300, 500, seed=7)
    a = Ensemble(sds, device="cuda:0", precision="bf16")
    b = Ensemble(sds, device="cuda:0", precision="bf16")
    a.set_fused(True)
    b.set_fused(False)
    la, ca = a.forward_cam(x)
    lb, cb = b.forward_cam(x)
    ca, cb = ca.cpu().numpy(), cb.cpu().numpy()
    for e in range(len(ks)):
        assert rel_err(ca[e], cb[e]) < 3e-2, (ks[e], rel_err(ca[e], cb[e]))
    assert np.abs(la.cpu().numpy() - lb.cpu().numpy()).max() < 5e-2 * max(1.0, float(np.abs(lb.cpu().numpy()).max()))
    a.close()
    b.close()


# ---- classifiers/resnet.py's ResNet (kernels 7/5/3, zero "same" padding) through camal_create_arch -----------------
@pytest.mark.parametrize("name", golden_cases("resnetvar_"))
def test_resnet_variant_fp32_vs_reference(torch_cuda, name):
    from camal_amd.engine import Ensemble
    c = load_case(name)
    ens = Ensemble(resnetvar_state_dicts(c), device="cuda:0", precision="fp32")
    assert ens.arch == "resnet"
    logits, cam = ens.forward_cam(c["data"].astype(np.float32))
    logits, cam = logits.cpu().numpy(), cam.cpu().numpy()
    for e in range(ens.E):
        assert rel_err(cam[e], c["cams"][e]) < REL_TOL_FP32
        assert rel_err(logits[e].T, c["logits"][e].T) < REL_TOL_FP32
        assert np.abs(_softmax(logits[e]) - c["probs"][e]).max() < REL_TOL_FP32
    ens.close()


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("L", [5, 7, 64, 130, 249, 777])  # L >= w_ma: np.convolve(..., "same") is longer than the row below it
def test_resnet_variant_localize_vs_oracle(torch_cuda, precision, L, fused):
    """Ensemble body on two ResNet-variant members against the oracle (ragged lengths, both precisions; blocks 2 and 3
    as one kernel per ConvBlock and as fused-block kernels with the 5-tap second layer and zero padding)."""
    from camal_amd import synth, weights
    from camal_amd.engine import Ensemble
    from oracle import camal_oracle as oracle
    x = synth.ukdale_windows(6, L, seed=400 + L).astype(np.float32)
    sds = [weights.make_state_dict(7, seed=s, arch=weights.ARCH_RESNET, head_gain=3.0) for s in (311, 312)]
    ens = Ensemble(sds, device="cuda:0", precision=precision)
    ens.set_fused(fused)
    res = ens.localize(x, want_avg_cam=True, want_members=True)
    states, prob, avg_cam, _ = oracle.camal_predict(sds, x, return_intermediate=True)
    mcam = res["member_cam"].cpu().numpy()
    for e, sd in enumerate(sds):
        cams, _labels, _probs = oracle.cam_batch(sd, x, 1)
        assert rel_err(mcam[e], cams) < (REL_TOL_FP32 if precision == "fp32" else 3e-2)
    p = res["prob"].cpu().numpy()
    assert np.abs(p - prob[:, 0]).max() < (REL_TOL_FP32 if precision == "fp32" else 2e-2)
    if precision == "fp32":
        safe = np.abs(prob[:, 0] - 0.5) > 2e-3
        assert (res["state"].cpu().numpy()[safe] == states[safe]).mean() >= STATE_AGREE_MIN
        assert np.abs(res["avg_cam"].cpu().numpy()[safe] - avg_cam[safe]).max() < REL_TOL_FP32
    ens.close()


def test_resnet_variant_mirror_class(torch_cuda):
    from camal_amd import ResNet
    c = load_case("resnetvar_l192")
    m = ResNet()
    m.to("cuda:0")
    m.load_state_dict(resnetvar_state_dicts(c)[0])
    m.eval()
    lg = m(c["data"].astype(np.float32)[:, None, :]).cpu().numpy()
    assert rel_err(lg.T, c["logits"][0].T) < REL_TOL_FP32
    # train() mode: batch-statistics forward through the training entry points, against the torch restatement
    import torch
    from oracle.torch_train import FunctionalResNet
    m.train()
    lt = m(c["data"].astype(np.float32)[:, None, :]).cpu().numpy()
    with torch.no_grad():
        ref = FunctionalResNet(resnetvar_state_dicts(c)[0]).forward(torch.from_numpy(c["data"].astype(np.float32))).numpy()
    assert rel_err(lt.T, ref.T) < REL_TOL_FP32


# ---- CamALResNet(mid_channels=32 / 128) through camal_create_widths ------------------------------------------------
@pytest.mark.parametrize("name", golden_cases("width"))
def test_other_widths_fp32_vs_reference(torch_cuda, name):
    from camal_amd.engine import Ensemble
    from tests.helpers import width_state_dicts
    c = load_case(name)
    sds = width_state_dicts(c)
    ens = Ensemble(sds, device="cuda:0", precision="fp32")
    assert ens.mid_channels == int(c["mid"])
    logits, cam = ens.forward_cam(c["data"].astype(np.float32))
    logits, cam = logits.cpu().numpy(), cam.cpu().numpy()
    for e in range(ens.E):
        assert rel_err(cam[e], c["cams"][e]) < REL_TOL_FP32
        assert rel_err(logits[e].T, c["logits"][e].T) < REL_TOL_FP32
        assert np.abs(_softmax(logits[e]) - c["probs"][e]).max() < REL_TOL_FP32
    ens.close()


@pytest.mark.parametrize("name", golden_cases("width"))
def test_other_widths_bf16_vs_reference(torch_cuda, name):
    """CamALResNet(mid_channels=32 / 128) on the bf16 MFMA kernels (one per ConvBlock: 32- and 256-channel layers walk
    one / eight K groups per tap) against the reference's fixtures, with the bf16 tolerances of the default widths."""
    from camal_amd.engine import Ensemble
    from tests.helpers import width_state_dicts
    c = load_case(name)
    sds = width_state_dicts(c)
    ens = Ensemble(sds, device="cuda:0", precision="bf16")
    logits, cam = ens.forward_cam(c["data"].astype(np.float32))
    logits, cam = logits.cpu().numpy(), cam.cpu().numpy()
    for e in range(ens.E):
        assert rel_err(cam[e], c["cams"][e]) < 3e-2, (e, rel_err(cam[e], c["cams"][e]))
        probs = _softmax(logits[e])
        assert np.abs(probs - c["probs"][e]).max() < 2e-2
        firm = np.abs(c["probs"][e][:, 1] - 0.5) > 5e-2
        assert ((probs[:, 1] > 0.5) == (c["labels"][e] > 0))[firm].all()
    ens.close()


@pytest.mark.parametrize("mid,L,k", [(32, 1, 5), (32, 300, 25), (32, 1031, 9), (32, 77, 1), (32, 130, 3), (128, 70, 25), (128, 257, 7),
                                     (128, 5, 41), (128, 600, 9)])
def test_other_widths_bf16_localize_vs_oracle(torch_cuda, mid, L, k):
    """ragged lengths, several tiles per window, first kernel sizes that leave every remainder of the 32-channel K loop
    (k mod 4 = 1, 3) and the narrow tiles of large kernels on 256 channels; the ensemble body on top"""
    from camal_amd import synth, weights
    from camal_amd.engine import Ensemble
    from oracle import camal_oracle as oracle
    x = synth.ukdale_windows(5, max(L, 8), seed=500 + L)[:, :L].astype(np.float32)
    sds = [weights.make_state_dict(kk, seed=s, mid_channels=mid, head_gain=3.0) for kk, s in ((k, 321), (7, 322))]
    ens = Ensemble(sds, device="cuda:0", precision="bf16")
    res = ens.localize(x, want_members=True)
    states, prob, avg_cam, outs = oracle.camal_predict(sds, x, return_intermediate=True)
    mcam = res["member_cam"].cpu().numpy()
    for e in range(len(sds)):
        assert rel_err(mcam[e], outs[e][0]) < 3e-2, (e, rel_err(mcam[e], outs[e][0]))
    assert np.abs(res["prob"].cpu().numpy() - prob[:, 0]).max() < 2e-2
    safe = np.abs(prob[:, 0] - 0.5) > 5e-2
    if L >= 5 and safe.any():
        assert (res["state"].cpu().numpy()[safe] == states[safe]).mean() >= STATE_AGREE_MIN
    ens.close()


@pytest.mark.parametrize("mid,L,k", [(32, 1, 5), (32, 300, 25), (32, 1031, 9), (128, 70, 25), (128, 257, 7), (128, 5, 41)])
def test_other_widths_localize_vs_oracle(torch_cuda, mid, L, k):
    """ragged lengths, several tiles per window, the extreme kernel size, the ensemble body on top"""
    from camal_amd import synth, weights
    from camal_amd.engine import Ensemble
    from oracle import camal_oracle as oracle
    x = synth.ukdale_windows(5, max(L, 8), seed=500 + L)[:, :L].astype(np.float32)
    sds = [weights.make_state_dict(kk, seed=s, mid_channels=mid, head_gain=3.0) for kk, s in ((k, 321), (7, 322))]
    ens = Ensemble(sds, device="cuda:0", precision="fp32")
    res = ens.localize(x, want_avg_cam=True, want_members=True)
    states, prob, avg_cam, outs = oracle.camal_predict(sds, x, return_intermediate=True)
    mcam = res["member_cam"].cpu().numpy()
    for e in range(len(sds)):
        assert rel_err(mcam[e], outs[e][0]) < REL_TOL_FP32
    assert np.abs(res["prob"].cpu().numpy() - prob[:, 0]).max() < REL_TOL_FP32
    safe = np.abs(prob[:, 0] - 0.5) > 2e-3
    if L >= 5 and safe.any():
        assert (res["state"].cpu().numpy()[safe] == states[safe]).mean() >= STATE_AGREE_MIN
    ens.close()


def test_other_widths_mirror_class_and_trainer(torch_cuda):
    """CamALResNet(mid_channels=32) on its own: eval forward, CAM.run, one epoch of the trainer, a 4-D test()-style use"""
    from camal_amd import CAM, CamALResNet, synth
    from camal_amd.trainer import BasedClassifTrainer, TSLoader
    from tests.helpers import width_state_dicts
    c = load_case("width32_l150")
    m = CamALResNet(kernel_size=5, mid_channels=32)
    m.to("cuda:0")
    m.load_state_dict(width_state_dicts(c)[0])
    m.eval()
    lg = m(c["data"].astype(np.float32)[:, None, :]).cpu().numpy()
    assert rel_err(lg.T, c["logits"][0].T) < REL_TOL_FP32
    cam = CAM(m, "cuda:0", last_conv_layer=m._modules["layers"][2], fc_layer_name=m._modules["linear"])
    got, label, proba = cam.run(instance=c["data"][1], returned_cam_for_label=1)
    assert rel_err(got[None], c["cams"][0][1:2]) < REL_TOL_FP32 and label == int(c["labels"][0][1])
    x, y = synth.ideal_windows(24, 64, seed=9)
    m.train()
    tr = BasedClassifTrainer(m, TSLoader(x, y, batch_size=8, shuffle=True), valid_loader=TSLoader(x[:8], y[:8], batch_size=1),
                             device="cuda:0", weight_decay=0)
    tr.train(2)
    assert len(tr.loss_valid_history) == 2 and np.isfinite(tr.loss_train_history).all()
    assert m.state_dict()["linear.weight"].shape == (2, 64)
