"""GPU: the HIP training step (camal_train_fwd_bwd / camal_adamw_step through the C ABI and the
BasedClassifTrainer mirror) against fixtures from the reference's own trainer and against the
torch-autograd oracle.  Tolerance: gradients within 1e-3 relative (per tensor, normalised by the
tensor's max |g|); loss within 1e-4; loss history of a short run within 2e-3 relative."""
import numpy as np
import pytest

from camal_amd import weights
from tests.helpers import golden_cases, is_bn_fed_bias, load_case, pack_train_views, train_case_inputs

pytestmark = pytest.mark.gpu
CASES = golden_cases("train_")


def _engine(sd0, k, B, L):
    from camal_amd.trainer import TrainEngine
    from tests.helpers import model_for
    m = model_for(sd0)
    m.to("cuda:0")
    m.load_state_dict(sd0)
    return m, TrainEngine(m, "cuda:0", B, L)


def _grads_dict(eng, model):
    flat = eng.grads.cpu().numpy()
    out, o = {}, 0
    for kk in eng.keys:
        n = model._params[kk].size
        out[kk] = flat[o:o + n].reshape(model._params[kk].shape)
        o += n
    return out


@pytest.mark.parametrize("name", CASES)
def test_one_step_vs_reference_fixture(name):
    c = load_case(name)
    k, x, y, sd0 = train_case_inputs(c)
    B = int(c["batch"])
    m, eng = _engine(sd0, k, B, x.shape[1])
    loss, ncorrect = eng.forward_backward(x[:B], y[:B])
    assert abs(loss.item() - float(c["step1_loss"])) < 1e-4
    grads = _grads_dict(eng, m)
    keys = eng.keys
    norms, small, big = pack_train_views(keys, grads)
    live = np.array([not is_bn_fed_bias(kk) for kk in keys])
    assert np.allclose(norms[live], c["step1_grad_norms"][live], rtol=1e-3, atol=1e-7)
    assert norms[~live].max() == 0.0
    assert np.abs(small - c["step1_small_grads"]).max() < 1e-3 * np.abs(c["step1_small_grads"]).max()
    assert np.abs(big - c["step1_big_grad_slices"]).max() < 1e-3 * np.abs(c["step1_big_grad_slices"]).max()
    assert np.allclose(eng.running.cpu().numpy(), c["step1_running"], rtol=1e-4, atol=1e-6)
    logits = eng.forward_train(x[:B]).cpu().numpy()
    assert np.abs(logits - c["step1_logits"]).max() < 1e-3 * np.abs(c["step1_logits"]).max()
    eng.close()


# L > 128: several segments per window in the Cin = 1 weight gradient; k = 9 / 15 / 25: two to four tap groups in one
# weight-gradient launch; "resnet": classifiers/resnet.py's block (7/5/3, zero padding)
@pytest.mark.parametrize("k,B,L,arch", [(9, 10, 77, "camal"), (5, 6, 300, "camal"), (25, 5, 257, "camal"),
                                        (15, 3, 129, "camal"), (7, 9, 200, "resnet"), (7, 4, 131, "resnet"),
                                        # the bench's batch: several windows per weight-gradient split, ragged last split
                                        (25, 128, 64, "camal"), (15, 100, 40, "camal")])
def test_every_gradient_tensor_vs_oracle(k, B, L, arch):
    from camal_amd import synth
    from oracle import torch_train
    x, y = synth.ideal_windows(B, L, seed=5)
    sd0 = weights.make_state_dict(k, seed=31, head_gain=1.0, arch=arch)
    loss_ref, nc_ref, g_ref, sd1 = torch_train.loss_and_grads(sd0, x, y)
    m, eng = _engine(sd0, k, B, L)
    loss, ncorrect = eng.forward_backward(x, y)
    assert abs(loss.item() - loss_ref) < 1e-4 and int(ncorrect.item()) == nc_ref
    grads = _grads_dict(eng, m)
    gmax = max(np.abs(g).max() for g in g_ref.values())
    checked = 0
    for kk in eng.keys:
        # exact-zero gradients (rounding noise in torch): conv biases feeding a train-mode BN, and the
        # beta of the skip BNs of blocks 1/2 (a per-channel constant added to a block output is removed
        # by the next block's BatchNorms)
        if is_bn_fed_bias(kk) or np.abs(g_ref[kk]).max() < 1e-5 * gmax:
            assert np.abs(grads[kk]).max() < 1e-4 * gmax, kk
            continue
        err = np.abs(grads[kk] - g_ref[kk]).max() / np.abs(g_ref[kk]).max()
        assert err < 1e-3, (kk, err)
        checked += 1
    assert checked >= len(eng.keys) - 13
    eng.close()


# ragged last batches on a trainer created for batch_size = 128: the weight-gradient split count is not monotone
# in B (k = 15: B = 80 needs 80 splits, B = 128 only 64), the partial buffer must cover every B <= max_batch
@pytest.mark.parametrize("k,Bs", [(15, (80, 96, 128, 65)), (25, (45, 96, 87)), (7, (129, 192, 100))])
def test_ragged_batches_on_a_full_size_trainer(k, Bs):
    from camal_amd import synth
    from camal_amd.trainer import TrainEngine
    from oracle import torch_train
    from tests.helpers import model_for
    L = 36
    sd0 = weights.make_state_dict(k, seed=17, head_gain=1.0)
    m = model_for(sd0)
    m.to("cuda:0")
    m.load_state_dict(sd0)
    eng = TrainEngine(m, "cuda:0", 256 if max(Bs) > 128 else 128, L)
    for B in Bs:
        x, y = synth.ideal_windows(B, L, seed=B)
        loss_ref, nc_ref, g_ref, _ = torch_train.loss_and_grads(sd0, x, y)
        loss, ncorrect = eng.forward_backward(x, y, update_running=False)
        assert abs(loss.item() - loss_ref) < 1e-4 and int(ncorrect.item()) == nc_ref, B
        grads = _grads_dict(eng, m)
        for kk in ("layers.2.layers.0.layers.0.weight", "layers.1.layers.1.layers.0.weight", "layers.1.residual.0.weight",
                   "layers.0.layers.2.layers.0.weight", "linear.weight"):
            err = np.abs(grads[kk] - g_ref[kk]).max() / np.abs(g_ref[kk]).max()
            assert err < 1e-3, (B, kk, err)
    eng.close()


@pytest.mark.parametrize("name", CASES)
def test_trainer_mirror_vs_reference_history(name):
    from camal_amd.resnet import CamALResNet
    from camal_amd.trainer import BasedClassifTrainer, TSLoader
    c = load_case(name)
    k, x, y, sd0 = train_case_inputs(c)
    from tests.helpers import model_for
    m = model_for(sd0)
    m.load_state_dict(sd0)
    loader = TSLoader(x, y, batch_size=int(c["batch"]), shuffle=False)
    tr = BasedClassifTrainer(m, train_loader=loader, valid_loader=None, learning_rate=float(c["lr"]), weight_decay=0,
                             patience_es=None, patience_rlr=None, device="cuda:0", n_warmup_epochs=0)
    tr.train(int(c["epochs"]))
    assert np.allclose(tr.loss_train_history, c["loss_train_history"], rtol=2e-3)
    assert np.allclose(tr.accuracy_train_history, c["accuracy_train_history"], atol=1.5 / len(x))
    sd = {kk: v.numpy() for kk, v in m.state_dict().items()}
    assert np.abs(sd["linear.weight"] - c["final_linear_weight"]).max() < 1e-3
    run = np.concatenate([sd[kk].ravel() for kk in sd if kk.endswith("running_mean") or kk.endswith("running_var")])
    # torch's conv biases drift by ~lr per step on rounding-noise gradients (exact gradient: 0), which
    # shifts the pre-BN means it tracks; ours stay put.  Function-level agreement is checked below.
    assert np.allclose(run, c["final_running"], rtol=2e-2, atol=float(c["lr"]) * 2 * len(tr.loss_train_history) * (len(x) // int(c["batch"])))
    m.eval()
    logits = m(x[:int(c["batch"])]).cpu().numpy()
    assert np.abs(logits - c["final_eval_logits"]).max() < 2e-2 * max(1.0, np.abs(c["final_eval_logits"]).max())
    assert set(tr.log) >= {"best_model_state_dict", "optimizer_state_dict", "loss_train_history", "training_time"}


def test_camal_train_end_to_end(tmp_path):
    """CamAL.train (core.py:47-189): undersampling, one trainer per kernel size, ranking by test
    loss, camal_logs schema, then predict with the fitted ensemble."""
    import pickle
    from camal_amd import CamAL, synth
    x, y = synth.ideal_windows(96, 64, seed=3)
    camal = CamAL(n_predictors=2, device="cuda:0")
    path = tmp_path / "logs.pkl"
    camal.train((x[:64], y[:64]), (x[64:80], y[64:80]), (x[80:], y[80:]), list_kernel_sizes=[5, 7, 9],
                path=str(path), epochs=2, batch_size=16)
    logs = camal.camal_logs
    assert logs["is_fitted"] and len(logs["camal_predictors_id"]) == 2
    assert set(logs["ensemble_training_logs"]) == {"Predictor_0", "Predictor_1", "Predictor_2"}
    kept = [n for n, l in logs["ensemble_training_logs"].items() if "best_model_state_dict" in l]
    assert sorted(kept) == sorted(logs["camal_predictors_id"])
    for n in kept:
        assert len(logs["ensemble_training_logs"][n]["best_model_state_dict"]) == 79
    states, prob = camal.predict(x[80:], return_prob=True)
    assert states.shape == (16, 64) and prob.shape == (16, 1) and np.isfinite(prob).all()
    fresh = CamAL(n_predictors=2, device="cuda:0")
    fresh.load(str(path))
    s2 = fresh.predict(x[80:])
    assert np.array_equal(states, s2)
    assert pickle.load(open(path, "rb"))["training_time"] >= 0


def test_trainer_with_validation_vs_reference():
    """tests/golden/trainer_valid_l64.npz: the reference's BasedClassifTrainer with a batch-1 validation loader,
    shuffle=True, ReduceLROnPlateau(patience 1), early stopping (patience 2) after one warm-up epoch
    (torch_trainer.py:92-171).  Same shuffled batches (torch RNG), same LR sequence, same stop and best epochs;
    histories within 2e-2 relative (six epochs of AdamW at lr 5e-3 amplify float32 rounding differences)."""
    import torch
    from camal_amd.trainer import BasedClassifTrainer, TSLoader
    from tests.helpers import model_for, trainer_valid_inputs
    c = load_case("trainer_valid_l64")
    k, seed = int(c["k"]), int(c["seed"])
    xt, yt, xv, yv = trainer_valid_inputs(seed, int(c["n_train"]), int(c["n_valid"]), int(c["length"]))
    sd0 = weights.make_state_dict(k, seed=seed, head_gain=1.0)
    m = model_for(sd0)
    m.load_state_dict(sd0)
    torch.manual_seed(int(c["shuffle_seed"]))
    tr = BasedClassifTrainer(m, train_loader=TSLoader(xt, yt, batch_size=int(c["batch"]), shuffle=True),
                             valid_loader=TSLoader(xv, yv, batch_size=1), learning_rate=float(c["lr"]), weight_decay=0,
                             patience_es=2, patience_rlr=1, device="cuda:0", n_warmup_epochs=1)
    lrs = []
    orig = tr.scheduler.step

    def step(v):
        lrs.append(orig(v))
        return lrs[-1]

    tr.scheduler.step = step
    tr.train(int(c["epochs"]))
    assert len(tr.loss_train_history) == int(c["epochs_run"]) < int(c["epochs"])  # stopped early, at the same epoch
    assert np.array_equal(np.array(lrs), c["lrs"]) and lrs[-1] < lrs[0]            # the LR drop at the same epoch
    assert tr.passed_epochs == int(c["passed_epochs"])
    assert tr.log["epoch_best_loss"] == int(c["epoch_best_loss"])
    # the first (warm-up) validation loss is a batch-statistics artefact of ~50: compare relative to each value
    assert np.allclose(tr.loss_valid_history, c["loss_valid_history"], rtol=2e-2), (tr.loss_valid_history, c["loss_valid_history"])
    assert np.allclose(tr.loss_train_history, c["loss_train_history"], rtol=2e-2)
    assert np.allclose(tr.accuracy_valid_history, c["accuracy_valid_history"], atol=1.5 / len(yv))
    assert abs(tr.log["value_best_loss"] - float(c["value_best_loss"])) < 2e-2 * float(c["value_best_loss"])
    loss, _ = tr.evaluate(TSLoader(xv, yv, batch_size=1))
    assert abs(loss - float(c["eval_loss"])) < 2e-2 * float(c["eval_loss"])


def test_camal_train_members_side_by_side():
    """CamAL.member_streams: the candidates train concurrently on one GPU (a host thread + HIP stream + shuffle
    generator each).  Reproducible under torch.manual_seed, same selection logic and log schema as the sequential
    loop, and each member's result equals training that member ALONE with the same generator."""
    import torch
    from camal_amd import CamAL, synth
    from camal_amd.resnet import CamALResNet
    from camal_amd.trainer import BasedClassifTrainer, TSLoader
    x, y = synth.ideal_windows(96, 64, seed=3)
    runs = []
    for _ in range(2):
        torch.manual_seed(11)
        camal = CamAL(n_predictors=2, device="cuda:0")
        camal.member_streams = 3
        camal.train((x[:64], y[:64]), (x[64:80], y[64:80]), (x[80:], y[80:]), list_kernel_sizes=[5, 7, 9], epochs=3, batch_size=16)
        runs.append(camal.camal_logs)
    a, b = runs
    assert a["camal_predictors_id"] == b["camal_predictors_id"] and len(a["camal_predictors_id"]) == 2
    assert set(a["ensemble_training_logs"]) == {"Predictor_0", "Predictor_1", "Predictor_2"}
    for n in a["ensemble_training_logs"]:
        la, lb = a["ensemble_training_logs"][n], b["ensemble_training_logs"][n]
        assert la["loss_train_history"] == lb["loss_train_history"] and la["loss_valid_history"] == lb["loss_valid_history"]
        assert ("best_model_state_dict" in la) == (n in a["camal_predictors_id"])
    # member 1 (k = 7) trained ALONE from the same initial weights with the same generator: same histories
    torch.manual_seed(11)
    base = int(torch.empty((), dtype=torch.int64).random_().item())
    models = [CamALResNet(kernel_size=k) for k in (5, 7, 9)]
    gen = torch.Generator().manual_seed((base + 7919 * 1) % (1 << 62))
    from camal_amd.core import undersampler
    xu, yu = undersampler(x[:64], y[:64], sampling_strategy="auto", seed=0)  # what CamAL.train feeds its trainers
    tr = BasedClassifTrainer(models[1], train_loader=TSLoader(xu, yu, batch_size=16, shuffle=True, generator=gen),
                             valid_loader=TSLoader(x[64:80], y[64:80], batch_size=1, generator=gen), learning_rate=1e-3,
                             weight_decay=0, patience_es=5, patience_rlr=3, n_warmup_epochs=1, device="cuda:0")
    tr.train(3)
    l1 = a["ensemble_training_logs"]["Predictor_1"]
    assert tr.loss_train_history == l1["loss_train_history"] and tr.loss_valid_history == l1["loss_valid_history"]
    states = camal.predict(x[80:])
    assert states.shape == (16, 64)


def test_criterion_arguments_and_reserved_workspace():
    """criterion / valid_criterion of the reference's constructor (torch_trainer.py:20,23,52-56): the default
    cross-entropy object is accepted (and runs in the library), another training criterion is taken as it is
    (test_custom_training_criterion_vs_oracle) and becomes the validation criterion too, a caller's validation criterion
    is applied to the eval-mode logits; evaluation losses equal torch's cross-entropy on the same logits."""
    import torch
    from camal_amd import synth
    from camal_amd.resnet import CamALResNet
    from camal_amd.trainer import BasedClassifTrainer, TSLoader
    x, y = synth.ideal_windows(40, 64, seed=8)
    torch.manual_seed(3)
    m = CamALResNet(kernel_size=5)
    tl, vl = TSLoader(x[:32], y[:32], batch_size=16), TSLoader(x[32:], y[32:], batch_size=1)
    smooth = torch.nn.CrossEntropyLoss(label_smoothing=0.1)
    tr0 = BasedClassifTrainer(m, tl, criterion=smooth, device="cuda:0")
    assert tr0.train_criterion is smooth and tr0.valid_criterion is smooth
    tr0.engine.close()
    tr = BasedClassifTrainer(m, tl, valid_loader=vl, criterion=torch.nn.CrossEntropyLoss(), device="cuda:0", weight_decay=0)
    assert tr.train_criterion is None and tr.valid_criterion is None  # the library's own cross-entropy
    tr.train(1)
    loss, metrics = tr.evaluate(vl)
    m.eval()
    logits = m(x[32:]).float()
    labels = torch.as_tensor(y[32:]).to(logits.device).long()
    ref = torch.nn.functional.cross_entropy(logits, labels, reduction="none").cpu().numpy()
    assert abs(loss - ref.mean()) < 1e-6 and abs(tr.loss_valid_history[0] - ref.mean()) < 1e-6
    assert metrics["ACCURACY"] == round(float((logits.argmax(1) == labels).float().mean()), 5)
    # a validation criterion of the caller's: summed instead of averaged, batches of 4
    m.train()
    tr2 = BasedClassifTrainer(m, tl, valid_loader=TSLoader(x[32:], y[32:], batch_size=4), device="cuda:0",
                              valid_criterion=torch.nn.CrossEntropyLoss(reduction="sum"))
    l2, _ = tr2.evaluate(TSLoader(x[32:], y[32:], batch_size=4))
    m.eval()
    ref2 = torch.nn.functional.cross_entropy(m(x[32:]).float(), labels, reduction="none").cpu().numpy()
    assert abs(l2 - ref2.reshape(2, 4).sum(axis=1).mean()) < 1e-4


def test_reserve_keeps_the_call_path_free_of_allocations():
    """camal_reserve: the workspace is sized once; smaller and equal shapes afterwards give the same numbers as a
    fresh context (and the library's workspace pointer does not move: observed through HIP's free-memory counter)."""
    import torch
    from camal_amd import synth, weights
    from camal_amd.engine import Ensemble
    sds = [weights.make_state_dict(k, seed=4, head_gain=3.0) for k in (5, 9)]
    x = synth.ukdale_windows(24, 300, seed=6)
    a = Ensemble(sds, device="cuda:0", precision="fp32")
    a.reserve(24, 300)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    ra = a.localize(x, p_on=2000.0, scale_mul=1000.0)
    rb = a.localize(x[:7, :128].copy(), p_on=2000.0, scale_mul=1000.0)
    torch.cuda.synchronize()
    outputs = sum(t.numel() * t.element_size() for r in (ra, rb) for t in r.values()) + x.nbytes + x[:7, :128].nbytes
    assert free0 - torch.cuda.mem_get_info()[0] <= outputs + (8 << 20)  # only the callers' tensors (torch's pool granularity)
    b = Ensemble(sds, device="cuda:0", precision="fp32")
    rc = b.localize(x, p_on=2000.0, scale_mul=1000.0)
    for k in ("prob", "state", "power"):
        assert torch.equal(ra[k], rc[k])
    with pytest.raises(Exception):
        a.reserve(0, 10)
    a.close()
    b.close()


@pytest.mark.parametrize("k,B,L", [(7, 12, 96), (25, 9, 70)])
def test_custom_training_criterion_vs_oracle(k, B, L):
    """BasedClassifTrainer(criterion=...) (torch_trainer.py:20,52,290): a criterion other than the default
    cross-entropy — class weights + label smoothing — is evaluated by torch on the [B, 2] logits and the library runs
    the backward pass from its gradient (camal_train_bwd_dlogits): loss, every gradient tensor and the running
    statistics against torch autograd through the whole model; then two epochs through the trainer."""
    import torch
    from camal_amd import synth
    from camal_amd.trainer import BasedClassifTrainer, TSLoader
    from oracle import torch_train
    from tests.helpers import model_for
    x, y = synth.ideal_windows(B, L, seed=6)
    sd0 = weights.make_state_dict(k, seed=32, head_gain=1.0)
    w = torch.tensor([0.3, 1.7])
    loss_ref, nc_ref, g_ref, sd1 = torch_train.loss_and_grads(sd0, x, y, criterion=torch.nn.CrossEntropyLoss(weight=w, label_smoothing=0.1))
    crit = torch.nn.CrossEntropyLoss(weight=w.cuda(), label_smoothing=0.1)
    m, eng = _engine(sd0, k, B, L)
    loss, ncorrect = eng.forward_backward(x, y, criterion=crit)
    assert abs(loss.item() - loss_ref) < 1e-4 and int(ncorrect.item()) == nc_ref
    grads = _grads_dict(eng, m)
    gmax = max(np.abs(g).max() for g in g_ref.values())
    checked = 0
    for kk in eng.keys:
        if is_bn_fed_bias(kk) or np.abs(g_ref[kk]).max() < 1e-5 * gmax:
            assert np.abs(grads[kk]).max() < 1e-4 * gmax, kk
            continue
        err = np.abs(grads[kk] - g_ref[kk]).max() / np.abs(g_ref[kk]).max()
        assert err < 1e-3, (kk, err)
        checked += 1
    assert checked >= len(eng.keys) - 13
    eng.store_to_model(m)
    for kk in ("layers.2.layers.2.layers.1.running_mean", "layers.0.residual.1.running_var"):
        assert np.allclose(m._params[kk], sd1[kk], rtol=1e-4, atol=1e-6)
    eng.close()
    # the same criterion through the trainer: it is the validation criterion too unless one is given (:52-56)
    m2 = model_for(sd0)
    m2.to("cuda:0")
    m2.load_state_dict(sd0)
    tr = BasedClassifTrainer(m2, TSLoader(x, y, batch_size=5, shuffle=True), valid_loader=TSLoader(x[:4], y[:4], batch_size=1),
                             criterion=crit, device="cuda:0", weight_decay=0)
    assert tr.train_criterion is crit and tr.valid_criterion is crit
    tr.train(2)
    assert len(tr.loss_train_history) == 2 and np.isfinite(tr.loss_train_history).all() and np.isfinite(tr.loss_valid_history).all()
    tr.engine.close()


def test_resident_training_set_equals_per_batch_copies():
    """The trainer keeps a TSLoader's windows in HBM and gathers every batch there from the epoch's permutation
    (TrainEngine.make_resident); `resident = False` copies batch by batch from the host like the reference
    (torch_trainer.py:284-285).  Same shuffles, same numbers, bit for bit."""
    import torch
    from camal_amd import synth
    from camal_amd.resnet import CamALResNet
    from camal_amd.trainer import BasedClassifTrainer, TSLoader
    x, y = synth.ideal_windows(50, 72, seed=12)

    def run(resident):
        torch.manual_seed(21)
        m = CamALResNet(kernel_size=9)
        tr = BasedClassifTrainer(m, TSLoader(x[:44], y[:44], batch_size=16, shuffle=True),
                                 valid_loader=TSLoader(x[44:], y[44:], batch_size=1), device="cuda:0", weight_decay=0)
        tr.resident = resident
        tr.train(3)
        assert (tr._resident is not None) == resident
        tr.engine.close()
        return tr.loss_train_history, tr.loss_valid_history, {k: v.numpy() for k, v in m.state_dict().items()}

    a, b = run(True), run(False)
    assert a[0] == b[0] and a[1] == b[1]
    assert all(np.array_equal(a[2][k], b[2][k]) for k in a[2])


def test_camal_train_over_devices_from_one_process():
    """CamAL.devices for train(): the candidates go round robin over the listed GPUs, one host thread each, no gradient
    exchange.  One GPU here: two entries for cuda:0 stand in for two GPUs; every member's histories equal the
    member_streams run with the same seed (same per-candidate generators), and predict then shards over the same list."""
    import torch
    from camal_amd import CamAL, synth
    x, y = synth.ideal_windows(96, 64, seed=3)
    logs = []
    for mode in ("devices", "streams"):
        torch.manual_seed(11)
        camal = CamAL(n_predictors=2, device="cuda:0")
        if mode == "devices":
            camal.devices = ["cuda:0", "cuda:0"]
        else:
            camal.member_streams = 2
        camal.train((x[:64], y[:64]), (x[64:80], y[64:80]), (x[80:], y[80:]), list_kernel_sizes=[5, 7, 9], epochs=2, batch_size=16)
        logs.append(camal.camal_logs)
        if mode == "devices":
            assert len(camal._ensembles) == 2
            assert camal.predict(x[80:]).shape == (16, 64)
    a, b = logs
    assert a["camal_predictors_id"] == b["camal_predictors_id"]
    for n in a["ensemble_training_logs"]:
        la, lb = a["ensemble_training_logs"][n], b["ensemble_training_logs"][n]
        assert la["loss_train_history"] == lb["loss_train_history"] and la["loss_valid_history"] == lb["loss_valid_history"]


def test_trainer_takes_the_references_dataloaders():
    """DataLoader(TSDataset(X, y), batch_size, shuffle=True) as CamAL.train builds them in the reference (core.py:92-119)
    handed to BasedClassifTrainer: the same run, number for number, as with TSLoader (same shuffles under manual_seed)."""
    import torch
    from camal_amd import synth
    from camal_amd.resnet import CamALResNet
    from camal_amd.trainer import BasedClassifTrainer, TSDataset, TSLoader
    x, y = synth.ideal_windows(50, 64, seed=21)
    xv, yv = synth.ideal_windows(10, 64, seed=22)
    runs = []
    for kind in ("tsloader", "dataloader"):
        torch.manual_seed(5)
        m = CamALResNet(kernel_size=5)
        if kind == "tsloader":
            tl, vl, te = TSLoader(x, y, batch_size=16, shuffle=True), TSLoader(xv, yv, batch_size=1), TSLoader(xv, yv, batch_size=1)
        else:
            tl = torch.utils.data.DataLoader(TSDataset(x, y), batch_size=16, shuffle=True)
            vl = torch.utils.data.DataLoader(TSDataset(xv, yv), batch_size=1, shuffle=False)
            te = torch.utils.data.DataLoader(TSDataset(xv, yv), batch_size=1, shuffle=False)
        tr = BasedClassifTrainer(m, train_loader=tl, valid_loader=vl, learning_rate=1e-3, weight_decay=0, device="cuda:0")
        tr.train(3)
        loss, metrics = tr.evaluate(te)
        runs.append((tr.loss_train_history, tr.loss_valid_history, loss, metrics))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2] and runs[0][3] == runs[1][3]
