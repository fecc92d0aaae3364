#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run only in the dev container (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

It imports the reference's own modules (``CamALResNet``, ``CAM``, ``CamAL``,
``BasedClassifTrainer``) from ``/root/reference``, loads weights produced by this
repo's seeded generator (``camal_amd.weights.make_state_dict``) into them, runs them on
this repo's seeded synthetic windows and stores ONLY inputs' seeds/calibration constants
and the reference's outputs.  No reference source travels.

``src.models.camal.core`` imports ``imblearn`` (absent here) through
``data_processing``; ``predict/eval/test`` never use it, so an empty placeholder module
is registered for the import line only.  ``CamAL.train`` is therefore not used here;
training fixtures drive ``BasedClassifTrainer`` directly.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from camal_amd import synth, weights  # noqa: E402

if "imblearn" not in sys.modules:
    try:
        import imblearn  # noqa: F401
    except Exception:
        m = types.ModuleType("imblearn")
        mu = types.ModuleType("imblearn.under_sampling")
        mu.RandomUnderSampler = object
        m.under_sampling = mu
        sys.modules["imblearn"] = m
        sys.modules["imblearn.under_sampling"] = mu

from src.helpers.class_activation_map import CAM  # noqa: E402
from src.models.camal.classifiers.camal_resnet import CamALResNet  # noqa: E402
from src.models.camal.core import CamAL  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)

KERNELS = [5, 7, 9, 15, 25]


def to_torch_sd(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def ref_model(sd):
    m = CamALResNet(kernel_size=weights.kernel_size_of(sd))
    m.load_state_dict(to_torch_sd(sd))
    m.eval()
    return m


def calibrate_heads(data, seed, spread):
    """Pick linear.weight's gain and linear.bias per member so that the logit difference
    l1-l0 over `data` has median 0 and standard deviation `spread` (probabilities then
    spread around 0.5 without saturating).  Uses the reference model."""
    gains, biases = [], []
    for k in KERNELS:
        sd = weights.make_state_dict(k, seed=seed, head_gain=1.0)
        with torch.no_grad():
            lg = ref_model(sd)(torch.from_numpy(data[:, None, :]).float()).numpy()
        d = (lg[:, 1] - lg[:, 0]).astype(np.float64)
        g = float(np.float32(spread / max(d.std(), 1e-6)))
        gains.append(g)
        biases.append((0.0, float(np.float32(-g * np.median(d)))))
    return gains, biases


def run_case(name, data, seed, gain, w_ma=5, gated=False, members=None):
    members = members or KERNELS
    data32 = data.astype(np.float32)
    gains, biases = calibrate_heads(data32, seed, gain)
    gains = [gains[KERNELS.index(k)] for k in members]
    biases = [biases[KERNELS.index(k)] for k in members]
    sds = [weights.make_state_dict(k, seed=seed, head_gain=g, head_bias=b)
           for k, g, b in zip(members, gains, biases)]

    # --- per member: reference CAM.run window by window (class_activation_map.py:34) ---
    n, length = data.shape
    cams = np.zeros((len(members), n, length), np.float32)
    labels = np.zeros((len(members), n), np.int32)
    probs = np.zeros((len(members), n, 2), np.float32)
    logits = np.zeros((len(members), n, 2), np.float32)
    for e, sd in enumerate(sds):
        m = ref_model(sd)
        cam = CAM(model=m, device="cpu", last_conv_layer=m._modules["layers"][2],
                  fc_layer_name=m._modules["linear"])
        for i in range(n):
            c, lab, p = cam.run(instance=data[i], returned_cam_for_label=1)
            cams[e, i], labels[e, i], probs[e, i] = c, lab, p
        with torch.no_grad():
            logits[e] = m(torch.from_numpy(data32[:, None, :])).numpy()

    # --- ensemble: reference CamAL.predict on hand-built camal_logs (core.py:252) ---
    logs = {"ensemble_training_logs": {}, "camal_predictors_id": [], "is_fitted": True}
    for e, (k, sd) in enumerate(zip(members, sds)):
        nm = f"Predictor_{e}"
        logs["ensemble_training_logs"][nm] = {"kernel_size": k, "best_model_state_dict": to_torch_sd(sd)}
        logs["camal_predictors_id"].append(nm)
    camal = CamAL(n_predictors=len(members), device="cpu")
    camal.camal_logs = logs
    camal.is_fitted = True
    captured = {}
    orig_att = camal.attention_sigmoid_module

    def capture(d, soft_label, w_ma=5):  # records the reference's own intermediate
        captured["avg_cam"] = np.array(soft_label)
        return orig_att(d, soft_label, w_ma=w_ma)

    camal.attention_sigmoid_module = capture
    y = None
    if gated:
        # the reference's device != 'cpu' branch (core.py:292-293) computes y from member 0
        y = camal._predict_subsequence_class(ref_model(sds[0]), data)
    states, prob_detect = camal.predict(data, y=y, return_prob=True, w_ma=w_ma)

    out = dict(
        members=np.array(members), seed=np.array(seed), head_gain=np.array(gains, np.float64),
        head_bias=np.array(biases, np.float64), w_ma=np.array(w_ma), gated=np.array(gated),
        data=data, cams=cams, labels=labels, probs=probs, logits=logits,
        avg_cam=captured["avg_cam"], states=states, prob_detect=prob_detect,
    )
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    on = float((states != 0).mean())
    print(f"{name}: N={n} L={length} members={members} labels1={labels.mean():.2f} "
          f"prob_detect in [{prob_detect.min():.3f},{prob_detect.max():.3f}] states_on={on:.3f} "
          f"size={os.path.getsize(os.path.join(HERE, name + '.npz')) / 1024:.0f} KB")


def train_case(name, k, seed, n, length, batch, epochs, lr=1e-3, arch=weights.ARCH_CAMAL, mid=64):
    """Training fixtures from the reference's own classes: one backward pass of CamALResNet in
    train() mode under nn.CrossEntropyLoss (torch_trainer.py:288-291), and `epochs` epochs of
    BasedClassifTrainer.train (AdamW, torch_trainer.py:92-171,273-300) on an unshuffled loader."""
    from src.helpers.torch_trainer import BasedClassifTrainer
    from src.helpers.torch_dataset import TSDataset
    x, y = synth.ideal_windows(n, length, seed=seed)
    sd0 = weights.make_state_dict(k, seed=seed, head_gain=1.0, arch=arch, mid_channels=mid)
    keys = weights.trainable_keys(k, arch=arch, mid_channels=mid)
    if arch == weights.ARCH_RESNET:  # classifiers/resnet.py's ResNet under the same trainer
        from src.models.camal.classifiers.resnet import ResNet
        assert k == 7
        make_model = ResNet
    else:
        make_model = lambda: CamALResNet(kernel_size=k, mid_channels=mid)  # noqa: E731

    # (1) one forward/backward on the first batch
    m = make_model()
    m.load_state_dict(to_torch_sd(sd0))
    m.train()
    xb = torch.from_numpy(x[:batch, None, :]).float()
    yb = torch.from_numpy(y[:batch]).float()
    logits = m(xb)
    loss = torch.nn.CrossEntropyLoss()(logits.float(), yb.long())
    loss.backward()
    grads = {kk: p.grad.numpy().copy() for kk, p in m.named_parameters()}
    assert list(grads) == keys
    small = [kk for kk in keys if grads[kk].size <= 2048 and not kk.endswith("layers.0.bias") and not kk.endswith("residual.0.bias")]
    sd1 = {kk: v.numpy().copy() for kk, v in m.state_dict().items()}
    run_keys = [kk for kk in sd1 if kk.endswith("running_mean") or kk.endswith("running_var")]
    out = dict(
        k=np.array(k), seed=np.array(seed), n=np.array(n), length=np.array(length), batch=np.array(batch),
        resnet_variant=np.array(arch == weights.ARCH_RESNET), mid=np.array(mid),
        epochs=np.array(epochs), lr=np.array(lr),
        step1_loss=np.array(float(loss)), step1_logits=logits.detach().numpy(),
        step1_grad_norms=np.array([np.linalg.norm(grads[kk].astype(np.float64)) for kk in keys]),
        step1_small_grads=np.concatenate([grads[kk].ravel() for kk in small]),
        step1_big_grad_slices=np.concatenate([grads[kk].ravel()[::97] for kk in keys if grads[kk].size > 2048]),
        step1_running=np.concatenate([sd1[kk].ravel() for kk in run_keys]),
    )

    # (2) the reference trainer for a few epochs
    m2 = make_model()
    m2.load_state_dict(to_torch_sd(sd0))
    loader = torch.utils.data.DataLoader(TSDataset(x, y), batch_size=batch, shuffle=False)
    tr = BasedClassifTrainer(m2, train_loader=loader, valid_loader=None, learning_rate=lr, weight_decay=0,
                             patience_es=None, patience_rlr=None, device="cpu", all_gpu=False, n_warmup_epochs=0)
    tr.train(epochs)
    sd2 = {kk: v.numpy().copy() for kk, v in m2.state_dict().items()}
    fn_keys = [kk for kk in keys if not kk.endswith("layers.0.bias") and not kk.endswith("residual.0.bias")]
    out.update(
        loss_train_history=np.array(tr.loss_train_history), accuracy_train_history=np.array(tr.accuracy_train_history),
        final_param_norms=np.array([np.linalg.norm(sd2[kk].astype(np.float64)) for kk in fn_keys]),
        final_param_delta_norms=np.array([np.linalg.norm((sd2[kk] - sd0[kk]).astype(np.float64)) for kk in fn_keys]),
        final_linear_weight=sd2["linear.weight"], final_running=np.concatenate([sd2[kk].ravel() for kk in run_keys]),
        final_eval_logits=m2.eval()(torch.from_numpy(x[:batch, None, :]).float()).detach().numpy(),
    )
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: k={k} loss1={float(loss):.4f} hist={np.round(tr.loss_train_history, 4).tolist()} "
          f"acc={np.round(tr.accuracy_train_history, 3).tolist()} size={os.path.getsize(os.path.join(HERE, name + '.npz')) / 1024:.0f} KB")


def big_ensemble_case(name="cfg0_l1024", n=64, length=1024, data_seed=0, seed=107, gain=2.0, members=None, refit=False):
    """BASELINE configs[0] at full size through the reference's own CamAL.predict (E=5, k=5,7,9,15,25, 64 x 1024
    UK-DALE-shaped windows, data seed 0): the windows come back from the seed, the states are stored as int8, the
    per-member CAMs and the averaged CAM at every 8th timestamp.  (members / refit: the configs[4] week windows,
    L = 10080 REFIT-shaped, on a few windows and two members.)"""
    KERNELS = members or globals()["KERNELS"]
    data = (synth.refit_windows if refit else synth.ukdale_windows)(n, length, seed=data_seed)
    gains, biases = calibrate_heads(data, seed, gain)
    gains = [gains[globals()["KERNELS"].index(k)] for k in KERNELS]
    biases = [biases[globals()["KERNELS"].index(k)] for k in KERNELS]
    sds = [weights.make_state_dict(k, seed=seed, head_gain=g, head_bias=b) for k, g, b in zip(KERNELS, gains, biases)]
    cams = np.zeros((len(KERNELS), n, length), np.float32)
    labels = np.zeros((len(KERNELS), n), np.int32)
    probs = np.zeros((len(KERNELS), n, 2), np.float32)
    for e, sd in enumerate(sds):
        m = ref_model(sd)
        cam = CAM(model=m, device="cpu", last_conv_layer=m._modules["layers"][2], fc_layer_name=m._modules["linear"])
        for i in range(n):
            c, lab, p = cam.run(instance=data[i], returned_cam_for_label=1)
            cams[e, i], labels[e, i], probs[e, i] = c, lab, p
    logs = {"ensemble_training_logs": {}, "camal_predictors_id": [], "is_fitted": True}
    for e, (k, sd) in enumerate(zip(KERNELS, sds)):
        logs["ensemble_training_logs"][f"Predictor_{e}"] = {"kernel_size": k, "best_model_state_dict": to_torch_sd(sd)}
        logs["camal_predictors_id"].append(f"Predictor_{e}")
    camal = CamAL(n_predictors=len(KERNELS), device="cpu")
    camal.camal_logs, camal.is_fitted = logs, True
    captured = {}
    orig_att = camal.attention_sigmoid_module

    def capture(d, soft_label, w_ma=5):
        captured["avg_cam"] = np.array(soft_label)
        return orig_att(d, soft_label, w_ma=w_ma)

    camal.attention_sigmoid_module = capture
    states, prob_detect = camal.predict(data, return_prob=True)
    assert set(np.unique(states)) <= {0.0, 1.0}
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), members=np.array(KERNELS), seed=np.array(seed), data_seed=np.array(data_seed),
        n=np.array(n), length=np.array(length), head_gain=np.array(gains, np.float64), head_bias=np.array(biases, np.float64),
        cam_stride=np.array(8), cams_strided=cams[:, :, ::8], cam_row_max=cams.max(axis=-1), labels=labels, probs=probs,
        avg_cam_strided=captured["avg_cam"][:, ::8].astype(np.float32), states=states.astype(np.int8), prob_detect=prob_detect)
    print(f"{name}: N={n} L={length} labels1={labels.mean():.2f} prob_detect in [{prob_detect.min():.3f},{prob_detect.max():.3f}] "
          f"states_on={float((states != 0).mean()):.3f} min|p-0.5|={np.abs(prob_detect - 0.5).min():.4f} "
          f"size={os.path.getsize(os.path.join(HERE, name + '.npz')) / 1024:.0f} KB")


def trainer_control_flow_case(name="trainer_control_flow"):
    """BasedClassifTrainer.train's epoch logic (torch_trainer.py:92-171: ReduceLROnPlateau, EarlyStopper after the
    warm-up, best-loss snapshot) of the REFERENCE on scripted epoch results: its two private epoch methods are
    replaced by the scripts of tests/helpers.scripted_trainer_scenarios, everything else is its own code."""
    from src.helpers.torch_trainer import BasedClassifTrainer
    from tests.helpers import scripted_trainer_scenarios
    out = {}
    for i, sc in enumerate(scripted_trainer_scenarios()):
        model = torch.nn.Linear(2, 2)
        tr = BasedClassifTrainer(model, train_loader=[], valid_loader=[0] if sc["has_valid"] else None,
                                 learning_rate=sc["lr"], weight_decay=0, patience_es=sc["patience_es"],
                                 patience_rlr=sc["patience_rlr"], device="cpu", n_warmup_epochs=sc["n_warmup_epochs"])
        lrs, calls = [], {"n": -1}

        def fake_train(sc=sc, calls=calls):  # train() runs it exactly once per epoch, first
            calls["n"] += 1
            return sc["train_loss"][calls["n"]], sc["train_acc"][calls["n"]]

        def fake_eval(sc=sc, calls=calls):
            return sc["valid_loss"][calls["n"]], sc["valid_acc"][calls["n"]]

        tr._BasedClassifTrainer__train = fake_train
        tr._BasedClassifTrainer__evaluate = fake_eval
        if tr.scheduler is not None:
            orig = tr.scheduler.step

            def step(v, orig=orig, tr=tr, lrs=lrs):
                orig(v)
                lrs.append(tr.optimizer.param_groups[0]["lr"])

            tr.scheduler.step = step
        tr.train(sc["n_epochs"])
        out[f"lrs_{i}"] = np.array(lrs, np.float64)
        out[f"epochs_run_{i}"] = np.array(len(tr.loss_train_history))
        out[f"passed_epochs_{i}"] = np.array(tr.passed_epochs)
        out[f"best_loss_{i}"] = np.array(float(tr.best_loss))
        out[f"epoch_best_{i}"] = np.array(int(tr.log.get("epoch_best_loss", -1)))
        out[f"value_best_{i}"] = np.array(float(tr.log.get("value_best_loss", np.nan)))
        out[f"valid_metrics_{i}"] = np.array(float(tr.log.get("valid_metrics", np.nan)))
        out[f"final_lr_{i}"] = np.array(float(tr.optimizer.param_groups[0]["lr"]))
        print(f"{name}[{i}]: epochs {sc['n_epochs']} -> ran {len(tr.loss_train_history)}, best epoch {out[f'epoch_best_{i}']}, "
              f"lr {sc['lr']:g} -> {float(out[f'final_lr_{i}']):g}, es={sc['patience_es']} rlr={sc['patience_rlr']} warm={sc['n_warmup_epochs']} valid={sc['has_valid']}")
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def trainer_valid_case(name="trainer_valid_l64", k=7, seed=204, n_train=48, n_valid=24, length=64, batch=16, epochs=14,
                       lr=5e-3, shuffle_seed=7):
    """The reference trainer WITH a validation loader (batch size 1, core.py:114-116), shuffle=True, ReduceLROnPlateau
    (patience 1), early stopping (patience 2) after one warm-up epoch: pins the LR sequence, the stop epoch, the best
    epoch and the validation histories."""
    from src.helpers.torch_dataset import TSDataset
    from src.helpers.torch_trainer import BasedClassifTrainer
    from tests.helpers import trainer_valid_inputs
    xt, yt, xv, yv = trainer_valid_inputs(seed, n_train, n_valid, length)
    sd0 = weights.make_state_dict(k, seed=seed, head_gain=1.0)
    m = CamALResNet(kernel_size=k)
    m.load_state_dict(to_torch_sd(sd0))
    torch.manual_seed(shuffle_seed)
    tl = torch.utils.data.DataLoader(TSDataset(xt, yt), batch_size=batch, shuffle=True)
    vl = torch.utils.data.DataLoader(TSDataset(xv, yv), batch_size=1, shuffle=False)
    tr = BasedClassifTrainer(m, train_loader=tl, valid_loader=vl, learning_rate=lr, weight_decay=0, patience_es=2,
                             patience_rlr=1, device="cpu", all_gpu=False, n_warmup_epochs=1)
    lrs = []
    orig = tr.scheduler.step

    def step(v):
        orig(v)
        lrs.append(tr.optimizer.param_groups[0]["lr"])

    tr.scheduler.step = step
    tr.train(epochs)
    test_loss, _ = tr.evaluate(vl)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), k=np.array(k), seed=np.array(seed), n_train=np.array(n_train), n_valid=np.array(n_valid),
        length=np.array(length), batch=np.array(batch), epochs=np.array(epochs), lr=np.array(lr), shuffle_seed=np.array(shuffle_seed),
        lrs=np.array(lrs), epochs_run=np.array(len(tr.loss_train_history)), passed_epochs=np.array(tr.passed_epochs),
        epoch_best_loss=np.array(tr.log["epoch_best_loss"]), value_best_loss=np.array(tr.log["value_best_loss"]),
        loss_train_history=np.array(tr.loss_train_history), loss_valid_history=np.array(tr.loss_valid_history),
        accuracy_train_history=np.array(tr.accuracy_train_history), accuracy_valid_history=np.array(tr.accuracy_valid_history),
        eval_loss=np.array(test_loss))
    print(f"{name}: ran {len(tr.loss_train_history)}/{epochs} epochs, lrs={lrs}, best epoch {tr.log['epoch_best_loss']} "
          f"({tr.log['value_best_loss']:.4f})\n  valid={np.round(tr.loss_valid_history, 4).tolist()}\n  train={np.round(tr.loss_train_history, 4).tolist()}")


def metrics_case(name="metrics_ref"):
    """NILMmetrics / Classifmetrics of the reference (other.py:86-159, sklearn underneath) on seeded arrays:
    powers with NaN / +-inf entries, states as post_kernel emits them for unsigned inputs (-0, 0, 1), detection scores
    with ties.  Inputs come from seeds, only the reference's numbers are stored."""
    from src.helpers.other import Classifmetrics, NILMmetrics
    loc_keys = ["MAE", "MSE", "RMSE", "TECA", "NDE", "SAE", "MR", "ACCURACY", "BALANCED_ACCURACY", "PRECISION",
                "RECALL", "F1_SCORE"]
    clf_keys = ["ACCURACY", "BALANCED_ACCURACY", "PRECISION", "RECALL", "F1_SCORE", "F1_SCORE_MACRO",
                "ROC_AUC_SCORE", "AP"]
    seeds = [301, 302, 303, 304]
    loc = np.zeros((len(seeds), len(loc_keys)))
    clf = np.zeros((len(seeds), len(clf_keys)))
    for i, seed in enumerate(seeds):
        y, y_hat, y_state, y_hat_state, y_clf, score = synth.metric_arrays(seed)
        m = NILMmetrics()(y=y, y_hat=y_hat, y_state=y_state, y_hat_state=y_hat_state)
        loc[i] = [float(m[k]) for k in loc_keys]
        c = Classifmetrics()(y_clf, score)
        clf[i] = [float(c[k]) for k in clf_keys]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), seeds=np.array(seeds), loc_keys=np.array(loc_keys),
                        clf_keys=np.array(clf_keys), loc=loc, clf=clf)
    print(f"{name}: loc[0]={loc[0].tolist()} clf[0]={clf[0].tolist()}")


def windows_case(name="windows_ref"):
    """The reference's own NILMscaler (torch_dataset.py:22-156) and nilmdataset_to_clfdataset
    (data_processing.py:17-33) on 4-D arrays that oracle/windows_oracle.py builds from seeded series (the
    builders' window loop itself needs the datasets' files and cannot run here)."""
    from oracle import windows_oracle
    from src.helpers.data_processing import nilmdataset_to_clfdataset
    from src.helpers.torch_dataset import NILMscaler
    cases = [(401, 510, 510, 1000), (402, 128, 64, "StandardScaling"), (403, 96, 96, "MaxScaling"), (404, 200, 50, "MeanMaxScaling")]
    out = dict(seeds=np.array([c[0] for c in cases]), W=np.array([c[1] for c in cases]), S=np.array([c[2] for c in cases]),
               kinds=np.array([str(c[3]) for c in cases]))
    for seed, W, S, kind in cases:
        stems = synth.house_series(6000, seed=seed)
        X, kept = windows_oracle.nilm_dataset(stems, W, S)
        sc = NILMscaler(power_scaling_type=kind, appliance_scaling_type="SameAsPower")  # None fails the reference's own assert (:38)
        Xs = sc.fit_transform(X.copy())
        Xa, y = nilmdataset_to_clfdataset(Xs)
        out[f"stat_{seed}"] = np.array([float(sc.power_stat1), float(sc.power_stat2)])
        out[f"kept_{seed}"] = kept
        out[f"y_{seed}"] = np.asarray(y, np.int8)
        out[f"x_{seed}"] = np.asarray(Xa[::3], np.float64)  # every third window, float64 as the reference holds it
        out[f"app_{seed}"] = np.asarray(Xs[::7, 1, 0, ::5], np.float64)  # scaled appliance channel, sampled
        back = sc.inverse_transform(Xs)
        out[f"inv_err_{seed}"] = np.array(float(np.abs(back[:, 0, 0, :] - X[:, 0, 0, :]).max()))
        print(f"{name}: seed={seed} W={W} S={S} {kind}: {len(X)} windows kept, labels on {float(np.mean(y)):.2f}")
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def resnet_variant_case(name, data, seeds, spread=2.0):
    """classifiers/resnet.py's ``ResNet`` (kernels 7/5/3, zero "same" padding) member by member: forward logits and
    ``CAM.run`` window by window (class_activation_map.py:34), on this repo's seeded weights.  ``CamAL.eval`` only
    ever builds ``CamALResNet`` (core.py:386), so the reference has no ensemble path for this class to capture: the
    ensemble body on top of these members is the one the ``ensemble_*`` fixtures pin."""
    from src.models.camal.classifiers.resnet import ResNet
    data32 = data.astype(np.float32)
    n, length = data.shape
    gains, biases, sds = [], [], []
    for seed in seeds:
        sd = weights.make_state_dict(7, seed=seed, arch=weights.ARCH_RESNET)
        m = ResNet()
        m.load_state_dict(to_torch_sd(sd))
        m.eval()
        with torch.no_grad():
            lg = m(torch.from_numpy(data32[:, None, :])).numpy()
        d = (lg[:, 1] - lg[:, 0]).astype(np.float64)
        g = float(np.float32(spread / max(d.std(), 1e-6)))
        b = (0.0, float(np.float32(-g * np.median(d))))
        gains.append(g)
        biases.append(b)
        sds.append(weights.make_state_dict(7, seed=seed, arch=weights.ARCH_RESNET, head_gain=g, head_bias=b))
    cams = np.zeros((len(seeds), n, length), np.float32)
    labels = np.zeros((len(seeds), n), np.int32)
    probs = np.zeros((len(seeds), n, 2), np.float32)
    logits = np.zeros((len(seeds), n, 2), np.float32)
    for e, sd in enumerate(sds):
        m = ResNet()
        m.load_state_dict(to_torch_sd(sd))
        m.eval()
        cam = CAM(model=m, device="cpu", last_conv_layer=m._modules["layers"][2], fc_layer_name=m._modules["linear"])
        for i in range(n):
            c, lab, p = cam.run(instance=data[i], returned_cam_for_label=1)
            cams[e, i], labels[e, i], probs[e, i] = c, lab, p
        with torch.no_grad():
            logits[e] = m(torch.from_numpy(data32[:, None, :])).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), seeds=np.array(seeds), head_gain=np.array(gains, np.float64),
                        head_bias=np.array(biases, np.float64), data=data, cams=cams, labels=labels, probs=probs,
                        logits=logits)
    print(f"{name}: N={n} L={length} seeds={seeds} labels1={labels.mean():.2f} "
          f"p1 in [{probs[..., 1].min():.3f},{probs[..., 1].max():.3f}]")


def width_case(name, mid, kernels, data, seed, spread=2.0):
    """CamALResNet(kernel_size, mid_channels=mid) (camal_resnet.py:5-19: what CamAL.train builds from its
    resnet_kwargs, core.py:36-38,106) member by member: forward logits and CAM.run rows from the reference class."""
    data32 = data.astype(np.float32)
    n, length = data.shape
    gains, biases, sds = [], [], []
    for k in kernels:
        sd = weights.make_state_dict(k, seed=seed, mid_channels=mid)
        m = CamALResNet(kernel_size=k, mid_channels=mid)
        m.load_state_dict(to_torch_sd(sd))
        m.eval()
        with torch.no_grad():
            lg = m(torch.from_numpy(data32[:, None, :])).numpy()
        d = (lg[:, 1] - lg[:, 0]).astype(np.float64)
        g = float(np.float32(spread / max(d.std(), 1e-6)))
        b = (0.0, float(np.float32(-g * np.median(d))))
        gains.append(g)
        biases.append(b)
        sds.append(weights.make_state_dict(k, seed=seed, mid_channels=mid, head_gain=g, head_bias=b))
    cams = np.zeros((len(kernels), n, length), np.float32)
    labels = np.zeros((len(kernels), n), np.int32)
    probs = np.zeros((len(kernels), n, 2), np.float32)
    logits = np.zeros((len(kernels), n, 2), np.float32)
    for e, (k, sd) in enumerate(zip(kernels, sds)):
        m = CamALResNet(kernel_size=k, mid_channels=mid)
        m.load_state_dict(to_torch_sd(sd))
        m.eval()
        cam = CAM(model=m, device="cpu", last_conv_layer=m._modules["layers"][2], fc_layer_name=m._modules["linear"])
        for i in range(n):
            c, lab, p = cam.run(instance=data[i], returned_cam_for_label=1)
            cams[e, i], labels[e, i], probs[e, i] = c, lab, p
        with torch.no_grad():
            logits[e] = m(torch.from_numpy(data32[:, None, :])).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), mid=np.array(mid), members=np.array(kernels), seed=np.array(seed),
                        head_gain=np.array(gains, np.float64), head_bias=np.array(biases, np.float64), data=data,
                        cams=cams, labels=labels, probs=probs, logits=logits)
    print(f"{name}: mid={mid} N={n} L={length} kernels={kernels} labels1={labels.mean():.2f} "
          f"p1 in [{probs[..., 1].min():.3f},{probs[..., 1].max():.3f}]")


def width_cases():
    width_case("width32_l150", 32, [5, 25], synth.ukdale_windows(6, 150, seed=23), seed=401)
    width_case("width128_l90", 128, [9], synth.ukdale_windows(4, 90, seed=24), seed=402)
    train_case("train_width32_k9_l72", k=9, seed=206, n=24, length=72, batch=12, epochs=2, mid=32)
    train_case("train_width128_k5_l40", k=5, seed=207, n=12, length=40, batch=6, epochs=1, mid=128)


def camtap_case(name="camtap_l120", k=9, seed=403, mid=64):
    """CAM with the hook on another layer (class_activation_map.py:25-31): last_conv_layer = the SECOND ResNetBlock of
    the reference model; rows for label 1 and for the predicted label (returned_cam_for_label=None, label_instance=0)."""
    data = synth.ukdale_windows(6, 120, seed=25)
    m = CamALResNet(kernel_size=k, mid_channels=mid)
    m.load_state_dict(to_torch_sd(weights.make_state_dict(k, seed=seed, mid_channels=mid)))
    m.eval()
    with torch.no_grad():
        lg = m(torch.from_numpy(data[:, None, :]).float()).numpy()
    d = (lg[:, 1] - lg[:, 0]).astype(np.float64)
    gain = float(np.float32(2.0 / max(d.std(), 1e-6)))
    bias = (0.0, float(np.float32(-gain * np.median(d))))
    sd = weights.make_state_dict(k, seed=seed, mid_channels=mid, head_gain=gain, head_bias=bias)
    m.load_state_dict(to_torch_sd(sd))
    cam = CAM(model=m, device="cpu", last_conv_layer=m._modules["layers"][1], fc_layer_name=m._modules["linear"])
    n, length = data.shape
    cams1 = np.zeros((n, length), np.float32)
    camsp = np.zeros((n, length), np.float32)
    labels = np.zeros(n, np.int32)
    probs = np.zeros((n, 2), np.float32)
    for i in range(n):
        c, lab, p = cam.run(instance=data[i], returned_cam_for_label=1)
        cams1[i], labels[i], probs[i] = c, lab, p
        c2, _, _ = cam.run(instance=data[i], label_instance=0)
        camsp[i] = c2
    np.savez_compressed(os.path.join(HERE, name + ".npz"), k=np.array(k), seed=np.array(seed), mid=np.array(mid), data=data,
                        head_gain=np.array(gain), head_bias=np.array(bias), cams_label1=cams1, cams_predicted=camsp, labels=labels, probs=probs)
    print(f"{name}: labels={labels.tolist()} p1={probs[:, 1].round(3).tolist()}")


def resnet_variant_cases():
    resnet_variant_case("resnetvar_l192", synth.ukdale_windows(8, 192, seed=21), seeds=[301, 302])
    # shorter than the 7-tap kernel's reach and not a tile multiple: every position touches the zero padding
    resnet_variant_case("resnetvar_l5", synth.ukdale_windows(3, 5, seed=22), seeds=[303])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "resnetvar":
        return resnet_variant_cases()
    if len(sys.argv) > 1 and sys.argv[1] == "trainvar":
        return train_case("train_resnetvar_l80", k=7, seed=203, n=36, length=80, batch=12, epochs=2, arch=weights.ARCH_RESNET)
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":
        return metrics_case()
    if len(sys.argv) > 1 and sys.argv[1] == "windows":
        return windows_case()
    if len(sys.argv) > 1 and sys.argv[1] == "camtap":
        return camtap_case()
    if len(sys.argv) > 1 and sys.argv[1] == "widths":
        return width_cases()
    if len(sys.argv) > 1 and sys.argv[1] == "cfg0":
        return big_ensemble_case()
    if len(sys.argv) > 1 and sys.argv[1] == "week":
        return big_ensemble_case("week_l10080", n=6, length=10080, data_seed=4, seed=131, gain=2.0, members=[5, 25], refit=True)
    if len(sys.argv) > 1 and sys.argv[1] == "control":
        return trainer_control_flow_case()
    if len(sys.argv) > 1 and sys.argv[1] == "trainvalid":
        return trainer_valid_case()
    if len(sys.argv) > 1 and sys.argv[1] == "train510":
        return train_case("train_k7_l510", k=7, seed=205, n=128, length=510, batch=128, epochs=1)
    width_cases()
    camtap_case()
    big_ensemble_case()
    big_ensemble_case("week_l10080", n=6, length=10080, data_seed=4, seed=131, gain=2.0, members=[5, 25], refit=True)
    trainer_control_flow_case()
    trainer_valid_case()
    train_case("train_k7_l510", k=7, seed=205, n=128, length=510, batch=128, epochs=1)
    windows_case()
    metrics_case()
    resnet_variant_cases()
    train_case("train_k7_l96", k=7, seed=201, n=48, length=96, batch=16, epochs=2)
    train_case("train_k25_l70", k=25, seed=202, n=24, length=70, batch=12, epochs=1)
    train_case("train_resnetvar_l80", k=7, seed=203, n=36, length=80, batch=12, epochs=2, arch=weights.ARCH_RESNET)
    # 1. the BASELINE config-1 shape, shrunk: full ensemble, float32 data
    run_case("ensemble_l256", synth.ukdale_windows(12, 256, seed=11), seed=101, gain=2.0)
    # 2. float64 data (output dtype follows the input, quirk 9), ragged length (not a tile multiple)
    run_case("ensemble_l333_f64", synth.ukdale_windows(6, 333, seed=12).astype(np.float64), seed=102, gain=2.0)
    # 3. signed (shifted) input -> negative sigmoid arguments, states in {-1,-0,0,1} (quirk 8)
    x = synth.ukdale_windows(6, 200, seed=13).astype(np.float64)
    x = x - 2.0  # base load ~ -1.85: ma*x < -ln3 wherever the averaged CAM is high off-pulse
    run_case("ensemble_l200_signed", x, seed=103, gain=2.0)
    # 4. even moving-average window (asymmetric taps) + single window + 2 members
    run_case("pair_l128_wma4", synth.ukdale_windows(4, 128, seed=14), seed=104, gain=2.0, w_ma=4, members=[7, 25])
    # 5. the gated (device != cpu) variant of predict, quirk 7
    run_case("ensemble_l160_gated", synth.ukdale_windows(10, 160, seed=15), seed=105, gain=2.0, gated=True)
    # 6. all-zero input and constant input windows (cam.max()==0 style degenerate cases), short L
    z = np.zeros((4, 64), np.float32)
    z[1] = 0.5
    z[2, 10:20] = 3.0
    z[3] = synth.ukdale_windows(1, 64, seed=16)[0]
    run_case("ensemble_l64_degenerate", z, seed=106, gain=2.0)


if __name__ == "__main__":
    main()
