"""``CamAL`` — drop-in mirror of the reference's public class (``src/models/camal/core.py:19``).

Same constructor, ``train / predict / __call__ / test / load / eval / to`` signatures, same
``camal_logs`` dictionary schema and the same exceptions; everything from "ResNet forward"
downward (and the per-window Python loops of ``predict``) runs in libcamal_hip.so on MI355X.

Deliberate normalisations of reference quirks:
  * ``device`` must be a GPU ("cuda", "cuda:N"); the default ``"cpu"`` of the reference has no
    meaning here and raises.  ``predict`` follows the reference's *CPU-path* semantics (no
    member-0 pre-filter) unless ``self.prefilter = True`` (its ``device != 'cpu'`` path).
  * ``test`` returns ``(localization_metrics, classification_metrics)``: the reference builds
    a set literal of dicts (core.py:235-240), which raises ``TypeError``; its own scripts
    unpack two values (scripts/ukdale.py:144).
  * ``test`` without a scaler treats the data as unscaled (the reference hits a ``NameError``
    after printing its warning, core.py:205-212).
"""
from __future__ import annotations

import heapq
import pickle
import time
import warnings

import numpy as np
import torch

from .cam import CAM
from .engine import Ensemble
from .metrics import Classifmetrics, NILMmetrics
from .resnet import CamALResNet


def nilmdataset_to_clfdataset(data):
    """Weak window labels of a 4-D NILM array (reference data_processing.py:17-33)."""
    labels = [1 if data[i, 1, 1, :].any() > 0 else 0 for i in range(len(data))]
    return data[:, 0, 0, :], np.array(labels)


def _undersampling_targets(y, sampling_strategy):
    """{class: number of samples to keep} of the classes that get under-sampled, as imbalanced-learn resolves
    ``sampling_strategy`` for an under-sampler (imblearn/utils/_validation.py, declared ``imbalanced-learn>=0.13.0``
    in the reference's pyproject.toml; the package is not installed here, so this restates its documented rules):
    'majority', 'not minority' (= 'auto'), 'not majority', 'all' -> the minority class's count for the classes
    named; a float r (binary only) -> the majority keeps int(n_minority / r); a dict -> its own counts; a callable ->
    its dict."""
    classes, counts = np.unique(y, return_counts=True)
    stats = {int(c): int(n) for c, n in zip(classes, counts)}
    first_seen = list(dict.fromkeys(int(v) for v in y))  # Counter(y)'s key order: ties go to the first class seen
    n_min = min(stats.values())
    n_max = max(stats.values())
    c_min = min(first_seen, key=lambda c: stats[c])
    c_maj = max(first_seen, key=lambda c: stats[c])
    if callable(sampling_strategy):
        sampling_strategy = sampling_strategy(y)
    if isinstance(sampling_strategy, dict):
        for c, n in sampling_strategy.items():
            if c not in stats:
                raise ValueError(f"The {c} target class is not present in the data.")
            if n > stats[c]:
                raise ValueError("With under-sampling methods, the number of samples in a class should be less or equal "
                                 f"to the original number of samples. Originally, there is {stats[c]} samples and {n} samples are asked.")
        return {int(c): int(n) for c, n in sampling_strategy.items()}
    if isinstance(sampling_strategy, (int, float)) and not isinstance(sampling_strategy, bool):
        if len(stats) != 2:
            raise ValueError('"sampling_strategy" can be a float only when the type of target is binary.')
        r = float(sampling_strategy)
        if not 0 < r <= 1:
            raise ValueError(f"When 'sampling_strategy' is a float, it should be in the range (0, 1]. Got {r} instead.")
        out = {c: int(n_min / r) for c in stats if c != c_min}
        if any(n > stats[c] for c, n in out.items()):
            raise ValueError("The specified ratio required to generate new sample in the majority class while trying to "
                             "remove samples. Please increase the ratio.")
        return out
    if sampling_strategy in ("auto", "not minority"):
        return {c: n_min for c in stats if c != c_min}
    if sampling_strategy == "majority":
        return {c_maj: n_min}
    if sampling_strategy == "not majority":
        return {c: n_min for c in stats if c != c_maj}
    if sampling_strategy == "all":
        return {c: n_min for c in stats}
    raise ValueError(f"When 'sampling_strategy' is a string, it needs to be one of 'majority', 'not minority', "
                     f"'not majority', 'all', 'auto'. Got '{sampling_strategy}' instead.")


def undersampler(X, y, sampling_strategy="auto", seed=0):
    """Random under-sampling, then a seeded shuffle (reference data_processing.py:37-57: imblearn's
    ``RandomUnderSampler(sampling_strategy, random_state=seed).fit_resample`` + ``np.random.shuffle``).  imblearn is
    not available here: this restates RandomUnderSampler (one ``RandomState(seed)``; classes in sorted order; a class
    that is under-sampled keeps ``choice(range(n), size=target, replace=False)`` of its samples in that order, the
    others keep all of theirs) and every ``sampling_strategy`` form it documents.  Index-for-index parity with the
    installed package is unpinned (no copy of it to compare with)."""
    X = np.asarray(X)
    y = np.asarray(y).astype(int).ravel()
    targets = _undersampling_targets(y, sampling_strategy)
    rs = np.random.RandomState(seed)
    idx = []
    for c in np.unique(y):
        members = np.flatnonzero(y == c)
        if int(c) in targets:
            members = members[rs.choice(range(len(members)), size=targets[int(c)], replace=False)]
        idx.append(members)
    idx = np.concatenate(idx)
    mat = np.concatenate((X[idx], y[idx].reshape(-1, 1)), axis=1)
    np.random.seed(seed)
    np.random.shuffle(mat)
    mat = mat.astype(np.float32)
    return mat[:, :-1], mat[:, -1]


def _scaler_affine(scaler):
    """(mul, add) of NILMscaler.inverse_transform for the aggregate channel
    (reference torch_dataset.py:141-144)."""
    if scaler is None:
        return 1.0, 0.0
    s1, s2 = float(scaler.power_stat1), float(scaler.power_stat2)
    if getattr(scaler, "power_scaling_type", None) == "MinMax":
        return s2 - s1, s1
    return s2, s1


def _appliance_affine(scaler):
    """(mul, add) of NILMscaler.inverse_transform for the first appliance's power channel (torch_dataset.py:146-153 /
    camal_amd.windows.NILMscaler): (1, 0) without a scaler or without appliance scaling; None for a scaler object whose
    statistics cannot be read (the caller's own class: its inverse_transform is called on the whole array instead)."""
    if scaler is None or getattr(scaler, "appliance_scaling_type", "?") is None:
        return 1.0, 0.0
    try:
        a1, a2 = float(scaler.appliance_stat1[0]), float(scaler.appliance_stat2[0])
        kind, pkind = scaler.appliance_scaling_type, getattr(scaler, "power_scaling_type", None)
    except (AttributeError, IndexError, TypeError, ValueError):
        return None
    minmax = kind == "MinMax" or (kind == "SameAsPower" and pkind == "MinMax")
    return ((a2 - a1) if minmax else a2), a1


class CamAL(object):
    def __init__(self, n_predictors=5, device="cuda", batch_inference=128,
                 loc_metrics=NILMmetrics(), clf_metrics=Classifmetrics(), **resnet_kwargs):
        self.device = device
        self.n_predictors = n_predictors
        self.batch_inference = batch_inference
        self.loc_metrics = loc_metrics
        self.clf_metrics = clf_metrics

        resnet_kwargs.setdefault("in_channels", 1)
        resnet_kwargs.setdefault("mid_channels", 64)
        resnet_kwargs.setdefault("norm", "BatchNorm")
        self.resnet_kwargs = resnet_kwargs

        self.is_fitted = False
        self.camal_logs = None
        self.list_resnet_predictors = None
        self.list_cam_predictors = None

        # knobs that do not exist in the reference
        self.precision = "fp32"   # "bf16": bf16 MFMA convs with fp32 accumulation
        self.prefilter = False    # True: reference device!='cpu' gating (core.py:292-300)
        self.stream_windows = None  # windows per chunk of predict's host pipeline (None: about 6 M timesteps)
        # True under torch.distributed with world > 1: train() spreads the (kernel size, try) candidates over the
        # ranks (they are independent, core.py:104-158), each trained whole on one GPU with no gradient exchange;
        # the logs are gathered once at the end.  False: every rank trains every candidate data-parallel.
        self.member_parallel = False
        # > 1: this rank's candidates train side by side on one GPU, that many at a time, each in its own host thread
        # and HIP stream with its own shuffle generator (one member's short kernels fill the gaps of another's).
        # Not with data-parallel training over several ranks (their collectives must be issued in one order).
        self.member_streams = 1
        # Several GPUs from this ONE process (what all_gpu=True is for in the reference, torch_trainer.py:82-90):
        # ``devices`` = ["cuda:0", "cuda:1", ...], or ``all_devices = True`` for every visible GPU.
        # predict(): every GPU holds all the members and takes a contiguous block of the windows on a host thread of its
        # own; results land in disjoint slices of the output arrays; no collective.
        # train(): the (kernel size, try) candidates go round robin over the GPUs, each trained whole on one GPU by a
        # host thread (they are independent, core.py:104-158): no gradient exchange.  None / False: ``device`` alone.
        self.devices = None
        self.all_devices = False
        # dtypes of predict()'s arrays; the reference's are float64 (core.py:449 convolves against float64 ones).
        # np.int8 / np.float32 keep 10 M-window runs (configs[2]) at 5 instead of 16 bytes per timestamp of host memory.
        self.states_dtype = None
        self.power_dtype = np.float64
        self._ensemble = None
        self._ensembles = []

    # ------------------------------------------------------------------ train (core.py:47-189)
    def train(self, train_dataset: tuple, valid_dataset: tuple, test_dataset: tuple,
              list_kernel_sizes: list = [5, 7, 9, 15, 25], n_try_by_kernel: int = 1,
              path: str = None, **training_kwargs):
        assert n_try_by_kernel * len(list_kernel_sizes) >= self.n_predictors, (
            "The number of keep predictors does not meet the number of kernel to try and the try by kernel.")
        from .trainer import BasedClassifTrainer, TSLoader

        training_kwargs.setdefault("batch_size", 128)
        training_kwargs.setdefault("epochs", 50)
        training_kwargs.setdefault("lr", 1e-3)
        training_kwargs.setdefault("weight_decay", 0)
        training_kwargs.setdefault("patience_es", 5)
        training_kwargs.setdefault("patience_rlr", 3)
        training_kwargs.setdefault("n_warmup_epochs", 1)
        training_kwargs.setdefault("all_gpu", False)

        X_train, y_train = train_dataset[0], train_dataset[1]
        X_valid, y_valid = valid_dataset[0], valid_dataset[1]
        X_test, y_test = test_dataset[0], test_dataset[1]
        X_train, y_train = undersampler(X_train, y_train, sampling_strategy="auto", seed=0)

        tmp_loss_results = {}
        camal_logs = {"ensemble_training_logs": {}}
        start_training_time = time.time()

        from . import parallel
        rank, _, world = parallel.world_info()
        spread = bool(self.member_parallel) and world > 1
        local_results = {}
        train_devices = self._device_list()
        if train_devices is None and training_kwargs["all_gpu"] and world == 1 and torch.cuda.device_count() > 1:
            # the reference's all_gpu=True (nn.DataParallel over every GPU, torch_trainer.py:82-90) from a single
            # process: every GPU is used, by giving each a candidate of its own instead of a slice of every batch
            train_devices = [f"cuda:{i}" for i in range(torch.cuda.device_count())]
        over_devices = world == 1 and train_devices is not None and len(train_devices) > 1

        def run_candidate(idx, kernel_size, nth_try, generator=None, resnet_inst=None, device=None):
            device = self.device if device is None else device
            if resnet_inst is None:
                resnet_inst = CamALResNet(kernel_size=kernel_size, **self.resnet_kwargs)
            train_loader = TSLoader(X_train, y_train, batch_size=training_kwargs["batch_size"], shuffle=True, generator=generator)
            valid_loader = TSLoader(X_valid, y_valid, batch_size=1, shuffle=False, generator=generator)
            test_loader = TSLoader(X_test, y_test, batch_size=1, shuffle=False, generator=generator)
            clf_trainer = BasedClassifTrainer(
                resnet_inst, train_loader=train_loader, valid_loader=valid_loader,
                learning_rate=training_kwargs["lr"], weight_decay=training_kwargs["weight_decay"],
                patience_es=training_kwargs["patience_es"], patience_rlr=training_kwargs["patience_rlr"],
                n_warmup_epochs=training_kwargs["n_warmup_epochs"], device=device,
                all_gpu=training_kwargs["all_gpu"] and not over_devices, data_parallel=not spread and not over_devices)
            clf_trainer.train(training_kwargs["epochs"])
            clf_trainer.restore_best_weights()
            loss_eval, _ = clf_trainer.evaluate(test_loader)
            print(f"Trained predictor {idx} with kernel size: {kernel_size} (nth_try {nth_try}) - loss: {loss_eval}.")
            clf_trainer.log["kernel_size"] = kernel_size
            clf_trainer.log["nth_try"] = nth_try
            if getattr(clf_trainer, "engine", None) is not None:
                clf_trainer.engine.close()  # the candidate's training workspace goes back before the next one starts
            return loss_eval, clf_trainer.log

        jobs, idx_clf = [], 0
        for kernel_size in list_kernel_sizes:
            for nth_try in range(n_try_by_kernel):
                if not (spread and parallel.member_owner(idx_clf, world) != rank):
                    jobs.append((idx_clf, kernel_size, nth_try))
                idx_clf += 1

        n_side = int(self.member_streams or 1)
        if (n_side > 1 or over_devices) and (world == 1 or spread) and len(jobs) > 1:
            import torch
            from concurrent.futures import ThreadPoolExecutor
            # one process, several GPUs (self.devices / all_devices): the candidates go round robin over the GPUs, each
            # trained whole on its GPU by a host thread, member_streams of them side by side per GPU; no gradient exchange
            # (the candidates are independent, core.py:104-158) — what all_gpu=True is for in the reference, without
            # splitting any batch
            devs = [torch.device(d) for d in train_devices] if over_devices else [torch.device(self.device)]
            base = int(torch.empty((), dtype=torch.int64).random_().item())  # one draw: reproducible under manual_seed
            # initial weights come from the global RNG: drawn here, in candidate order, before the threads start
            models = {job[0]: CamALResNet(kernel_size=job[1], **self.resnet_kwargs) for job in jobs}
            slot = {job[0]: i for i, job in enumerate(jobs)}

            def threaded(job):
                idx = job[0]
                dev = devs[slot[idx] % len(devs)]
                gen = torch.Generator().manual_seed((base + 7919 * idx) % (1 << 62))
                with torch.cuda.device(dev):
                    stream = torch.cuda.Stream(dev)
                    with torch.cuda.stream(stream):
                        out = run_candidate(*job, generator=gen, resnet_inst=models[idx], device=str(dev))
                    stream.synchronize()
                return idx, out

            with ThreadPoolExecutor(max_workers=n_side * len(devs)) as pool:
                for idx, out in pool.map(threaded, jobs):
                    local_results[f"Predictor_{idx}"] = out
        else:
            for job in jobs:
                local_results[f"Predictor_{job[0]}"] = run_candidate(*job)

        if spread:
            local_results = parallel.merge_member_logs(local_results)
        for i in range(idx_clf):  # the reference's insertion order, whichever rank trained the candidate
            loss_eval, log = local_results[f"Predictor_{i}"]
            tmp_loss_results[f"Predictor_{i}"] = loss_eval
            camal_logs["ensemble_training_logs"][f"Predictor_{i}"] = log

        camal_logs["training_time"] = round((time.time() - start_training_time), 3)

        heap = [(loss, name) for name, loss in tmp_loss_results.items()]
        heapq.heapify(heap)
        smallest = heapq.nsmallest(self.n_predictors, heap)
        list_best_clf = [name for _, name in smallest]
        camal_logs["camal_predictors_id"] = list_best_clf
        for i in range(idx_clf):
            if f"Predictor_{i}" not in list_best_clf:
                camal_logs["ensemble_training_logs"][f"Predictor_{i}"].pop("best_model_state_dict", None)
        camal_logs["is_fitted"] = True

        if path is not None:
            with open(path, "wb") as f:
                pickle.dump(camal_logs, f, protocol=pickle.HIGHEST_PROTOCOL)

        self.camal_logs = camal_logs
        self.is_fitted = True
        self.eval()
        return

    # ------------------------------------------------------------------ test (core.py:191-240)
    def test(self, data_test, appliance_mean_on_power=None, scaler=None):
        if len(data_test.shape) < 4:
            raise ValueError("Provided test data need to be given in NILM standard (4D numpy array)")

        data = data_test[:, 0, 0, :]
        want_power = appliance_mean_on_power is not None
        mul, add = 1.0, 0.0
        if want_power:
            if scaler is None:
                warnings.warn("No scaler provided for inverse_transform, are you sure you didn't scale the data for training?")
            mul, add = _scaler_affine(scaler)

        if isinstance(self.loc_metrics, NILMmetrics):
            # localization metrics without the [N, L] results ever leaving the device, at any N: windows, true powers
            # and true states stream through the GPU in chunks, twelve sums per chunk come back (Ensemble.test_sums_host)
            self._check_fitted_and_eval()
            y_scaled, y_mul, y_add = None, 1.0, 0.0
            if want_power:
                aff = _appliance_affine(scaler)
                if aff is None:  # a scaler object of the caller's own: its inverse_transform on the whole array
                    y_scaled = scaler.inverse_transform(data_test)[:, 1, 0, :]
                else:
                    y_scaled, (y_mul, y_add) = data_test[:, 1, 0, :], aff
            sums, n, prob = self._ensemble.test_sums_host(
                data, y=y_scaled, y_state=data_test[:, 1, 1, :], gate_mode=int(bool(self.prefilter)),
                p_on=float(appliance_mean_on_power) if want_power else None, scale_mul=mul, scale_add=add, y_mul=y_mul,
                y_add=y_add, chunk_windows=self.stream_windows)
            prob_detect = prob.reshape(-1, 1)
            metric_softlabel = self.loc_metrics.from_sums(sums, n, regression=want_power)
        else:  # a caller-supplied metric object gets host arrays, like the reference's
            if want_power:
                res = self._localize_device(data, p_on=float(appliance_mean_on_power), scale_mul=mul, scale_add=add)
            else:
                res = self._localize_device(data)
            prob_detect = res["prob"].cpu().numpy().reshape(-1, 1)
            y_state = data_test[:, 1, 1, :].ravel().astype(dtype=int)
            y = None
            if want_power:
                rescaled = scaler.inverse_transform(data_test) if scaler is not None else data_test
                y = rescaled[:, 1, 0, :].ravel()
            y_hat_state = res["state"].cpu().numpy().ravel().astype(dtype=int)
            y_hat = res["power"].cpu().numpy().astype(np.float64).ravel() if want_power else None
            metric_softlabel = self.loc_metrics(y, y_hat, y_state, y_hat_state) if want_power else \
                self.loc_metrics(y=None, y_hat=None, y_state=y_state, y_hat_state=y_hat_state)

        _, y_clf_true = nilmdataset_to_clfdataset(data_test)
        metric_classif = self.clf_metrics(y=y_clf_true.ravel(), y_hat=prob_detect.ravel())
        return metric_softlabel, metric_classif

    # ------------------------------------------------------------------ load (core.py:242-250)
    def load(self, path: str):
        with open(path, "rb") as handle:
            self.camal_logs = pickle.load(handle)
        print("Log successfully loaded.")
        self.is_fitted = self.camal_logs["is_fitted"]
        self.list_resnet_predictors = None
        return

    # --------------------------------------------------------------- predict (core.py:252-332)
    def predict(self, data: np.array, y: np.array = None, return_prob: bool = False, w_ma: int = 5):
        """Per-timestamp on/off localization of the appliance for windows ``data`` [N,L] or [L].
        Returns states [N,L] float64 in {-1,-0,0,1} (and ``prob_detect`` [N,1] float64)."""
        return self._predict_impl(data, y=y, return_prob=return_prob, w_ma=w_ma)

    def _predict_impl(self, data, y=None, return_prob=False, w_ma=5, p_on=None, scale_mul=1.0, scale_add=0.0):
        if not self.is_fitted:
            raise ValueError(
                "CamAL instance is not fitted. Train CamAL or load a fitted instance of CamAL before trying to predict.")
        data = np.asarray(data)
        assert len(data.shape) <= 2, (
            f"Input data need to be a 1D or 2D (batched) numpy array, got a {len(data.shape)}D array of size {data.shape}.")
        if len(data.shape) == 1:
            data = np.expand_dims(data, axis=0)
        if self.list_resnet_predictors is None:
            self.eval()

        # NumPy in, NumPy out (core.py:329-332): chunks through pinned buffers, copies under the kernels.
        # The reference returns float64 states whatever the input (np.convolve against float64 ones, core.py:449)
        sdt = self.states_dtype or np.float64
        kw = dict(y=y, w_ma=w_ma, gate_mode=int(bool(self.prefilter)), p_on=p_on, scale_mul=scale_mul, scale_add=scale_add,
                  chunk_windows=self.stream_windows, states_dtype=sdt, power_dtype=self.power_dtype)
        if len(self._ensembles) > 1 and len(data) >= len(self._ensembles):
            from .engine import localize_host_sharded
            prediction, prob, power = localize_host_sharded(self._ensembles, data, **kw)
        else:
            prediction, prob, power = self._ensemble.localize_host(data, **kw)
        out = [prediction]
        if return_prob:
            out.append(prob.reshape(-1, 1))
        if p_on is not None:
            out.append(power)
        return out[0] if len(out) == 1 else tuple(out)

    def _check_fitted_and_eval(self):
        if not self.is_fitted:
            raise ValueError(
                "CamAL instance is not fitted. Train CamAL or load a fitted instance of CamAL before trying to predict.")
        if self.list_resnet_predictors is None:
            self.eval()

    def _localize_device(self, data, y=None, w_ma=5, p_on=None, scale_mul=1.0, scale_add=0.0):
        """predict's checks + the fused HIP path; results stay on the device (dict of torch tensors)."""
        if not self.is_fitted:
            raise ValueError(
                "CamAL instance is not fitted. Train CamAL or load a fitted instance of CamAL before trying to predict.")
        data = np.asarray(data)
        assert len(data.shape) <= 2, (
            f"Input data need to be a 1D or 2D (batched) numpy array, got a {len(data.shape)}D array of size {data.shape}.")
        if len(data.shape) == 1:
            data = np.expand_dims(data, axis=0)
        if self.list_resnet_predictors is None:
            self.eval()
        is64 = data.dtype != np.float32
        return self._ensemble.localize(
            np.ascontiguousarray(data, dtype=np.float32),
            x64=np.ascontiguousarray(data, dtype=np.float64) if is64 else None,
            y=y, w_ma=w_ma, gate_mode=int(bool(self.prefilter)), pred_f64=is64,
            p_on=p_on, scale_mul=scale_mul, scale_add=scale_add)

    def __call__(self, data: np.array, y: np.array = None, return_prob: bool = False):
        return self.predict(data, y=y, return_prob=return_prob)

    # -------------------------------------------------------------------- to (core.py:346-364)
    def to(self, device: str) -> None:
        if self.list_resnet_predictors is not None:
            for predictors in self.list_resnet_predictors:
                predictors.to(device)
            self.device = device
            self.eval()
        else:
            warnings.warn("Cannot move anything to a device: no predictors in cache.")
            self.device = device
        return

    def _device_list(self):
        """``devices`` as given, every visible GPU with ``all_devices``, else None (``device`` alone)."""
        if self.devices:
            return [str(d) for d in self.devices]
        if self.all_devices:
            return [f"cuda:{i}" for i in range(torch.cuda.device_count())]
        return None

    # ------------------------------------------------------------------ eval (core.py:366-413)
    def eval(self) -> None:
        if not self.is_fitted:
            raise ValueError(
                "CamAL instance is not fitted. Train CamAL or load a fitted instance of CamAL before trying to move it in eval mode.")
        list_resnet_predictors, list_cam_predictors, sds = [], [], []
        for resnet_name in self.camal_logs["camal_predictors_id"]:
            log = self.camal_logs["ensemble_training_logs"][resnet_name]
            # The reference rebuilds DEFAULT models here (core.py:386-390), so an ensemble trained with
            # CamAL(mid_channels=32 / 128) fails in its own load_state_dict; the width is read from the checkpoint instead
            # (64 gives the reference's model unchanged).
            sd_best = log["best_model_state_dict"]
            mid = int(np.asarray(sd_best["layers.0.layers.0.layers.0.weight"].cpu() if torch.is_tensor(
                sd_best["layers.0.layers.0.layers.0.weight"]) else sd_best["layers.0.layers.0.layers.0.weight"]).shape[0])
            resnet_inst = CamALResNet(kernel_size=log["kernel_size"], mid_channels=mid)
            resnet_inst.to(self.device)
            resnet_inst.load_state_dict(log["best_model_state_dict"])
            resnet_inst.eval()
            cam_resnet_inst = CAM(model=resnet_inst, device=self.device,
                                  last_conv_layer=resnet_inst._modules["layers"][2],
                                  fc_layer_name=resnet_inst._modules["linear"])
            list_resnet_predictors.append(resnet_inst)
            list_cam_predictors.append(cam_resnet_inst)
            sds.append(resnet_inst._params)
        self.list_resnet_predictors = list_resnet_predictors
        self.list_cam_predictors = list_cam_predictors
        for ens in self._ensembles or ([self._ensemble] if self._ensemble is not None else []):
            ens.close()
        devices = self._device_list()
        if devices:
            # the first entry is the instance's own device: CAM.run, test() and the mirror classes live there
            self._ensembles = [Ensemble(sds, device=d, precision=self.precision) for d in devices]
            self._ensemble = self._ensembles[0]
        else:
            self._ensemble = Ensemble(sds, device=self.device, precision=self.precision)
            self._ensembles = [self._ensemble]
        return

    # -------------------------------------------- helpers kept for API parity (core.py:415-482)
    def attention_sigmoid_module(self, data: np.array, soft_label: np.array, w_ma: int = 5) -> np.array:
        """Moving average + rectified sigmoid + round of an averaged CAM (core.py:415-437), on the
        device (``camal_attention``).  ``predict`` does not call this: its fused kernel includes it."""
        from . import _lib
        lib = _lib.load()
        data = np.atleast_2d(np.asarray(data, dtype=np.float64))
        soft = np.atleast_2d(np.asarray(soft_label, dtype=np.float64))
        B, L = data.shape
        dev = torch.device(self.device)
        with torch.cuda.device(dev):
            s_d = torch.from_numpy(np.ascontiguousarray(soft)).to(dev)
            x_d = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
            out = torch.empty_like(x_d)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.camal_attention(_lib.ptr(s_d), _lib.ptr(x_d), B, L, int(w_ma), _lib.ptr(out), stream))
        return out.cpu().numpy()

    def moving_average(self, x: np.array, w: int) -> np.array:
        """One row's moving average with zero edges (core.py:439-449), on the host: kept for callers of the reference's
        public helper; predict's fused kernel has its own."""
        return np.convolve(x, np.ones(w), "same") / w

    def sigmoid(self, z):
        """The rectified sigmoid 2 s(z) - 1 (core.py:451-461), on the host."""
        return 2 * (1.0 / (1.0 + np.exp(-z))) - 1

    def _predict_subsequence_class(self, model, data) -> np.array:
        """Batched argmax of one member's logits (core.py:463-482)."""
        model.to(self.device)
        model.eval()
        logits = model(np.asarray(data, dtype=np.float32))
        return torch.max(logits, 1)[1].cpu().numpy().flatten().astype(np.int8)
