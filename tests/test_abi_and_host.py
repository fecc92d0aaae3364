"""CPU tests: the C-ABI library loads and exports every symbol include/camal_hip.h declares
(no compute calls), and the host-side mirror of the reference API behaves like the reference
where no GPU is needed (schema, errors, folding, metrics, sampling)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "camal_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(camal_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from camal_amd import _lib
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"libcamal_hip.so lacks {s}"
        assert s in _lib.SIGNATURES, f"ctypes binding lacks {s}"
    assert set(_lib.SIGNATURES) == set(syms)
    assert b"gfx950" in lib.camal_version()


def test_blob_size_matches_parameter_count():
    from camal_amd import _lib, weights
    from camal_amd.engine import folded_blob
    lib = _lib.load()
    for k in (5, 7, 9, 15, 25):
        sd = weights.make_state_dict(k, seed=0)
        blob = folded_blob(sd)
        # folding removes the 4 BN vectors per conv and keeps conv w + b: same count as w+b+head
        assert blob.size == lib.camal_member_blob_floats(k)
        # P(k) = M(k) + 3714 trainable parameters, M(k) = 24640 k + 376896
        assert weights.count_parameters(k) == 24640 * k + 376896 + 3714


def test_argument_errors_without_gpu():
    from camal_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    ks = (ctypes.c_int * 1)(8)  # even kernel size
    assert lib.camal_create(0, 1, ks, 0, ctypes.byref(h)) == -1
    assert b"odd" in lib.camal_last_error()
    ks = (ctypes.c_int * 1)(7)
    assert lib.camal_create(0, 1, ks, 7, ctypes.byref(h)) == -1  # bad precision
    assert lib.camal_create(0, 0, ks, 0, ctypes.byref(h)) == -1  # no members
    assert lib.camal_nilm_sums(None, None, None, None, None, 10, None, None, 0, None) == -1
    assert lib.camal_nilm_workspace_bytes(1) == 12 * 8 and lib.camal_nilm_workspace_bytes(1 << 40) == 2048 * 12 * 8
    assert lib.camal_postprocess(1, None, None, None, None, None, 1, 1, 5, 0, 0, 0.0, 1.0, 0.0, None, None, None, None, None) == -1


def test_state_dict_schema_is_the_reference_one():
    from camal_amd import weights
    spec = weights.state_dict_spec(7)
    assert len(spec) == 79
    assert spec["layers.1.layers.0.layers.0.weight"][0] == (128, 64, 7)
    assert spec["layers.0.residual.0.weight"][0] == (64, 1, 1)
    assert spec["layers.1.residual.1.running_var"][0] == (128,)
    assert "layers.2.residual.0.weight" not in spec  # block 3 has no skip (quirk 2)
    assert spec["linear.weight"][0] == (2, 128)
    assert [k for _, _, _, k in weights.conv_layers(25)] == [25, 7, 3, 25, 7, 3, 25, 7, 3]


def test_resnet_mirror_state_dict_roundtrip():
    from camal_amd import weights
    from camal_amd.resnet import CamALResNet
    m = CamALResNet(kernel_size=9)
    sd = m.state_dict()
    assert list(sd.keys()) == list(weights.state_dict_spec(9).keys())
    assert float(sd["layers.0.layers.0.layers.1.weight"].min()) == 1.0  # BN gamma init
    new = weights.make_state_dict(9, seed=4)
    m.load_state_dict(new)
    assert np.array_equal(m.state_dict()["linear.weight"].numpy(), new["linear.weight"])
    with pytest.raises(RuntimeError):
        m.load_state_dict({"linear.weight": new["linear.weight"]})
    bad = dict(new)
    bad["linear.weight"] = np.zeros((3, 128), np.float32)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)
    assert len(m.parameters()) == len(weights.trainable_keys(9))


def test_camal_errors_like_the_reference():
    from camal_amd import CamAL
    c = CamAL(device="cuda")
    with pytest.raises(ValueError):  # core.py:271-274
        c.predict(np.zeros((2, 16), np.float32))
    with pytest.raises(ValueError):  # core.py:377-380
        c.eval()
    with pytest.raises(ValueError):  # core.py:192-195
        c.test(np.zeros((2, 2, 16)))
    c.is_fitted = True
    with pytest.raises(AssertionError):  # core.py:276-278
        c.predict(np.zeros((2, 2, 16), np.float32))
    assert c.resnet_kwargs == {"in_channels": 1, "mid_channels": 64, "norm": "BatchNorm"}
    with pytest.raises(AssertionError):  # core.py:66-68
        CamAL(n_predictors=5).train((None, None), (None, None), (None, None), list_kernel_sizes=[5, 7])


def test_cpu_device_is_refused_loudly():
    from camal_amd import _lib, weights
    from camal_amd.engine import Ensemble
    with pytest.raises(_lib.CamalHipError):
        Ensemble([weights.make_state_dict(7)], device="cpu")


def test_undersampler_balances_and_is_seeded():
    from camal_amd.core import undersampler
    rng = np.random.default_rng(0)
    X = rng.normal(size=(100, 8))
    y = (rng.random(100) < 0.25).astype(int)
    Xa, ya = undersampler(X, y, seed=0)
    Xb, yb = undersampler(X, y, seed=0)
    assert Xa.dtype == np.float32 and np.array_equal(Xa, Xb) and np.array_equal(ya, yb)
    assert (ya == 1).sum() == (ya == 0).sum() == y.sum()


def test_undersampler_sampling_strategies():
    """imblearn's documented sampling_strategy forms (string, float ratio, dict, callable) on three classes / two."""
    from camal_amd.core import undersampler
    rng = np.random.default_rng(1)
    X = rng.normal(size=(300, 4))
    y3 = np.repeat([0, 1, 2], [150, 100, 50])
    cnt = lambda yy: [int((yy == c).sum()) for c in (0, 1, 2)]  # noqa: E731
    assert cnt(undersampler(X, y3, "auto")[1]) == cnt(undersampler(X, y3, "not minority")[1]) == [50, 50, 50]
    assert cnt(undersampler(X, y3, "majority")[1]) == [50, 100, 50]
    assert cnt(undersampler(X, y3, "not majority")[1]) == [150, 50, 50]
    assert cnt(undersampler(X, y3, "all")[1]) == [50, 50, 50]
    assert cnt(undersampler(X, y3, {0: 70, 1: 60})[1]) == [70, 60, 50]
    assert cnt(undersampler(X, y3, lambda yy: {0: 10})[1]) == [10, 100, 50]
    y2 = np.repeat([0, 1], [240, 60])
    assert cnt(undersampler(X, y2, 0.5)[1]) == [120, 60, 0]
    for bad in (0.1, 1.5, "minority", {0: 999}, {7: 1}):
        with pytest.raises(ValueError):
            undersampler(X, y2, bad)
    with pytest.raises(ValueError):
        undersampler(X, y3, 0.5)  # a float only for binary targets
    # every kept row is a row of X with its own label, no row twice
    Xs, ys = undersampler(X, y3, "all", seed=3)
    rows = {tuple(np.float32(r)): int(l) for r, l in zip(X, y3)}
    assert len({tuple(r) for r in Xs}) == len(Xs) and all(rows[tuple(r)] == int(l) for r, l in zip(Xs, ys))


def test_metrics_keys_and_values():
    from camal_amd.metrics import Classifmetrics, NILMmetrics
    y = np.array([0, 0, 1, 1, 1, 0])
    p = np.array([0.1, 0.6, 0.8, 0.4, 0.9, 0.2])
    m = Classifmetrics()(y, p)
    assert set(m) == {"ACCURACY", "BALANCED_ACCURACY", "PRECISION", "RECALL", "F1_SCORE", "F1_SCORE_MACRO",
                      "ROC_AUC_SCORE", "AP"}
    assert m["ACCURACY"] == round(4 / 6, 5) and m["ROC_AUC_SCORE"] == round(8 / 9, 5)
    n = NILMmetrics()(y=np.array([0.0, 10.0, 20.0]), y_hat=np.array([0.0, 10.0, 10.0]),
                      y_state=np.array([0, 1, 1]), y_hat_state=np.array([0, 1, 1]))
    assert n["MAE"] == round(10 / 3, 3) and n["F1_SCORE"] == 1.0 and n["TECA"] == round(1 - 10 / 60, 3)


def test_metrics_match_the_reference_fixture():
    """tests/golden/metrics_ref.npz holds the reference's own NILMmetrics / Classifmetrics numbers (sklearn)."""
    from camal_amd import synth
    from camal_amd.metrics import Classifmetrics, NILMmetrics
    from tests.helpers import load_case
    c = load_case("metrics_ref")
    for i, seed in enumerate(c["seeds"]):
        y, y_hat, y_state, y_hat_state, y_clf, score = synth.metric_arrays(int(seed))
        m = NILMmetrics()(y=y, y_hat=y_hat, y_state=y_state, y_hat_state=y_hat_state)
        assert [m[k] for k in c["loc_keys"]] == c["loc"][i].tolist()
        q = Classifmetrics()(y_clf, score)
        assert [q[k] for k in c["clf_keys"]] == c["clf"][i].tolist()


def test_nilm_to_clf_labels():
    from camal_amd.core import nilmdataset_to_clfdataset
    d = np.zeros((3, 2, 2, 5))
    d[1, 1, 1, 2] = 1
    X, y = nilmdataset_to_clfdataset(d)
    assert X.shape == (3, 5) and list(y) == [0, 1, 0]


def test_hand_counted_weight_ring_is_untouched_by_the_compiler(tmp_path):
    """tools/check_ring.py on the device assembly: no compiler-made instruction may touch a register that a
    hand-issued weight load (inline asm, landing asynchronously) still has in flight, and every such load takes
    its scalar base through the s_mov_b64 in front of it (VALU-writes-SGPR -> VMEM hazard)."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    asm = tmp_path / "camal.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(asm),
                           os.path.join(ROOT, "camal_amd", "csrc", "camal_hip.hip")], stderr=subprocess.DEVNULL)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_ring
    text = asm.read_text()
    assert text.count("ASMSTART") > 100 and "v_mfma_f32_16x16x32_bf16" in text
    assert check_ring.check(str(asm)) == []
    # ... and it did look at the kernels: LLVM prints a function's label as "name:   ; @name", which an earlier version
    # of the checker did not recognise as a label (it then walked no kernel of real output and reported nothing)
    assert check_ring.check.kernels_seen > 100


def test_ring_checker_sees_hazards_and_follows_out_of_line_blocks(tmp_path):
    """The checker itself, on hand-written assembly: a compiler-made instruction that touches a register with a
    ring load in flight is reported, on the straight path and in a block the compiler moved behind the kernel's end
    (entered with what was in flight at the branch, not with what the text in front of it left); the same block is
    clean when the branch leaves after the consuming MFMA."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_ring
    load = ("\t;;#ASMSTART\n\ts_mov_b64 s[4:5], s[2:3]\n\tglobal_load_dwordx4 v[10:13], v1, s[4:5]\n\t;;#ASMEND\n")
    mfma = "\tv_mfma_f32_16x16x32_bf16 v[40:43], v[10:13], v[20:23], v[40:43]\n"
    tail = "\ts_endpgm\n.LBB0_9:\n\tds_read_b128 v[10:13], v2\n\ts_branch .LBB0_2\n"

    def run(body, label="_ZN5camal4testEv:                     ; @_ZN5camal4testEv\n"):
        f = tmp_path / "k.s"
        f.write_text(label + body)  # the label as LLVM prints it, comment included
        return check_ring.check(str(f))

    assert run(load + mfma + "\tv_mov_b32_e32 v10, 0\n\ts_endpgm\n") == []
    assert len(run(load + "\tv_mov_b32_e32 v10, 0\n" + mfma + "\ts_endpgm\n")) == 1
    # out-of-line block reached while the load is in flight: a finding, although the text in front of it ends clean
    assert len(run(load + "\ts_cbranch_vccz .LBB0_9\n.LBB0_2:\n" + mfma + tail)) == 1
    # reached after the consumer: clean
    assert run(load + mfma + "\ts_cbranch_vccz .LBB0_9\n.LBB0_2:\n" + tail) == []
    # loop back edge: the load issued at the bottom of the body is in flight at the loop header, where a
    # compiler-made copy of its register (a phi copy in front of the consumer) must be reported
    loop = (".LBB0_1:\n\tv_mov_b32_e32 v30, v10\n" + mfma + load + "\ts_cbranch_scc1 .LBB0_1\n\ts_waitcnt vmcnt(0)\n\ts_endpgm\n")
    assert len(run(load + loop)) == 1
    assert len(run(loop)) == 1   # seen only through the back edge: nothing is in flight on the way in
    clean_loop = (".LBB0_1:\n" + mfma + "\tv_mov_b32_e32 v30, v10\n" + load + "\ts_cbranch_scc1 .LBB0_1\n\ts_waitcnt vmcnt(0)\n\ts_endpgm\n")
    assert run(load + clean_loop) == []
    # a load whose scalar base does not come from the s_mov_b64 in front of it
    assert len(run("\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v1, s[4:5]\n\t;;#ASMEND\n" + mfma + "\ts_endpgm\n")) == 1


def test_bench_spawns_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` without a launcher: the parent starts the two ranks as children (before anything
    touches a GPU) and rank 0's line comes through; --dry-run rehearses that path on CPU with gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"dry_run": True, "n_gpus": 2, "max_rank_seen": 1}


def test_tsloader_shuffle_is_dataloaders_order():
    """TSLoader(shuffle=True) walks the samples in the order DataLoader(TSDataset, shuffle=True) (core.py:109-112)
    would under the same torch.manual_seed, epoch after epoch."""
    import numpy as np
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    from camal_amd.trainer import TSLoader
    n = 37
    torch.manual_seed(123)
    dl = DataLoader(TensorDataset(torch.arange(n)), batch_size=8, shuffle=True)
    ref = [[b[0].tolist() for b in dl] for _ in range(3)]
    torch.manual_seed(123)
    ours = TSLoader(np.zeros((n, 5), np.float32), np.arange(n), batch_size=8, shuffle=True)
    got = [[lab.tolist() for _, lab in ours] for _ in range(3)]
    assert got == ref


def test_reduce_lr_on_plateau_is_torchs():
    """trainer.ReduceLROnPlateau against torch.optim.lr_scheduler.ReduceLROnPlateau as the reference configures it
    (torch_trainer.py:63-66: 'min', patience, eps=1e-7) on random loss sequences with plateaus, ties and NaN."""
    import numpy as np
    import torch
    from camal_amd.trainer import ReduceLROnPlateau
    rng = np.random.default_rng(0)
    for trial in range(40):
        patience = int(rng.integers(0, 5))
        lr0 = float(10.0 ** rng.uniform(-6, -2))
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=lr0)
        ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "min", patience=patience, eps=1e-7)
        ours = ReduceLROnPlateau(lr0, patience=patience, eps=1e-7)
        losses = np.abs(np.cumsum(rng.normal(-0.02, 0.1, size=60))) + 0.05
        losses[rng.integers(0, 60, size=12)] = losses[rng.integers(0, 60)]  # exact ties
        losses = np.round(losses, int(rng.integers(1, 6)))                   # plateaus
        if trial % 7 == 0:
            losses[int(rng.integers(5, 50))] = np.nan
        for i, v in enumerate(losses):
            ref.step(float(v))
            got = ours.step(float(v))
            assert got == opt.param_groups[0]["lr"], (trial, i, got, opt.param_groups[0]["lr"])


def test_trainer_epoch_logic_equals_the_references_decisions():
    """tests/golden/trainer_control_flow.npz: what the REFERENCE's BasedClassifTrainer.train (torch_trainer.py:92-171)
    did on 16 scripted runs (epoch results replaced by scripts, everything else its own code: torch's
    ReduceLROnPlateau, EarlyStopper after the warm-up, best-loss snapshot).  This repo's trainer gets the same
    scripts, host state only (no GPU): LR after every epoch, epochs run, best epoch and value must be equal."""
    import numpy as np
    from camal_amd.trainer import BasedClassifTrainer
    from tests.helpers import load_case, scripted_trainer_scenarios
    ref = load_case("trainer_control_flow")

    class FakeModel:
        def state_dict(self):
            return {"w": 1}

    class FakeEngine:
        def _dp(self):
            return False

        def store_to_model(self, model):
            pass

        def optimizer_state_dict(self, lr, wd):
            return {"lr": lr}

    stops = drops = 0
    for i, sc in enumerate(scripted_trainer_scenarios()):
        tr = BasedClassifTrainer.__new__(BasedClassifTrainer)
        tr._init_host(FakeModel(), [], [0] if sc["has_valid"] else None, sc["lr"], 0, sc["patience_es"], sc["patience_rlr"],
                      "cpu", False, sc["n_warmup_epochs"], None)
        tr.engine = FakeEngine()
        calls = {"n": -1}
        lrs = []

        def fake_train(sc=sc, calls=calls, tr=tr, lrs=lrs):
            if calls["n"] >= 0:
                lrs.append(tr.lr)  # the LR the previous epoch's scheduler step left
            calls["n"] += 1
            return sc["train_loss"][calls["n"]], sc["train_acc"][calls["n"]]

        def fake_eval(sc=sc, calls=calls):
            return sc["valid_loss"][calls["n"]], sc["valid_acc"][calls["n"]]

        tr._BasedClassifTrainer__train = fake_train
        tr._BasedClassifTrainer__evaluate = fake_eval
        tr.train(sc["n_epochs"])
        lrs.append(tr.lr)
        if sc["patience_rlr"] is not None:
            assert np.array_equal(np.array(lrs), ref[f"lrs_{i}"]), (i, lrs, ref[f"lrs_{i}"])
        assert tr.lr == float(ref[f"final_lr_{i}"]), i
        assert len(tr.loss_train_history) == int(ref[f"epochs_run_{i}"]), i
        assert tr.passed_epochs == int(ref[f"passed_epochs_{i}"]), i
        assert tr.best_loss == float(ref[f"best_loss_{i}"]), i
        assert tr.log.get("epoch_best_loss", -1) == int(ref[f"epoch_best_{i}"]), i
        if int(ref[f"epoch_best_{i}"]) >= 0:
            assert tr.log["value_best_loss"] == float(ref[f"value_best_{i}"]), i
            assert tr.log["valid_metrics"] == float(ref[f"valid_metrics_{i}"]), i
        assert {"training_time", "loss_train_history", "loss_valid_history"} <= set(tr.log)
        stops += len(tr.loss_train_history) < sc["n_epochs"]
        drops += tr.lr < sc["lr"]
    assert stops >= 5 and drops >= 8  # the scripts do exercise early stops and LR drops


def test_control_path_collectives_use_a_gpu_tensor_under_rccl(monkeypatch):
    """under an RCCL process group ("nccl") every tensor handed to a collective must live on a GPU
    (a CPU tensor raises 'No backend type associated with device type cpu').  A fake torch.distributed whose backend
    says "nccl" + a recording control_device: the trainer's epoch loop (train), its per-epoch sums (__train) and the
    loader's seed broadcast must all ask for the ENGINE's device, never fall back to the host."""
    import numpy as np
    import torch
    from camal_amd import parallel
    from camal_amd.trainer import BasedClassifTrainer, TSLoader

    class FakeDist:
        class ReduceOp:
            SUM = "sum"

        def __init__(self, backend):
            self.backend, self.calls = backend, []

        def is_available(self):
            return True

        def is_initialized(self):
            return True

        def get_world_size(self):
            return 2

        def get_rank(self):
            return 0

        def get_backend(self):
            return self.backend

        def broadcast(self, t, src):
            self.calls.append(("broadcast", t.device))

        def all_reduce(self, t, op=None, async_op=False):
            self.calls.append(("all_reduce", t.device))

    # the device rule itself
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 3)
    assert parallel.control_device(FakeDist("gloo"), "cuda:1") == torch.device("cpu")
    assert parallel.control_device(FakeDist("nccl"), "cuda:1") == torch.device("cuda:1")
    assert parallel.control_device(FakeDist("nccl")) == torch.device("cuda", 3)

    fake = FakeDist("nccl")
    asked = []

    def recording_control_device(d, device=None):
        assert d is fake and d.get_backend() == "nccl"
        asked.append(device)
        return torch.device("cpu")  # no GPU here: the tensor is made on the host, what was ASKED for is the point

    monkeypatch.setattr(parallel, "control_device", recording_control_device)
    monkeypatch.setattr(parallel, "_active", lambda dist=None: fake)
    dev = torch.device("cuda", 0)

    class FakeEngine:
        device = dev

        def _dp(self):
            return True

        def store_to_model(self, model):
            pass

        def optimizer_state_dict(self, lr, wd):
            return {}

        def forward_backward(self, ts, labels, global_batch=None, loss_out=None, ncorrect_out=None, criterion=None):
            pass

        def optimizer_step(self, lr, wd):
            pass

    class FakeModel:
        def state_dict(self):
            return {}

    loader = TSLoader(np.zeros((10, 4), np.float32), np.arange(10) % 2, batch_size=4, shuffle=True)
    loader.sync_shuffle, loader.sync_device = True, dev
    tr = BasedClassifTrainer.__new__(BasedClassifTrainer)
    tr._init_host(FakeModel(), loader, None, 1e-3, 0, 2, 1, "cuda:0", True, 0, None)
    tr.engine = FakeEngine()
    real_zeros = torch.zeros
    monkeypatch.setattr(torch, "zeros", lambda *a, **k: real_zeros(*a, **{**k, "device": "cpu"}))
    import contextlib
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    tr.train(2)
    kinds = [k for k, _ in fake.calls]
    # per epoch: the shuffle seed (broadcast), the loss / accuracy sums (all_reduce), rank 0's decision value (broadcast)
    assert kinds == ["broadcast", "all_reduce", "broadcast"] * 2, kinds
    assert asked == [dev] * 6, asked
    # a loader that was not told to share rank 0's shuffle issues no collective (members spread over the ranks)
    fake.calls.clear()
    list(TSLoader(np.zeros((10, 4), np.float32), np.arange(10) % 2, batch_size=4, shuffle=True))
    assert fake.calls == []


def test_trainer_host_helpers_save_compress_plot_reduce_lr(tmp_path):
    """save / compress / plot_history / reduce_lr of BasedClassifTrainer (torch_trainer.py:209-258), host state only."""
    import lzma
    import os

    import matplotlib
    matplotlib.use("Agg")
    import torch
    from camal_amd.trainer import BasedClassifTrainer

    class FakeModel:
        def state_dict(self):
            return {"w": 1}

    class FakeEngine:
        def _dp(self):
            return False

        def store_to_model(self, model):
            pass

        def optimizer_state_dict(self, lr, wd):
            return {"lr": lr}

    tr = BasedClassifTrainer.__new__(BasedClassifTrainer)
    fig, ckpt = str(tmp_path / "loss.png"), str(tmp_path / "model")
    tr._init_host(FakeModel(), [], [0], 1e-3, 0, None, 2, "cpu", False, 0, None, plotloss=True, path_fig=fig,
                  save_checkpoint=True, path_checkpoint=ckpt)
    tr.engine = FakeEngine()
    losses = iter([(0.9, 0.5), (0.7, 0.6), (0.6, 0.7)])
    tr._BasedClassifTrainer__train = lambda: next(losses)
    tr._BasedClassifTrainer__evaluate = lambda: (0.8, 0.55)
    tr.train(3)
    assert os.path.getsize(fig) > 0                       # plotloss=True draws the curves at the end of train()
    log = torch.load(ckpt + ".pt", weights_only=False)   # a file this test wrote
    assert log["loss_train_history"] == [0.9, 0.7, 0.6] and "best_model_state_dict" in log
    tr.compress()
    assert not os.path.exists(ckpt + ".pt")
    with open(ckpt + ".xz", "rb") as f:
        assert len(lzma.decompress(f.read())) > 0
    tr.reduce_lr(1e-5)
    assert tr.lr == 1e-5 and tr.scheduler.lr == 1e-5


def test_dataloader_over_tsdataset_becomes_the_same_tsloader():
    """BasedClassifTrainer takes the reference's DataLoader(TSDataset(X, y), batch_size, shuffle) (core.py:92-119): the
    loader built from it walks the same batches in the same order under torch.manual_seed as the DataLoader itself."""
    import numpy as np
    import torch
    from camal_amd.trainer import TSDataset, TSLoader, as_tsloader
    rng = np.random.default_rng(0)
    X, y = rng.normal(size=(23, 40)).astype(np.float32), rng.integers(0, 2, 23)
    ds = TSDataset(X, y)
    assert ds.samples.shape == (23, 1, 40) and len(ds) == 23 and ds[3][1] == y[3]
    for shuffle in (False, True):
        dl = torch.utils.data.DataLoader(ds, batch_size=5, shuffle=shuffle)
        tl = as_tsloader(dl)
        assert isinstance(tl, TSLoader) and tl.batch_size == 5 and tl.shuffle == shuffle and len(tl) == len(dl)
        torch.manual_seed(11)
        ref = [(ts.numpy(), lb.numpy()) for ts, lb in dl]
        torch.manual_seed(11)
        got = list(tl)
        assert len(ref) == len(got)
        for (a, b), (c, d) in zip(ref, got):
            assert np.array_equal(a, c) and np.array_equal(b, d)
    assert as_tsloader(None) is None and as_tsloader(tl) is tl
    scaled = TSDataset(rng.normal(3.0, 2.0, size=(6, 50)), scaler=True)
    assert np.allclose(scaled.samples.mean(axis=-1), 0, atol=1e-5) and scaled.labels is None


def test_scaler_channel_inverses_match_inverse_transform():
    """NILMscaler.inverse_transform_appliance / inverse_transform_agg_power (torch_dataset.py:158-193) against
    inverse_transform on the same channels, NumPy and torch."""
    import numpy as np
    import torch
    from camal_amd.windows import NILMscaler
    rng = np.random.default_rng(1)
    data = np.abs(rng.normal(200, 80, size=(9, 2, 2, 30)))
    for kind in ("StandardScaling", "MaxScaling", 1000):
        sc = NILMscaler(power_scaling_type=kind, appliance_scaling_type="SameAsPower")
        scaled = sc.fit_transform(data.copy())
        back = sc.inverse_transform(scaled)
        assert np.allclose(back[:, :, 0, :], data[:, :, 0, :])
        app = sc.inverse_transform_appliance(scaled[:, 1:, 0, :])
        agg = sc.inverse_transform_agg_power(scaled[:, :1, 0, :])
        assert np.allclose(app[:, 0], data[:, 1, 0, :]) and np.allclose(agg[:, 0], data[:, 0, 0, :])
        app_t = sc.inverse_transform_appliance(torch.from_numpy(scaled[0, 1:, 0, :]))
        assert torch.is_tensor(app_t) and app_t.shape == (1, 1, 30) and np.allclose(app_t.numpy()[0, 0], data[0, 1, 0, :])
