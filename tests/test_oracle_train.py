"""CPU: oracle/torch_train.py (functional restatement of the training step) against fixtures
produced by the reference's own CamALResNet / CrossEntropyLoss / AdamW / BasedClassifTrainer."""
import numpy as np
import pytest

from camal_amd import weights
from oracle import torch_train
from tests.helpers import golden_cases, is_bn_fed_bias, load_case, pack_train_views, train_case_inputs

CASES = golden_cases("train_")


def test_training_fixtures_present():
    assert len(CASES) >= 2


@pytest.mark.parametrize("name", CASES)
def test_one_step_gradients(name):
    c = load_case(name)
    k, x, y, sd0 = train_case_inputs(c)
    B = int(c["batch"])
    loss, _n, grads, sd1 = torch_train.loss_and_grads(sd0, x[:B], y[:B])
    keys = weights.trainable_keys(k)
    assert abs(loss - float(c["step1_loss"])) < 1e-5
    norms, small, big = pack_train_views(keys, grads)
    live = np.array([not is_bn_fed_bias(kk) for kk in keys])
    assert np.allclose(norms[live], c["step1_grad_norms"][live], rtol=1e-3, atol=1e-7)
    # the BN-fed conv biases have exact gradient 0: the reference's values are rounding noise
    assert c["step1_grad_norms"][~live].max() < 1e-4 * c["step1_grad_norms"][live].max()
    assert np.allclose(small, c["step1_small_grads"], rtol=1e-3, atol=1e-5 * np.abs(c["step1_small_grads"]).max())
    assert np.allclose(big, c["step1_big_grad_slices"], rtol=1e-3, atol=1e-4 * np.abs(c["step1_big_grad_slices"]).max())
    run = np.concatenate([sd1[kk].ravel() for kk in sd1 if kk.endswith("running_mean") or kk.endswith("running_var")])
    assert np.allclose(run, c["step1_running"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", CASES)
def test_trainer_epochs(name):
    c = load_case(name)
    k, x, y, sd0 = train_case_inputs(c)
    B, epochs = int(c["batch"]), int(c["epochs"])
    batches = [(x[s:s + B], y[s:s + B]) for s in range(0, len(x), B)] * epochs
    losses, sd = torch_train.train_steps(sd0, batches, lr=float(c["lr"]))
    per_epoch = np.array(losses).reshape(epochs, -1).mean(axis=1)
    assert np.allclose(per_epoch, c["loss_train_history"], rtol=2e-3)
    assert np.abs(sd["linear.weight"] - c["final_linear_weight"]).max() < 5e-4
