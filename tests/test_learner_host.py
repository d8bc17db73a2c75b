"""CPU: the device-independent arithmetic of the packaged DQN learner (uavenv_amd/learner.py) -- the schedules SB3 evaluates
for the reference's hyper-parameters (agents/dqn/dqn.py:1077-1099), the TD loss with its validity mask, and the network shapes.
The loops themselves need the environment (a GPU): tests/test_gpu_learner.py."""
import math

import pytest
import torch

from uavenv_amd import learner as LR


def test_reference_hyperparameters_as_sb3_evaluates_them():
    hp = LR.REFERENCE_HYPERPARAMS
    assert (hp["buffer_size"], hp["batch_size"], hp["learning_starts"], hp["target_update_interval"], hp["train_freq"],
            hp["gradient_steps"]) == (150_000, 256, 25_000, 5_000, 4, 1)                     # dqn.py:1083-1090
    assert (hp["gamma"], hp["exploration_fraction"], hp["exploration_final_eps"]) == (0.99, 0.25, 0.03)
    assert tuple(hp["net_arch"]) == (512, 512, 256) and hp["total_timesteps"] == 3_000_000
    lr = hp["learning_rate"]                                # SB3 calls the schedule with progress_REMAINING: the rate rises
    assert [lr(p) for p in (1.0, 0.75, 0.5, 0.0)] == pytest.approx([6e-5, 1.2e-4, 1.8e-4, 3e-4])
    assert lr(1.0) == pytest.approx(3e-4 * 0.2) and min(lr(p / 10) for p in range(11)) == pytest.approx(6e-5)


def test_linear_epsilon_is_sb3_get_linear_fn():
    f = lambda p: LR.linear_epsilon(p, 1.0, 0.03, 0.25)
    assert f(1.0) == 1.0 and f(0.75) == pytest.approx(0.03) and f(0.7) == 0.03 and f(0.0) == 0.03
    assert f(0.9) == pytest.approx(1.0 + 0.1 * (0.03 - 1.0) / 0.25)
    xs = [f(1.0 - k / 100) for k in range(101)]
    assert all(a >= b for a, b in zip(xs, xs[1:]))          # never rises


def test_td_loss_is_masked_smooth_l1_on_the_bootstrapped_target():
    torch.manual_seed(0)
    q, t = LR.QNetwork(7, 3, (16,)), LR.QNetwork(7, 3, (16,))
    B = 12
    batch = dict(obs=torch.randn(B, 21), next_obs=torch.randn(B, 21), action=torch.randint(0, 5, (B,)),
                 reward=torch.randn(B) * 3, valid=torch.tensor([True] * 9 + [False] * 3))
    loss = LR.td_loss(q, t, batch, 0.9, 0.5)
    with torch.no_grad():
        target = 0.5 * batch["reward"] + 0.9 * t(batch["next_obs"]).max(1).values        # no (1 - done): truncation bootstraps
        cur = q(batch["obs"])[torch.arange(B), batch["action"]]
        d = (cur - target).abs()
        huber = torch.where(d < 1.0, 0.5 * d * d, d - 0.5)
    assert float(loss.detach()) == pytest.approx(float(huber[:9].mean()), rel=1e-6)
    loss.backward()
    assert all(p.grad is not None for p in q.parameters()) and all(p.grad is None for p in t.parameters())
    none_valid = dict(batch, valid=torch.zeros(B, dtype=torch.bool))
    assert float(LR.td_loss(q, t, none_valid, 0.9, 0.5).detach()) == 0.0                   # empty mask: 0 / max(0, 1)


def test_q_network_shapes_follow_sb3_mlp_policy_and_the_attention_extractor():
    q = LR.QNetwork(153, 4, (512, 512, 256))
    assert [m.in_features for m in q.head if hasattr(m, "in_features")] == [612, 512, 512, 256]
    assert q(torch.zeros(3, 612)).shape == (3, 5)
    qa = LR.QNetwork(153, 10, (64,), extractor="attention")
    assert qa.features.features_dim == 128 and qa(torch.rand(2, 1530)).shape == (2, 5)
    n_params = sum(p.numel() for p in qa.features.parameters())
    assert n_params == (30 * 64 + 64) + 2 * 64 + (3 * 64 + 64) + (3 * 64 * 64 + 3 * 64) + (64 * 64 + 64) + 2 * 64 + (128 * 128 + 128)
