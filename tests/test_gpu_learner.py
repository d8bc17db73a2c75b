"""the packaged DQN learner (uavenv_amd/learner.py) -- the reference's trainer loop
(agents/dqn/dqn.py:1077-1099, :1276-1288, :1324: SB3 DQN over VecFrameStack(DummyVecEnv)) on device.

stable-baselines3 is not installed in the build image, so a run of SB3 cannot pin this row ("parity unpinned"); what is
checked here: (a) the arithmetic of one update -- TD target, smooth-L1, masking, gradient clipping, the Adam step, the
target-network cadence, the schedules -- against formulas written out by hand; (b) that a short run actually learns.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mods():
    import torch
    import uavenv_amd as U
    from uavenv_amd import learner as LR
    return torch, U, LR


def test_reference_hyperparameters_and_schedules():
    torch, U, LR = _mods()
    hp = LR.REFERENCE_HYPERPARAMS
    assert (hp["buffer_size"], hp["batch_size"], hp["learning_starts"], hp["target_update_interval"], hp["train_freq"]) == \
        (150_000, 256, 25_000, 5_000, 4)                                        # dqn.py:1083-1089
    assert (hp["gamma"], hp["exploration_fraction"], hp["exploration_final_eps"]) == (0.99, 0.25, 0.03)
    assert tuple(hp["net_arch"]) == (512, 512, 256) and hp["n_stack"] == 4 and hp["total_timesteps"] == 3_000_000
    # dqn.py:1081 as SB3 evaluates it: the lambda receives progress_REMAINING
    lr = hp["learning_rate"]
    assert lr(1.0) == pytest.approx(3e-4 * 0.2) and lr(0.5) == pytest.approx(3e-4 * 0.6) and lr(0.0) == pytest.approx(3e-4)
    # SB3 get_linear_fn(1.0, 0.03, 0.25)
    assert LR.linear_epsilon(1.0, 1.0, 0.03, 0.25) == 1.0
    assert LR.linear_epsilon(0.875, 1.0, 0.03, 0.25) == pytest.approx(1.0 + 0.125 * (0.03 - 1.0) / 0.25)
    assert LR.linear_epsilon(0.74, 1.0, 0.03, 0.25) == 0.03 and LR.linear_epsilon(0.0, 1.0, 0.03, 0.25) == 0.03
    env = U.BatchedUAVEnv(64, num_sensors=10, seed=0)
    L = LR.DQNLearner(env, **hp)
    assert L.target_every == 5000 // 64 and L.ring.capacity * 64 >= 150_000 and (L.ring.capacity - L.ring.L) * 64 <= 150_000 + 64 * L.ring.L
    q = L.q
    assert [m.out_features for m in q.head if hasattr(m, "out_features")] == [512, 512, 256, 5]
    assert q.head[0].in_features == 4 * env.obs_dim
    env.close()


def test_one_update_matches_hand_written_arithmetic():
    """TD target / smooth-L1 / valid mask / clipping / Adam on a fixed batch drawn with `sample_stacked`."""
    torch, U, LR = _mods()
    E, k, gamma = 96, 3, 0.9
    env = U.BatchedUAVEnv(E, num_sensors=10, max_steps=9, grid_size=(60, 60), seed=5)
    L = LR.DQNLearner(env, learning_rate=1e-2, buffer_size=96 * 40, batch_size=64, gamma=gamma, learning_starts=0,
                      target_update_interval=96 * 1000, train_freq=1, gradient_steps=1, net_arch=(32,), n_stack=k,
                      total_timesteps=10**6, max_grad_norm=0.5, seed=3, reward_scale=1e-3)
    L.collect(30)
    with torch.no_grad():                     # make the target network differ from the online one
        for p in L.q_target.parameters():
            p.mul_(0.5)
    gen_state = L.gen.get_state()
    batch = L.ring.sample_stacked(64, k, generator=L.gen)
    assert batch["done"].any() and batch["valid"].all() and batch["obs"].shape == (64, k * env.obs_dim)
    # ---- by hand -------------------------------------------------------------------------------------------
    W1, b1, W2, b2 = [p.detach().double().clone() for p in L.q.parameters()]
    T1, c1, T2, c2 = [p.detach().double() for p in L.q_target.parameters()]
    x, xn = batch["obs"].double(), batch["next_obs"].double()
    a, r = batch["action"], batch["reward"].double()
    h = torch.relu(x @ W1.t() + b1)
    qsa = (h @ W2.t() + b2)[torch.arange(64), a]
    tgt = 1e-3 * r + gamma * (torch.relu(xn @ T1.t() + c1) @ T2.t() + c2).max(1).values
    d = qsa - tgt
    huber = torch.where(d.abs() < 1.0, 0.5 * d * d, d.abs() - 0.5)
    loss_by_hand = huber.mean()
    loss = LR.td_loss(L.q, L.q_target, batch, gamma, 1e-3)
    assert float(loss.detach()) == pytest.approx(float(loss_by_hand), rel=1e-5)
    # gradient of the hand-written loss w.r.t. the output layer: dL/dq = clamp(d, -1, 1) / 64 on the taken action
    g_out = torch.zeros(64, 5, dtype=torch.float64, device=env.device)
    g_out[torch.arange(64), a] = d.clamp(-1, 1) / 64
    gW2, gb2 = g_out.t() @ h, g_out.sum(0)
    gh = (g_out @ W2) * (h > 0)
    gW1, gb1 = gh.t() @ x, gh.sum(0)
    grads = [gW1, gb1, gW2, gb2]
    norm = math.sqrt(sum(float((g * g).sum()) for g in grads))
    clip = min(1.0, 0.5 / (norm + 1e-6))                                         # torch clip_grad_norm_
    # first Adam step (bias-corrected m / sqrt(v) = sign-like): p -= lr * g / (|g| + eps * sqrt(1 - b2) ...) -> use the exact form
    lr, b1_, b2_, eps = L.lr_schedule(L.progress_remaining()), 0.9, 0.999, 1e-8
    want = []
    for p0, g in zip([W1, b1, W2, b2], grads):
        g = g * clip
        m, v = (1 - b1_) * g, (1 - b2_) * g * g
        want.append(p0 - lr * (m / (1 - b1_)) / ((v / (1 - b2_)).sqrt() + eps))
    L.gen.set_state(gen_state)                                                   # train() draws the same batch
    L.train(1)
    for p, w in zip(L.q.parameters(), want):
        assert torch.allclose(p.detach().double(), w, rtol=2e-4, atol=2e-6)
    assert L.n_updates == 1
    # masking: an invalid transition contributes nothing and the mean is over the valid ones
    b2m = dict(batch); v = batch["valid"].clone(); v[:10] = False; b2m["valid"] = v
    with torch.no_grad():
        lm = LR.td_loss(L.q, L.q_target, b2m, gamma, 1e-3)
        full = torch.nn.functional.smooth_l1_loss(
            L.q(batch["obs"]).gather(1, a.unsqueeze(1)).squeeze(1),
            1e-3 * batch["reward"] + gamma * L.q_target(batch["next_obs"]).max(1).values, reduction="none")
    assert float(lm) == pytest.approx(float(full[10:].mean()), rel=1e-5)
    env.close()


def test_target_network_cadence_and_counters():
    torch, U, LR = _mods()
    E = 32
    env = U.BatchedUAVEnv(E, num_sensors=5, max_steps=50, grid_size=(40, 40), seed=1)
    L = LR.DQNLearner(env, learning_rate=1e-3, buffer_size=E * 64, learning_starts=E * 3, target_update_interval=E * 10,
                      train_freq=2, gradient_steps=3, net_arch=(16,), n_stack=2, total_timesteps=E * 40, seed=0)
    assert L.target_every == 10
    same = lambda: all(torch.equal(a, b) for a, b in zip(L.q.state_dict().values(), L.q_target.state_dict().values()))
    seen = []
    L.learn(callback=lambda l: seen.append((l.n_calls, l.n_updates, same())))
    assert L.num_timesteps == E * 40 and L.n_calls == 40
    # learning starts once more than learning_starts transitions are in: after the rollout that reaches 4 vector steps
    assert [u for c, u, _ in seen] == [0] + [3 * i for i in range(1, 20)]
    # the copy happens at vector steps 10, 20, 30, 40 (inside collect); updates after a rollout make the nets differ again
    for c, u, s in seen:
        assert s == (u == 0)                                # (the callback runs after train(): equal only before the first update)
    L.collect(0)
    assert L.exploration_rate() == 0.03 and L.progress_remaining() == 0.0
    env.close()


def test_short_run_beats_the_uniform_random_policy():
    """300 k transitions on a 20 x 20 grid with 5 sensors and 80-step episodes (about 6 s on an MI355X): the greedy policy of
    the trained network earns at least 1.2 x the mean episode return of the uniform-random policy on 256 held-out
    environments (measured while writing the test: 1.41 x and 1.57 x for seeds 0 and 1; the rewards are scaled by 1e-4 in
    the loss, see DQNLearner.reward_scale).  On this small grid every sensor is in radio range from everywhere, so the task
    is WHEN to collect, not where to fly; it shows the loop learns, not that it solves the reference's 500 x 500 task."""
    torch, U, LR = _mods()
    kw = dict(num_sensors=5, grid_size=(20, 20), max_steps=80)
    env = U.BatchedUAVEnv(256, seed=1, **kw)
    held_out = U.BatchedUAVEnv(256, seed=99, **kw)
    L = LR.DQNLearner(env, learning_rate=1e-3, buffer_size=100_000, learning_starts=2_000, target_update_interval=2_000,
                      train_freq=1, gradient_steps=4, net_arch=(128, 128), n_stack=2, total_timesteps=300_000,
                      exploration_fraction=0.5, reward_scale=1e-4, seed=1)
    random_return, n = L.evaluate(held_out, 1, "random")
    assert n == 256 and random_return > 0
    L.learn()
    assert L.n_updates > 4000 and np.isfinite(float(L.last_loss.detach()))
    greedy_return, n = L.evaluate(held_out, 1, "greedy")
    assert n == 256
    assert greedy_return >= 1.2 * random_return, (greedy_return, random_return)
    st = env.episode_stats()
    assert (st["valid"] == 1).all() and (st["length"] == 80).all()              # every training environment finished episodes
    env.close(); held_out.close()
