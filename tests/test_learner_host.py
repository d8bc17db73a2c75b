"""CPU: the device-independent arithmetic of the packaged DQN learner (uavenv_amd/learner.py) -- the schedules SB3 evaluates
for the reference's hyper-parameters (agents/dqn/dqn.py:1077-1099), the TD loss with its validity mask, and the network shapes.
The loops themselves need the environment (a GPU): tests/test_gpu_learner.py."""
import math
import os

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


# ---------------------------------------------------------------------------------------------------------------
# several ranks (BASELINE config 4 as a training run), on CPU over gloo with the doubles of tests/learner_doubles.py
# ---------------------------------------------------------------------------------------------------------------
def _learner_worker(rank, world, port, q, replicate=True):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist
    import uavenv_amd  # noqa: F401
    from uavenv_amd import learner as LR
    from learner_doubles import ToyEnv, TorchFrameStack
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    E, D, k = 6, 12, 3
    env = ToyEnv(E, D, rank=rank, period=7)
    L = LR.DQNLearner(env, learning_rate=1e-2, buffer_size=2 * E * 40, batch_size=32, gamma=0.9, learning_starts=0,
                      target_update_interval=2 * E * 6, train_freq=4, gradient_steps=1, net_arch=(16, 8), n_stack=k,
                      total_timesteps=10**6, seed=11, chunk_len=4, frame_stack_cls=TorchFrameStack, replicate_replay=replicate)
    ok = L.world == world and L.local_batch == 32 // world and L.n_envs_total == E * world and not L._graphs_usable()
    init = [p.detach().clone() for p in L.q.parameters()]
    L.learn(total_timesteps=2 * E * 4 * 7)                 # 7 rollouts of 4 vector steps; updates start once k + 2 slots are visible
    ok &= L.n_updates >= 5 and L.n_calls == 28
    # (1) replicas: bit-identical weights on every rank after the updates, and they did change
    flat = torch.cat([p.detach().reshape(-1) for p in L.q.parameters()])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    ok &= all(torch.equal(both[0], b) for b in both[1:])
    ok &= any(not torch.equal(a, b.detach()) for a, b in zip(init, L.q.parameters()))
    # (2) the ranks drew DIFFERENT batches (own generators) of batch_size / world transitions each
    ok &= L.gen.initial_seed() == 11 * 7919 + 13 + rank
    if not replicate:
        # (3') rank-local replay: nothing but this rank's own transitions in its ring, no exchange -- and still identical replicas
        ring = L.ring
        ok &= ring.world == 1 and not ring.exchange
        b = ring.sample_stacked(300, k, generator=torch.Generator().manual_seed(3))
        ok &= bool((torch.floor(b["obs"][:, -D] / 1000) == rank).all()) and bool(b["done"].any()) and bool(b["valid"].all())
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, bool(ok)))
        return
    # (3) this rank's ring holds the other rank's transitions, episode ends with their terminal rows included
    ring, other = L.ring, 1 - rank
    L.ring.drain()
    n, oldest = ring.window_state()
    seen_remote_end = 0
    for j in range(n - 1):
        slot = (oldest + j) % ring.capacity
        nxt = (slot + 1) % ring.capacity
        for e in range(E):
            o = ring.obs_at(slot, other, e)
            t = round(float(o[0] % 1) * 1000)
            ok &= int(o[0]) == 1000 * other + e                                      # the other rank's env e, step t
            aux = ring.aux_at(nxt, other, e)
            ended = (t + 1 + e + other) % 7 == 0
            ok &= bool(aux[2] > 0.5) == ended
            if ended:
                tk = int(ring.tickets_at(nxt, other, e))
                row = ring.terminal_at(nxt, other, tk)
                ok &= tk >= 0 and abs(float(row[0]) - (1000 * other + e + (t + 1) / 1000 + 0.5)) < 1e-3
                seen_remote_end += 1
    ok &= seen_remote_end >= 3
    b = ring.sample_stacked(400, k, generator=torch.Generator().manual_seed(3))
    src = torch.floor(b["obs"][:, -D] / 1000)                                        # newest frame's rank tag
    ok &= bool((src == other).any()) and bool((src == rank).any()) and bool(b["valid"].all())
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("replicate", [True, False], ids=["shared_ring_allgather", "rank_local_rings"])
def test_two_rank_learner_keeps_replicas_identical_and_shares_the_ring(replicate):
    """BASELINE config 4 as training: each rank steps its shard, the ring all-gathers chunks (terminal sections included),
    every rank draws batch_size / world samples, ONE flat all-reduce averages the gradients: identical weights everywhere.
    replicate_replay=False: no transition crosses ranks (stratified sampling of the union of the rank-local buffers)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + (0 if replicate else 7)
    procs = [ctx.Process(target=_learner_worker, args=(r, 2, port, q, replicate)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_checkpoint_round_trip_on_the_cpu_doubles(tmp_path):
    """DQNLearner.save / load with the PyTorch update path on CPU stand-ins for the environment and the frame stack: a fresh learner
    (no optimiser state yet) that loads the file holds the saver's networks, Adam moments / step and counters, and makes the
    same next update on the same batch."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from learner_doubles import ToyEnv, TorchFrameStack
    E, D, k = 6, 12, 3

    def make(seed):
        return LR.DQNLearner(ToyEnv(E, D, rank=0, period=7), learning_rate=1e-2, buffer_size=E * 40, batch_size=32, gamma=0.9,
                             learning_starts=0, target_update_interval=E * 6, train_freq=4, gradient_steps=1, net_arch=(16, 8), n_stack=k,
                             total_timesteps=10**6, seed=seed, chunk_len=4, frame_stack_cls=TorchFrameStack)
    A = make(11)
    A.learn(total_timesteps=E * 4 * 7)
    assert A.n_updates >= 5
    path = str(tmp_path / "dqn.pt")
    A.save(path)
    ck = torch.load(path, weights_only=True)
    assert ck["optimizer_step"] == A.n_updates and set(ck["exp_avg"]) == set(dict(A.q.named_parameters()))
    B = make(5)
    assert not B.opt.state                                   # Adam's state does not exist before a step: load creates it
    B.load(path)
    for a, b in zip(list(A.q.parameters()) + list(A.q_target.parameters()), list(B.q.parameters()) + list(B.q_target.parameters())):
        assert torch.equal(a, b)
    ma, sa = A._moments(); mb, sb = B._moments()
    assert sa == sb == A.n_updates
    for name in ma:
        assert torch.equal(ma[name][0], mb[name][0]) and torch.equal(ma[name][1], mb[name][1]), name
    assert (B.num_timesteps, B.n_calls, B.n_updates) == (A.num_timesteps, A.n_calls, A.n_updates)
    assert torch.equal(A.gen.get_state(), B.gen.get_state())
    batch = A._sample()
    for L in (A, B):
        L._set_lr(1e-2)
        L._backward(batch); L._apply()
    for a, b in zip(A.q.parameters(), B.q.parameters()):
        assert torch.equal(a, b)
