"""the packaged DQN learner (uavenv_amd/learner.py) -- the reference's trainer loop
(agents/dqn/dqn.py:1077-1099, :1276-1288, :1324: SB3 DQN over VecFrameStack(DummyVecEnv)) on device.

stable-baselines3 is not installed in the build image, so a run of SB3 cannot pin this row ("parity unpinned"); what is
checked here: (a) the arithmetic of one update -- TD target, smooth-L1, masking, gradient clipping, the Adam step, the
target-network cadence, the schedules -- against formulas written out by hand; (b) that a short run actually learns.
"""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mods():
    import torch
    import uavenv_amd as U
    from uavenv_amd import learner as LR
    return torch, U, LR


def _torch_adam_twin(torch, L, q0):
    """A torch.optim.Adam over the copy `q0` of the learner's online network, in the learner's optimiser state -- whichever form
    the learner keeps it in (torch's own, or the fused update's flat moment buffers and step count)."""
    import copy
    opt0 = torch.optim.Adam(q0.parameters(), lr=torch.tensor(0.0, device=L.dev), capturable=True)
    if L._mlp is None:
        opt0.load_state_dict(copy.deepcopy(L.opt.state_dict()))
        return opt0
    # the library's flat buffers hold the layers after the extractor first, then ("hybrid") the extractor's parameters
    step = float(L._mlp.step_count)
    order = (list(q0.head.parameters()) + list(q0.features.parameters())) if L._hybrid else list(q0.parameters())
    off = 0
    for p in order:
        n = p.numel()
        opt0.state[p] = {"step": torch.tensor(step, device=L.dev), "exp_avg": L._mlp.exp_avg[off:off + n].view_as(p).clone(),
                         "exp_avg_sq": L._mlp.exp_avg_sq[off:off + n].view_as(p).clone()}
        off += n
    assert off == L._mlp.n_params
    return opt0


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


@pytest.mark.parametrize("fused", [True, False], ids=["hip_update", "torch_update"])
def test_one_update_matches_hand_written_arithmetic(fused):
    """TD target / smooth-L1 / valid mask / clipping / Adam on a fixed batch drawn with `sample_stacked` -- through the library's
    own update kernels (the default for the MLP policy) and through torch autograd + torch.optim.Adam."""
    torch, U, LR = _mods()
    E, k, gamma = 96, 3, 0.9
    env = U.BatchedUAVEnv(E, num_sensors=10, max_steps=9, grid_size=(60, 60), seed=5)
    L = LR.DQNLearner(env, learning_rate=1e-2, buffer_size=96 * 40, batch_size=64, gamma=gamma, learning_starts=0,
                      target_update_interval=96 * 1000, train_freq=1, gradient_steps=1, net_arch=(32,), n_stack=k,
                      total_timesteps=10**6, max_grad_norm=0.5, seed=3, reward_scale=1e-3, fused_update=fused)
    assert (L._mlp is not None) == fused
    L.collect(30)
    with torch.no_grad():                     # make the target network differ from the online one
        for p in L.q_target.parameters():
            p.mul_(0.5)
    gen_state = L.gen.get_state()
    if fused:          # the library's update draws inside the gather kernel, keyed by (seed, optimiser step): train(1) below repeats it
        batch = {k_: (v.clone() if torch.is_tensor(v) else tuple(t.clone() for t in v)) for k_, v in L._sample().items()}
    else:
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
    the trained network earns at least 1.1 x the mean episode return of the uniform-random policy on 256 held-out
    environments.  The learning rate decays linearly to 0 and gradients are clipped at 1.0: with a constant rate the final
    policy is whatever the last updates left (tools/learner_seed_spread.py: -0.9 x ... 1.4 x over seeds 0-3 x eager / graph
    replays), with the decay all eight runs end at 1.26-1.53 x.  The rewards are scaled by 1e-4 in the loss (see
    DQNLearner.reward_scale).  On this small grid every sensor is in radio range from everywhere, so the task is WHEN to
    collect, not where to fly; it shows the loop learns, not that it solves the reference's 500 x 500 task.  The loops run as
    graph replays here (use_graphs defaults to on for one process on a GPU)."""
    torch, U, LR = _mods()
    kw = dict(num_sensors=5, grid_size=(20, 20), max_steps=80)
    env = U.BatchedUAVEnv(256, seed=1, **kw)
    held_out = U.BatchedUAVEnv(256, seed=99, **kw)
    L = LR.DQNLearner(env, learning_rate=lambda progress_remaining: 1e-3 * progress_remaining, buffer_size=100_000,
                      learning_starts=2_000, target_update_interval=2_000, train_freq=1, gradient_steps=4, net_arch=(128, 128),
                      n_stack=2, total_timesteps=300_000, exploration_fraction=0.5, reward_scale=1e-4, max_grad_norm=1.0, seed=1)
    random_return, n = L.evaluate(held_out, 1, "random")
    assert n == 256 and random_return > 0
    L.learn()
    assert L._act_graphs is not None and L._train_graph is not None
    assert L.n_updates > 4000 and np.isfinite(float(L.last_loss.detach()))
    greedy_return, n = L.evaluate(held_out, 1, "greedy")
    assert n == 256
    assert greedy_return >= 1.1 * random_return, (greedy_return, random_return)
    st = env.episode_stats()
    assert (st["valid"] == 1).all() and (st["length"] == 80).all()              # every training environment finished episodes
    env.close(); held_out.close()


def test_graph_replay_of_the_acting_loop_writes_what_the_eager_loop_writes():
    """use_graphs: one captured graph per ring slot (Q forward, epsilon-greedy choice, environment step into the slot, frame
    stack, target copy) against the same steps launched one by one, from the same seeds.  Half of the actions are random
    (epsilon = 0.5: the graphs draw from the learner's generator exactly as the eager calls do), the other half greedy; the
    output layer is given zero weights and well separated biases so that the greedy action cannot hinge on the rounding of a
    GEMM.  Both learners must then fill their rings with the same transitions and leave the environments in the same state.
    (Terminal ROWS are claimed with an atomic counter, so their order inside a chunk's section is compared through the tickets.)"""
    torch, U, LR = _mods()
    from uavenv_amd import _native as N
    kw = dict(num_sensors=10, grid_size=(40, 40), max_steps=25, seed=4)
    hp = dict(learning_rate=1e-3, buffer_size=64 * 30, batch_size=32, learning_starts=10**9, target_update_interval=64,
              train_freq=4, gradient_steps=1, net_arch=(64, 64), n_stack=3, total_timesteps=10**6, seed=9,
              exploration_initial_eps=0.5, exploration_final_eps=0.5)
    envs = [U.BatchedUAVEnv(64, **kw) for _ in range(2)]
    Lg, Le = LR.DQNLearner(envs[0], use_graphs=True, **hp), LR.DQNLearner(envs[1], use_graphs=False, **hp)
    assert Lg.target_every == 1 and Lg.ring.capacity == Le.ring.capacity
    for L in (Lg, Le):
        with torch.no_grad():
            L.q.head[-1].weight.zero_()
            L.q.head[-1].bias.copy_(torch.tensor([0.3, 0.1, 0.5, 0.2, 0.4], device=L.dev))     # greedy = action 2
        L.collect(70)                              # more than two revolutions of the ring, several auto-resets (25-step episodes)
    assert Lg._act_graphs is not None and Le._act_graphs is None and len(Lg._act_graphs) == Lg.ring.capacity
    assert (Lg.n_calls, Lg.num_timesteps, Lg.ring.head, Lg.ring.size) == (Le.n_calls, Le.num_timesteps, Le.ring.head, Le.ring.size)
    torch.cuda.synchronize()
    rg, re_ = Lg.ring, Le.ring
    assert torch.equal(rg._obs5, re_._obs5)
    assert torch.equal(rg._aux5[..., :3], re_._aux5[..., :3])                      # action, reward, done of every transition
    acts = rg._aux5[..., 0].flatten()
    counts = torch.bincount(acts[acts >= 0].long(), minlength=5)
    assert int(counts[2]) > int(counts.sum()) // 2 and int((counts > 0).sum()) == 5   # greedy 2 half of the time + all random ones
    done = rg._aux5[..., 2] > 0.5
    assert int(done.sum()) >= 64                                                   # episode ends were recorded ...
    tg, te = rg._ticket5[done], re_._ticket5[done]
    assert bool((tg >= 0).all()) and bool((te >= 0).all())
    c_idx = torch.nonzero(done)[:, 0]                                              # ... with the same terminal observations
    assert torch.equal(rg._term4[c_idx, 0, tg.long() % rg.T], re_._term4[c_idx, 0, te.long() % re_.T])
    assert torch.equal(Lg.fs.stacked, Le.fs.stacked)
    for f in (N.F_BUFFER, N.F_GEN, N.F_TX, N.F_LOST, N.F_FLAGS, N.F_RECORD):
        assert torch.equal(envs[0].get_state(f), envs[1].get_state(f))
    for p, t in zip(Lg.q.parameters(), Lg.q_target.parameters()):                  # target_every == 1: copied inside the graphs
        assert torch.equal(p, t)
    for e in envs:
        e.close()


@pytest.mark.parametrize("fused", [True, False], ids=["hip_update", "torch_update"])
def test_graph_replay_of_the_update_is_the_eager_update_on_the_same_draw(fused):
    """The captured gradient step (sample -> TD loss -> backward -> clip -> Adam) against the same step done by hand in
    eager mode on the transitions the replay drew (`_g_index`), from the same weights and optimiser state."""
    import copy
    torch, U, LR = _mods()
    env = U.BatchedUAVEnv(96, num_sensors=10, max_steps=9, grid_size=(60, 60), seed=5)
    L = LR.DQNLearner(env, learning_rate=1e-2, buffer_size=96 * 40, batch_size=64, gamma=0.9, learning_starts=0,
                      target_update_interval=96 * 7, train_freq=2, gradient_steps=1, net_arch=(32, 16), n_stack=3,
                      total_timesteps=10**6, max_grad_norm=0.5, seed=3, reward_scale=1e-3, use_graphs=True, fused_update=fused)
    L.learn(total_timesteps=96 * 2 * 8)            # 8 rollouts: the first 3 updates are eager, then the graph is captured
    assert L._train_graph is not None and L.n_updates >= 5
    q0, t0 = copy.deepcopy(L.q), copy.deepcopy(L.q_target)
    opt0 = _torch_adam_twin(torch, L, q0)
    before = [p.detach().clone() for p in L.q.parameters()]
    L.train(1)                                     # one replay
    torch.cuda.synchronize()
    assert any(not torch.equal(a, b) for a, b in zip(before, [p.detach() for p in L.q.parameters()]))
    j, slot, r, e = [L._g_index[i] for i in range(4)]
    n, oldest = L.ring.window_state()
    assert int(j.min()) >= 0 and int(j.max()) <= n - 2 and torch.equal(slot, (oldest + j) % L.ring.capacity)
    assert int(e.min()) >= 0 and int(e.max()) < 96 and len(torch.unique(e)) > 20 and int(r.max()) == 0
    batch = L.ring.stacked_batch_at(j, slot, r, e, L.k)
    for g in opt0.param_groups:
        g["lr"].fill_(L.lr_schedule(L.progress_remaining()))
    loss = LR.td_loss(q0, t0, batch, 0.9, 1e-3)
    opt0.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(q0.parameters(), 0.5)
    opt0.step()
    assert float(loss.detach()) == pytest.approx(float(L.last_loss), rel=1e-5)
    tol = dict(rtol=1e-4, atol=1e-6) if fused else dict(rtol=1e-5, atol=1e-7)      # (split-K partial sums in another order)
    for p, w in zip(L.q.parameters(), q0.parameters()):
        assert torch.allclose(p.detach(), w.detach(), **tol), float((p.detach() - w.detach()).abs().max())
    env.close()


def test_attention_learner_acts_through_the_fused_kernel_with_current_weights():
    """extractor="attention" under graph replay: acting uses the fused inference kernel, whose weight block is re-packed at the
    end of every captured update -- after training, its features must be those of the (updated) PyTorch module."""
    torch, U, LR = _mods()
    env = U.BatchedUAVEnv(128, num_sensors=50, grid_size=(100, 100), max_steps=40, seed=2)
    L = LR.DQNLearner(env, learning_rate=1e-3, buffer_size=128 * 24, batch_size=64, learning_starts=0, target_update_interval=128 * 3,
                      train_freq=2, gradient_steps=1, net_arch=(64,), n_stack=4, total_timesteps=10**6, extractor="attention", seed=4,
                      reward_scale=1e-4)
    w0 = None
    L.learn(total_timesteps=128 * 2 * 6)
    assert L._fused is not None and L._act_graphs is not None and L._train_graph is not None
    w0 = L._fused.weights.clone()
    L.learn(total_timesteps=128 * 2 * 12)          # six more graph updates
    assert not torch.equal(w0, L._fused.weights)
    with torch.no_grad():
        want = L.q.features(L.fs.stacked)
    got = L._fused(L.fs.stacked)
    assert torch.allclose(got, want, rtol=2e-4, atol=2e-5), float((got - want).abs().max())
    env.close()


# ---------------------------------------------------------------------------------------------------------------
# several ranks on ONE GPU (gloo): the real environment, HIP graphs and the chunk exchange together
# ---------------------------------------------------------------------------------------------------------------
def _two_rank_gpu_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import uavenv_amd as U
    from uavenv_amd import learner as LR
    torch.cuda.set_device(0)
    E = 64
    bad = []

    def check(name, cond):
        if not cond:
            bad.append(name)
    env = U.BatchedUAVEnv(E, env_index_base=rank * E, num_sensors=10, grid_size=(40, 40), max_steps=25, seed=4)
    L = LR.DQNLearner(env, learning_rate=1e-3, buffer_size=2 * E * 48, batch_size=64, learning_starts=0, target_update_interval=2 * E * 5,
                      train_freq=4, gradient_steps=1, net_arch=(64, 64), n_stack=3, total_timesteps=10**6, seed=9, chunk_len=8,
                      reward_scale=1e-3)
    check("setup", L.world == 2 and L.local_batch == 32 and L.ring.exchange and L._graphs_usable())
    L.learn(total_timesteps=2 * E * 4 * 30)                  # 120 vector steps: > 2 revolutions of the 56-slot ring, 4 auto-resets per env
    torch.cuda.synchronize()
    check("graphs", L._act_graphs is not None and L._train_graph is not None and L._train_graph_b is not None)
    check(f"updates {L.n_updates}", L.n_updates >= 25)
    # replicas identical
    flat = torch.cat([p.detach().reshape(-1) for p in L.q.parameters()]).cpu()
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    check(f"replicas differ by {float((both[0] - both[1]).abs().max())}", torch.equal(both[0], both[1]))
    check("finite", bool(torch.isfinite(flat).all()))
    # every gathered chunk part is the same on both ranks: sums of the (chunk, rank) parts, head chunk excluded
    L.ring.drain()
    torch.cuda.synchronize()
    sums = L.ring._i32.long().sum(-1).cpu()                 # [chunk][rank], over the raw words (ticket -1 is a NaN as float)
    head_chunk = L.ring.head // L.ring.L
    keep = [c for c in range(L.ring.n_chunks) if c != head_chunk]
    views = [torch.zeros_like(sums) for _ in range(world)]
    dist.all_gather(views, sums)
    check("rings equal", torch.equal(views[0][keep], views[1][keep]))
    check("rings filled", bool((sums[keep] != 0).all()))
    # episode ends of the OTHER rank are valid transitions here (terminal rows travelled with the chunks)
    b = L.ring.sample_stacked(4000, 3, generator=torch.Generator(device="cuda").manual_seed(1))
    j, slot, r, e = b["index"]
    remote_end = (r == 1 - rank) & b["done"]
    check(f"remote ends {int(remote_end.sum())}", int(remote_end.sum()) > 10)
    check("remote ends valid", bool(b["valid"][remote_end].all()))
    dist.barrier()
    dist.destroy_process_group()
    env.close()
    q.put((rank, bad))


def test_two_rank_learner_on_one_gpu_graphs_coexist_with_the_exchange():
    """BASELINE config 4 as a training run, rehearsed with two ranks on this one GPU over gloo: per-slot act graphs and the
    two-graph update are replayed while the chunk all-gathers and the flat gradient all-reduce are issued from the host between
    them; the replicas must end with bit-identical weights and identical rings."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_two_rank_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, []), (1, [])], res


@pytest.mark.parametrize("extractor,fused_a,fused_b", [("mlp", True, True), ("mlp", True, False), ("mlp", False, True), ("attention", True, True),
                                                       ("attention", False, True)])
def test_checkpoint_round_trip_between_learners_and_update_paths(tmp_path, extractor, fused_a, fused_b):
    """DQNLearner.save / load (the reference's model.save / DQN.load): a learner that trained for a while, saved, and a FRESH
    learner that loads the file -- possibly with the other update path (library kernels <-> PyTorch) -- hold the same networks,
    Adam state and counters, act identically on the same observations, and make the same next update on the same batch."""
    torch, U, LR = _mods()
    k = 4 if extractor == "mlp" else 3
    def make(fused, seed):
        env = U.BatchedUAVEnv(96, num_sensors=50 if extractor == "attention" else 10, max_steps=9, grid_size=(60, 60), seed=5)
        return env, LR.DQNLearner(env, learning_rate=1e-3, buffer_size=96 * 40, batch_size=64, gamma=0.9, learning_starts=0,
                                  target_update_interval=96 * 7, train_freq=2, gradient_steps=1, net_arch=(32, 16), n_stack=k,
                                  total_timesteps=10**6, max_grad_norm=0.5, seed=seed, reward_scale=1e-3, extractor=extractor,
                                  fused_update=fused, use_graphs=False)
    env_a, A = make(fused_a, 3)
    for _ in range(12):
        A.collect(A.train_freq); A.train()
    path = str(tmp_path / "dqn.pt")
    A.save(path)
    ck = torch.load(path, weights_only=True)                            # plain tensors and numbers only
    assert ck["n_updates"] == A.n_updates > 5 and ck["optimizer_step"] == A.n_updates
    env_b, B = make(fused_b, 99)                                         # other initial weights, other generator
    assert any(not torch.equal(a, b) for a, b in zip(A.q.parameters(), B.q.parameters()))
    B.load(path)
    for a, b in zip(list(A.q.parameters()) + list(A.q_target.parameters()), list(B.q.parameters()) + list(B.q_target.parameters())):
        assert torch.equal(a, b)
    ma, sa = A._moments(); mb, sb = B._moments()
    assert sa == sb == A.n_updates and set(ma) == set(mb)
    for name in ma:
        assert torch.equal(ma[name][0], mb[name][0]) and torch.equal(ma[name][1], mb[name][1]), name
    assert (B.num_timesteps, B.n_calls, B.n_updates) == (A.num_timesteps, A.n_calls, A.n_updates)
    assert B.exploration_rate() == A.exploration_rate() and torch.equal(A.gen.get_state(), B.gen.get_state())
    x = A._stacked.clone()
    with torch.no_grad():                                               # (same weights at other addresses: the GEMM library may pick another kernel)
        assert torch.allclose(A.q(x), B.q(x), rtol=1e-5, atol=1e-6)
    # the same next update on the same batch (A's ring), each with its own update path
    batch = {k_: (v.clone() if torch.is_tensor(v) else v) for k_, v in A._sample().items()}
    for L in (A, B):
        L._set_lr(1e-3)
        L._backward(batch); L._apply()
    lr = 1e-3
    for (name, a), b in zip(A.q.named_parameters(), B.q.parameters()):
        d = float((a.detach() - b.detach()).abs().max())
        assert d <= (1e-6 if fused_a == fused_b else 2.0 * lr), (name, d)
    with torch.no_grad():
        qa, qb = A.q(x), B.q(x)
    assert torch.allclose(qa, qb, rtol=1e-3, atol=1e-3 * float(qa.abs().max()) + 1e-6)
    env_a.close(); env_b.close()


def test_acting_forward_with_fused_bias_relu_epilogues_equals_the_module():
    """QNetwork.head_inference (Linear + ReLU pairs as one GEMM with the bias + ReLU epilogue) against the module it replaces for
    acting, at the reference's widths and at 4096 rows: the same Q-values (to fp32 summation order) and the same greedy actions."""
    torch, U, LR = _mods()
    torch.manual_seed(1)
    q = LR.QNetwork(153, 4).cuda()
    x = torch.rand(4096, 612, device="cuda")
    with torch.no_grad():
        want = q(x)
    got = q.q_inference(x)
    assert got.shape == want.shape == (4096, 5)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), float((got - want).abs().max())
    top2 = want.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert int(clear.sum()) > 3000 and torch.equal(got.argmax(1)[clear], want.argmax(1)[clear])
    q1 = LR.QNetwork(7, 3, (16,)).cuda()                                 # one hidden layer, odd widths
    x1 = torch.randn(33, 21, device="cuda")
    with torch.no_grad():
        assert torch.allclose(q1.q_inference(x1), q1(x1), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shared", [False, True], ids=["own_coins", "shared_coin"])
def test_output_layer_and_selection_in_one_launch_equal_the_two_step_form(shared):
    """uavenv_q_head_select (last Linear + epsilon-greedy in one launch, what acting runs) against the library GEMM followed by
    uavenv_epsilon_greedy at the same draw counter: the same coins and random actions per environment, the same greedy actions where
    the two summation orders cannot flip the argmax; both advance the counter by one; 300 environments (not a multiple of 64)."""
    torch, U, LR = _mods()
    from uavenv_amd import _native as N
    import ctypes as C
    env = U.BatchedUAVEnv(300, num_sensors=10, seed=3)
    L = LR.DQNLearner(env, learning_rate=1e-3, buffer_size=300 * 20, batch_size=64, learning_starts=0, net_arch=(64, 256), n_stack=4,
                      total_timesteps=10**6, seed=4, shared_exploration_coin=shared, use_graphs=False)
    L.collect(3)
    x = L._stacked.clone()
    eps = torch.tensor(0.35, device=L.dev)
    for c0 in (7.0, 8.0):
        L._act_counter.fill_(c0)
        with torch.no_grad():
            q = L.q.q_inference(x)
        a_two = L._select_actions(q, eps).clone()
        assert float(L._act_counter) == c0 + 1
        L._act_counter.fill_(c0)
        a_one = L._act_device(x, eps).clone()
        assert float(L._act_counter) == c0 + 1 and int(L._act_ticket) == 0
        top2 = q.topk(2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-5
        assert int(clear.sum()) > 250 and torch.equal(a_one[clear], a_two[clear])
        greedy = q.argmax(1).to(torch.int32)
        explored = a_two != greedy
        assert 40 < int(explored.sum()) < 150 or shared          # ~0.35 * 4/5 of 300 take another action (one coin: all or none)
    # the Q-values it computes, on request
    qo = torch.zeros(300, 5, device=L.dev)
    last = L.q.head[-1]
    h = L.q.head_inference(L.q.features(x), upto_last=True).contiguous()
    rc = N.lib().uavenv_q_head_select(C.c_void_p(h.data_ptr()), C.c_void_p(last.weight.data_ptr()), C.c_void_p(last.bias.data_ptr()), 300, 256, 5,
                                      C.c_void_p(eps.data_ptr()), C.c_void_p(L._act_counter.data_ptr()), C.c_void_p(L._act_ticket.data_ptr()), 1, 0,
                                      C.c_void_p(L._act_out.data_ptr()), C.c_void_p(qo.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.allclose(qo, q, rtol=1e-5, atol=1e-6)
    # bad arguments
    assert N.lib().uavenv_q_head_select(C.c_void_p(h.data_ptr()), C.c_void_p(last.weight.data_ptr()), C.c_void_p(last.bias.data_ptr()), 300, 256, 9,
                                        C.c_void_p(eps.data_ptr()), C.c_void_p(L._act_counter.data_ptr()), C.c_void_p(L._act_ticket.data_ptr()), 1, 0,
                                        C.c_void_p(L._act_out.data_ptr()), None, None) == N.E_INVALID
    env.close()


def test_updates_with_autograd_in_them_are_not_graph_replayed_above_batch_256():
    """Reductions that autograd spreads over several workgroups replay stale from HIP graphs on this stack (tools/graph_grad_check.py
    at BATCH=512): with the attention extractor or the PyTorch update, a batch above 256 keeps the UPDATE eager (acting is still
    replayed); the library's own MLP update is replayed at any batch."""
    torch, U, LR = _mods()
    def run(**kw):
        env = U.BatchedUAVEnv(128, num_sensors=50, max_steps=9, grid_size=(60, 60), seed=5)
        L = LR.DQNLearner(env, learning_rate=1e-3, buffer_size=128 * 40, gamma=0.9, learning_starts=0, target_update_interval=128 * 7,
                          train_freq=2, gradient_steps=1, net_arch=(32, 16), total_timesteps=10**6, seed=3, reward_scale=1e-3, **kw)
        for _ in range(8):
            L.collect(L.train_freq); L.train()
        torch.cuda.synchronize()
        out = (L._act_graphs is not None, L._train_graph is not None, L.n_updates)
        env.close()
        return out
    assert run(extractor="attention", n_stack=3, batch_size=512) == (True, False, 8)
    assert run(extractor="attention", n_stack=3, batch_size=256) == (True, True, 8)
    assert run(extractor="mlp", n_stack=4, batch_size=512, fused_update=False) == (True, False, 8)
    assert run(extractor="mlp", n_stack=4, batch_size=512) == (True, True, 8)
