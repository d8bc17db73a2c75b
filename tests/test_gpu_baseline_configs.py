"""BASELINE.json's configurations at their STATED sizes, as far as one GPU holds them:

  C2  4096 envs x N=20 on 500 x 500, random policy: the first 256 environments x 500 steps against the CPU oracle
  C3  4096 envs x N=50 driving the DQN learner (MLP with n_stack 4 / attention extractor with n_stack 10) through the
      HIP-graph path, with gradient updates running
  C5  the single-GPU slice of the domain-randomised sweep: 8192 environments, the nine (grid, N) combinations interleaved
      in ONE handle, 18 of them checked against their own oracle instances
  +   the episode statistics of 256 domain-randomised environments against the oracle's restatement of dqn.py:305-331

(C4 and the 8-GPU split of C5 need more than one GPU: tests/test_host_logic.py covers their exchange over gloo.)
Every test stays well below 20 s on an MI355X.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OBS_ATOL = 1e-6
REW_RTOL = 1e-9


def _mods():
    import torch
    import uavenv_amd as U
    from oracle import oracle as O
    return torch, U, O


def _rel(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


def test_config2_4096x20_first_256_envs_500_steps_vs_oracle():
    torch, U, O = _mods()
    E, Echk, n, steps, K = 4096, 256, 20, 500, 50
    env = U.BatchedUAVEnv(E, num_sensors=n, grid_size=(500, 500), seed=0)
    assert env.obs_dim == 63 and env.lane_stride == 32
    want = O.trace_keyed(O.default_config(num_sensors=n, grid_size=(500, 500), seed=0), Echk, steps)
    assert np.array_equal(env.reset().cpu().numpy()[:Echk], want["reset_obs"])
    max_obs = max_rew = 0.0
    for s0 in range(0, steps, K):
        if s0 % 100 == 0:                         # alternate the two launch forms: K single-step launches / one fused rollout
            o = np.empty((K, Echk, 63), np.float32); r = np.empty((K, Echk)); d = np.empty((K, Echk), np.uint8)
            a = np.empty((K, Echk), np.int32)
            for k in range(K):
                ot, rt, dt = env.step_random()
                o[k], r[k], d[k] = ot[:Echk].cpu().numpy(), rt[:Echk].cpu().numpy(), dt[:Echk].cpu().numpy()
                a[k] = env.actions_taken[:Echk].cpu().numpy()
        else:
            ro = env.rollout(K)
            o, r, d, a = (ro[k][:, :Echk].cpu().numpy() for k in ("obs", "reward", "done", "actions"))
        sl = slice(s0, s0 + K)
        assert np.array_equal(a, want["actions"][sl]) and np.array_equal(d, want["done"][sl]), s0
        max_obs = max(max_obs, float(np.max(np.abs(o - want["obs"][sl]))))
        max_rew = max(max_rew, float(np.max(_rel(r, want["reward"][sl]))))
    assert max_obs <= OBS_ATOL and max_rew <= REW_RTOL, (max_obs, max_rew)
    # and the whole batch: the invariants that need no oracle
    ss = env.sensor_state()
    assert np.allclose(ss["buffer"] + ss["tx"] + ss["lost"], ss["gen"], rtol=1e-12, atol=1e-9)
    obs = env.obs.cpu().numpy()
    assert np.isfinite(obs).all() and obs.min() >= -1.0 and obs.max() <= 1.0
    env.close()


def _oracle_replay_of_ring(O, ring, cfg_over, envs, steps):
    """Drive oracle instances with the actions the ring recorded and compare every stored transition."""
    for e in envs:
        orc = O.OracleEnv(O.default_config(**cfg_over), e)
        assert np.array_equal(ring.obs_at(0, 0, e).cpu().numpy(), orc.reset_keyed()), e
        for s in range(1, steps + 1):
            aux = ring.aux_at(s, 0, e).cpu().numpy()
            oo, rr, tr = orc.step_keyed(int(aux[0]))
            assert not tr and aux[2] == 0.0
            assert np.max(np.abs(ring.obs_at(s, 0, e).cpu().numpy() - oo)) <= OBS_ATOL, (e, s)
            assert abs(aux[1] - rr) <= 1e-6 * max(1.0, abs(rr)), (e, s, aux[1], rr)          # float32 reward in the aux row


@pytest.mark.parametrize("extractor,n_stack", [("mlp", 4), ("attention", 10)])
def test_config3_learner_4096x50_graph_path_equals_eager_and_oracle(extractor, n_stack):
    """The reference's hyper-parameters (dqn.py:1077-1099) at 4096 environments x 50 sensors: 40 vector steps = 10 rollouts with
    an update after each (learning_starts 0), the first three eager, the rest replayed as HIP graphs.  Exploration is held at
    1.0 so that the graph learner and an eager twin take the same (generator-drawn) actions whatever the weights do."""
    torch, U, O = _mods()
    from uavenv_amd import learner as LR
    E, steps = 4096, 40
    over = dict(num_sensors=50, grid_size=(500, 500), seed=21)
    hp = dict(LR.REFERENCE_HYPERPARAMS)
    hp.update(n_stack=n_stack, learning_starts=0, exploration_initial_eps=1.0, exploration_final_eps=1.0, extractor=extractor,
              seed=5, reward_scale=1e-3)
    envs = [U.BatchedUAVEnv(E, **over) for _ in range(2)]
    Lg, Le = LR.DQNLearner(envs[0], use_graphs=True, **hp), LR.DQNLearner(envs[1], use_graphs=False, **hp)
    assert Lg.ring.capacity == 45 and Lg.ring.L == 9 and Lg.target_every == 1 and Lg.train_freq == 4
    init = [p.detach().clone() for p in Lg.q.parameters()]
    for L in (Lg, Le):
        L.learn(total_timesteps=E * steps)
    torch.cuda.synchronize()
    assert Lg._act_graphs is not None and Lg._train_graph is not None and Le._act_graphs is None
    # the layers after the extractor (all of the MLP policy) are updated by the library's own kernels; the attention extractor
    # itself by autograd + torch Adam, clipped with the same coefficient
    assert Lg._mlp is not None and Lg._hybrid == (extractor == "attention")
    # (a rollout is followed by an update once the ring holds n_stack + 2 slots: 9 of the 10 rollouts with 4 frames, 8 with 10)
    assert Lg.n_updates == Le.n_updates == (9 if n_stack == 4 else 8) and Lg.n_calls == Le.n_calls == steps
    rg, re_ = Lg.ring, Le.ring
    assert (rg.head, rg.size) == (re_.head, re_.size) == (steps + 1, steps + 1)
    assert torch.equal(rg._obs5, re_._obs5) and torch.equal(rg._aux5[..., :3], re_._aux5[..., :3])
    acts = rg._aux5[..., 0].flatten()[: (steps + 1) * E]
    assert int(torch.bincount(acts.long(), minlength=5).min()) > steps * E // 8              # all five actions, evenly
    assert torch.equal(Lg.fs.stacked, Le.fs.stacked)
    assert all(not torch.equal(p, p0) for p, p0 in zip(Lg.q.parameters(), init))
    # (the two learners draw their batches differently -- integer draws vs the graph's device-side window arithmetic -- so
    #  their weights are not comparable; instead: one more captured update against the same update done by hand, eagerly, on
    #  the transitions the replay drew)
    import copy
    q0, t0 = copy.deepcopy(Lg.q), copy.deepcopy(Lg.q_target)
    from test_gpu_learner import _torch_adam_twin
    opt0 = _torch_adam_twin(torch, Lg, q0)
    Lg.train(1)
    torch.cuda.synchronize()
    j, slot, r, e = [Lg._g_index[i] for i in range(4)]
    assert int(e.max()) < E and len(torch.unique(e)) > 200 and int(j.max()) <= Lg.ring.sampleable() - 2
    batch = Lg.ring.stacked_batch_at(j, slot, r, e, Lg.k)
    assert batch["obs"].shape == (256, n_stack * 153) and bool(batch["valid"].all())
    for g in opt0.param_groups:
        g["lr"].fill_(Lg.lr_schedule(Lg.progress_remaining()))
    loss = LR.td_loss(q0, t0, batch, Lg.gamma, Lg.reward_scale)
    opt0.zero_grad(set_to_none=True)
    loss.backward()
    raw = [w.grad.detach().clone() for w in q0.parameters()]
    torch.nn.utils.clip_grad_norm_(q0.parameters(), Lg.max_grad_norm)
    opt0.step()
    assert float(loss.detach()) == pytest.approx(float(Lg.last_loss), rel=1e-4)
    # (1) the GRADIENTS of the captured update against the hand-made ones: the well-conditioned comparison, possible where the update
    #     keeps them in a buffer of its own (the library's update; the PyTorch update's gradient tensors live in the graph's pool and
    #     are recycled by later nodes of the same graph)
    got = [t for pair in zip(Lg._mlp.gw, Lg._mlp.gb) for t in pair]
    names = [n_ for n_, _ in Lg.q.named_parameters()]
    raw_head = raw[len(raw) - len(got):]                         # (the head's parameters come last; the MLP policy is all head)
    gmax = max(float(w.abs().max()) for w in raw)
    for name, g_, w_ in zip(names[len(raw) - len(got):], got, raw_head):
        assert float((g_ - w_).abs().max()) <= 2e-4 * max(float(w_.abs().max()), 1e-3 * gmax), (name, float((g_ - w_).abs().max()))
    # (the total norm the clipping used: head + extractor)
    want_norm2 = float(sum((w.double() ** 2).sum() for w in raw))
    from uavenv_amd import _native as N
    assert float(Lg._mlp.scalars[N.UPD_NORM2]) == pytest.approx(want_norm2, rel=2e-3)
    # (2) the PARAMETERS after Adam: its step is lr * m / (sqrt(v) + 1e-8), and where gradients are cancellation noise (the attention's
    #     in-projection biases: 1e-7 out of summands of 1e-3; the key bias is exactly zero in theory) last-bit differences between graph
    #     replay and eager launches become visible fractions of lr.  Every element must stay within a step; the well-conditioned
    #     statement is about what the network computes: the two updated networks agree on the Q-values of the batch.
    lr_now = Lg.lr_schedule(Lg.progress_remaining())
    for (name, p), w in zip(Lg.q.named_parameters(), q0.parameters()):
        diff = (p.detach() - w.detach()).abs()
        assert float(diff.max()) <= 2.0 * lr_now, (name, float(diff.max()), lr_now)
    with torch.no_grad():
        qa, qb = Lg.q(batch["obs"]), q0(batch["obs"])
    assert torch.allclose(qa, qb, rtol=1e-3, atol=1e-3 * float(qb.abs().max())), float((qa - qb).abs().max())
    _oracle_replay_of_ring(O, rg, over, [0, 1, 63, 64, 1000, 2047, 2048, 4095], steps)
    if extractor == "attention":
        assert Lg._fused is not None
        with torch.no_grad():
            q_mod = Lg.q(Lg.fs.stacked)
            q_fus = Lg.q.head(Lg._fused(Lg.fs.stacked))
        assert torch.allclose(q_fus, q_mod, rtol=1e-3, atol=1e-4), float((q_fus - q_mod).abs().max())
        top2 = q_mod.topk(2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-3
        assert int(clear.sum()) > E // 2
        assert torch.equal(q_fus.argmax(1)[clear], q_mod.argmax(1)[clear])
    for e in envs:
        e.close()


def test_config5_slice_8192_envs_nine_combinations_in_one_handle():
    """Grid in {250, 500, 1000} x N in {10, 20, 50}, interleaved over the environments of one handle (uavenv_set_env_params);
    two environments of every combination against their own oracle instance, in-kernel random policy."""
    torch, U, O = _mods()
    E, steps, seed = 8192, 120, 77
    combos = [(g, n) for g in (250, 500, 1000) for n in (10, 20, 50)]
    grids = np.array([combos[k % 9][0] for k in range(E)], np.int32)
    ns = np.array([combos[k % 9][1] for k in range(E)], np.int32)
    rng = np.random.default_rng(5)
    pos = (rng.random((E, 50, 2), dtype=np.float32) * grids[:, None, None]).astype(np.float32)
    pos[np.arange(50)[None, :] >= ns[:, None]] = 0.0
    env = U.BatchedUAVEnv(E, num_sensors=50, seed=seed, sensor_positions=pos)
    env.set_env_params(grid_w=grids, grid_h=grids, num_sensors=ns)
    chk = list(range(9)) + [E - 9 + k for k in range(9)]                                      # both ends of the batch
    assert sorted(set((int(grids[k]), int(ns[k])) for k in chk)) == sorted(combos)
    orcs = {k: O.OracleEnv(O.default_config(num_sensors=int(ns[k]), pad_sensors=50, grid_size=(int(grids[k]),) * 2, seed=seed),
                           k, pos[k, :ns[k], 0], pos[k, :ns[k], 1]) for k in chk}
    o = env.reset().cpu().numpy()
    for k in chk:
        assert np.array_equal(o[k], orcs[k].reset_keyed()), k
    idx = torch.tensor(chk, device=env.device)
    for s in range(steps):
        ot, rt, dt = env.step_random()
        o, r, d, a = ot[idx].cpu().numpy(), rt[idx].cpu().numpy(), dt[idx].cpu().numpy(), env.actions_taken[idx].cpu().numpy()
        for j, k in enumerate(chk):
            assert a[j] == orcs[k].next_random_action(), (s, k)
            oo, rr, tr = orcs[k].step_keyed(int(a[j]))
            assert np.max(np.abs(o[j] - oo)) <= OBS_ATOL and _rel(r[j], rr) <= REW_RTOL and bool(d[j]) == tr, (s, k)
    rec = env.records()
    assert np.array_equal(rec["grid_w"], grids) and np.array_equal(rec["num_sensors"], ns)
    ss = env.sensor_state()
    live = np.arange(50)[None, :] < ns[:, None]
    assert np.allclose((ss["buffer"] + ss["tx"] + ss["lost"])[live], ss["gen"][live], rtol=1e-12, atol=1e-9)
    env.close()


@pytest.mark.parametrize("n", [10, 20, 40])
def test_episode_stats_of_256_domain_rand_envs_vs_oracle(n):
    """Every UavEnvEpisodeStats field the kernel writes at an episode end (dqn.py:305-331) against the oracle's
    orc_episode_stats, for 256 domain-randomised environments over several episodes (random policy, 35-step episodes)."""
    torch, U, O = _mods()
    E, steps, seed = 256, 110, 3
    over = dict(num_sensors=n, grid_size=(100, 100), grid_choices=[(100, 100), (200, 200), (300, 300)], pad_sensors=50, flags=15,
                max_steps=35, duty_cycle=60.0, seed=seed)
    env = U.BatchedUAVEnv(E, auto_reset=True, env_index_base=500, **over)
    orcs = [O.OracleEnv(O.default_config(**over), 500 + k) for k in range(E)]
    o = env.reset().cpu().numpy()
    for k in range(E):
        assert np.array_equal(o[k], orcs[k].reset_keyed())
    checked = 0
    for s in range(steps):
        ot, rt, dt = env.step_random()
        a, d = env.actions_taken.cpu().numpy(), dt.cpu().numpy()
        stats = env.episode_stats() if d.any() else None
        for k in range(E):
            _, _, tr = orcs[k].step_keyed(int(a[k]))
            assert tr == bool(d[k]), (s, k)
            if tr:
                w, g = orcs[k].episode_stats(), stats[k]
                for key in ("total_generated", "total_collected", "total_lost", "battery_remaining", "jains_index", "fairness_std"):
                    assert _rel(float(g[key]), w[key]) <= REW_RTOL, (s, k, key, g[key], w[key])
                assert (g["grid_w"], g["grid_h"]) == w["grid_size"] and g["num_sensors"] == n and g["length"] == w["length"] == 35
                assert g["sensors_visited"] == round(w["ndr"] * n / 100) and g["first_full_coverage_step"] == w["first_full_coverage_step"]
                orcs[k].reset_keyed()
                checked += 1
    assert checked == E * (steps // 35)
    env.close()


def test_set_config_on_a_live_handle_matches_the_oracle_at_the_new_constants():
    """sim_to_real_sweep.py:109-117 writes `shadowing_std_db` on the sensors of a LIVE environment: uavenv_set_config
    (BatchedUAVEnv.set_config / `env.sensors[i].shadowing_std_db = x`) re-derives the constants; the following steps must be
    the oracle's at the new value.  The default constants run the literal kernel variant, the changed ones the generic one."""
    torch, U, O = _mods()
    E, seed = 40, 12
    over = dict(num_sensors=20, grid_size=(120, 120), max_steps=50, duty_cycle=50.0, seed=seed)
    env = U.BatchedUAVEnv(E, **over)
    orcs = [O.OracleEnv(O.default_config(**over), k) for k in range(E)]
    o = env.reset().cpu().numpy()
    for k in range(E):
        assert np.array_equal(o[k], orcs[k].reset_keyed())

    def run(steps):
        for s in range(steps):
            ot, rt, dt = env.step_random()
            o, r, d, a = ot.cpu().numpy(), rt.cpu().numpy(), dt.cpu().numpy(), env.actions_taken.cpu().numpy()
            for k in range(E):
                oo, rr, tr = orcs[k].step_keyed(int(a[k]))
                if tr:
                    oo = orcs[k].reset_keyed()
                assert np.max(np.abs(o[k] - oo)) <= OBS_ATOL and _rel(r[k], rr) <= REW_RTOL and bool(d[k]) == tr, (s, k)

    run(30)
    env.set_config(shadowing_std_db=9.0, rssi_threshold=-80.0, penalty_hover=-7.0)
    assert env.cfg.shadowing_std_db == 9.0
    for orc in orcs:
        orc.e.cfg.shadowing_std_db, orc.e.cfg.rssi_threshold, orc.e.cfg.penalty_hover = 9.0, -80.0, -7.0
    run(45)
    env.set_config(shadowing_std_db=4.0, rssi_threshold=-85.0, penalty_hover=-5.0)          # back to the literal variant
    for orc in orcs:
        orc.e.cfg.shadowing_std_db, orc.e.cfg.rssi_threshold, orc.e.cfg.penalty_hover = 4.0, -85.0, -5.0
    run(30)
    with pytest.raises(U.UavEnvError):
        env.set_config(num_sensors=10)
    env.close()
    # the single-environment mirror: the attribute write the reference's sweep does
    g = U.UAVEnvironment(grid_size=(100, 100), num_sensors=5, seed=1)
    g.reset()
    assert g.sensors[0].shadowing_std_db == 4.0
    for s in g.sensors:
        s.shadowing_std_db = 0.0
        s.path_loss_exponent = 3.8
    assert g.sensors[3].shadowing_std_db == 0.0 and g._cfg.shadowing_std_db == 0.0
    ref = O.OracleEnv(O.default_config(grid_size=(100, 100), num_sensors=5, seed=1, shadowing_std_db=0.0), 0)
    ref.reset_keyed()            # (the reset observation above was drawn at sigma 4; the state it leaves is sigma-free except SF)
    for i in range(5):
        ref.e.sf[i] = int(g.sensors[i].spreading_factor)
        ref.e.avg_valid[i] = 0 if g.sensors[i].avg_rssi is None else 1
        ref.e.avg_rssi[i] = 0.0 if g.sensors[i].avg_rssi is None else g.sensors[i].avg_rssi
    for a in (4, 3, 0, 4, 4):
        obs, r, _, _, _ = g.step(a)
        oo, rr, _ = ref.step_keyed(a)
        assert np.max(np.abs(obs - oo)) <= OBS_ATOL and abs(r - rr) <= REW_RTOL * max(1.0, abs(rr))
    g.close()
