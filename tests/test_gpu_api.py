"""GPU: the host-side mirrors of the reference interfaces (gymnasium-style UAVEnvironment,
DomainRandEnv, SB3-style UAVVecEnv) against the oracle, written like tests the reference could hold."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INFO_KEYS = {"uav_position", "battery", "battery_percent", "sensors_collected", "current_step", "total_reward",
             "total_data_collected", "coverage_percentage", "is_alive", "max_urgency", "avg_urgency",
             "high_urgency_sensors", "capture_effect_triggers", "boundary_hits", "edge_steps",
             "last_step_bytes_collected", "sensor_collection_ratios"}          # uav_env.py:676-700


def _mods():
    import torch
    import uavenv_amd as U
    from oracle import oracle as O
    return torch, U, O


def test_gym_env_api_matches_oracle():
    torch, U, O = _mods()
    kw = dict(grid_size=(120, 120), num_sensors=20, max_steps=60, sensor_duty_cycle=70.0)
    env = U.UAVEnvironment(seed=5, env_index=3, **kw)
    assert env.action_space.n == 5 and env.observation_space.shape == (63,)
    assert env.observation_space.dtype == np.float32
    orc = O.OracleEnv(O.default_config(grid_size=(120, 120), num_sensors=20, max_steps=60, duty_cycle=70.0, seed=5), 3)
    obs, info = env.reset()
    assert obs.dtype == np.float32 and obs.shape == (63,) and set(info) == INFO_KEYS
    assert np.array_equal(obs, orc.reset_keyed())
    assert np.array_equal(env.uav.position, np.zeros(2, np.float32)) and env.uav.battery == 274.0
    rng = np.random.default_rng(0)
    total = 0.0
    for s in range(130):
        a = int(rng.integers(0, 5))
        obs, r, term, trunc, info = env.step(a)
        oo, rr, tr = orc.step_keyed(a)
        assert term is False and isinstance(r, float) and isinstance(trunc, bool)
        assert np.max(np.abs(obs - oo)) <= 1e-6 and abs(r - rr) <= 1e-9 * max(1, abs(rr)) and trunc == tr
        total += r
        st = orc.state()
        assert info["current_step"] == st["step"] and abs(info["battery"] - st["battery"]) < 1e-9
        assert info["sensors_collected"] == int(st["visited"].sum())
        assert abs(info["total_data_collected"] - st["total_collected"]) < 1e-6
        assert len(info["sensor_collection_ratios"]) == 20
        if s % 25 == 0:      # attribute reads the reference's callers perform
            assert env.sensors[3].spreading_factor == st["sf"][3]
            assert abs(env.sensors[3].data_buffer - st["buffer"][3]) < 1e-9
            assert env.sensors_visited == set(np.nonzero(st["visited"])[0].tolist())
            assert env.current_step == st["step"] and abs(env.total_data_collected - st["total_collected"]) < 1e-6
            assert np.array_equal(env.uav.position, np.array([st["uav_x"], st["uav_y"]], np.float32))
            assert env.uav.is_alive() == (st["battery"] > 0.02 * 274.0)
        if trunc:
            obs, info = env.reset()
            assert np.array_equal(obs, orc.reset_keyed())
            assert env.current_step == 0 and env.sensors_visited == set()
    env.close()


def test_gym_env_invalid_action_raises_after_ageing():
    torch, U, O = _mods()
    env = U.UAVEnvironment(grid_size=(50, 50), num_sensors=4)
    env.reset(seed=1)
    g0 = [s.total_data_generated for s in env.sensors]
    with pytest.raises(ValueError):
        env.step(5)
    assert env.current_step == 1 and all(s.total_data_generated > g for s, g in zip(env.sensors, g0))
    obs, r, te, tr, info = env.step(4)          # still usable afterwards
    assert obs.shape == (15,)
    env.close()


def test_explicit_sensor_positions_and_reseed():
    torch, U, O = _mods()
    pos = [(10.0, 10.0), (40.5, 12.25), (25.0, 45.0)]
    env = U.UAVEnvironment(grid_size=(50, 50), sensor_positions=pos, uav_start_position=(25, 25))
    o1, _ = env.reset(seed=9)
    assert env.num_sensors == 3 and env.sensor_positions == pos
    a = [env.step(4)[0] for _ in range(5)]
    o2, _ = env.reset(seed=9)
    b = [env.step(4)[0] for _ in range(5)]
    env2 = U.UAVEnvironment(grid_size=(50, 50), sensor_positions=pos, uav_start_position=(25, 25))
    o3, _ = env2.reset(seed=9)
    c = [env2.step(4)[0] for _ in range(5)]
    assert np.array_equal(o1, o3) and all(np.array_equal(x, y) for x, y in zip(a, c))   # ... same seed+episode => same run
    env.close(); env2.close()


def test_domain_rand_env_matches_oracle_flags():
    torch, U, O = _mods()
    base = dict(max_steps=50, sensor_duty_cycle=60.0)
    env = U.DomainRandEnv(fixed_num_sensors=20, curriculum_stage=2, base_config=base, seed=12, env_index=40)
    assert env.observation_space.shape == (153,)
    ocfg = O.default_config(num_sensors=20, pad_sensors=50, max_steps=50, duty_cycle=60.0, seed=12, flags=1 | 2 | 4 | 8,
                            grid_size=(100, 100), grid_choices=[(100, 100), (200, 200), (300, 300)])
    orc = O.OracleEnv(ocfg, 40)
    obs, info = env.reset()
    assert np.array_equal(obs, orc.reset_keyed())
    assert env.grid_size == (int(orc.e.grid_w), int(orc.e.grid_h)) and env.grid_size in [(100, 100), (200, 200), (300, 300)]
    assert np.all(obs[3 + 3 * 20:] == 0)                    # ghost slots (dqn.py:286-298)
    assert env.last_episode_stats is None
    rng = np.random.default_rng(1)
    for s in range(120):
        a = int(rng.integers(0, 5))
        obs, r, te, tr, info = env.step(a)
        oo, rr, otr = orc.step_keyed(a)
        assert np.max(np.abs(obs - oo)) <= 1e-6 and abs(r - rr) <= 1e-9 * max(1, abs(rr)) and tr == otr
        if tr:
            st = orc.state()
            obs, info = env.reset()
            assert np.array_equal(obs, orc.reset_keyed())
            les = env.last_episode_stats
            assert les is not None and les["num_sensors"] == 20
            assert abs(les["total_collected"] - st["tx"].sum()) < 1e-6
            assert abs(les["ndr"] - st["visited"].sum() / 20 * 100) < 1e-9
    env.set_curriculum_stage(0)
    for _ in range(60):
        _, _, _, tr, _ = env.step(0)
        if tr:
            env.reset()
            assert env.grid_size == (100, 100)
            break
    env.close()


def test_vec_env_sb3_contract():
    torch, U, O = _mods()
    E = 6
    venv = U.UAVVecEnv(E, num_sensors=10, grid_size=(80, 80), max_steps=20, sensor_duty_cycle=100.0, seed=4)
    assert venv.num_envs == E and venv.action_space.n == 5 and venv.observation_space.shape == (33,)
    ocfg = O.default_config(num_sensors=10, grid_size=(80, 80), max_steps=20, duty_cycle=100.0, seed=4)
    envs = [O.OracleEnv(ocfg, k) for k in range(E)]
    obs = venv.reset()
    assert obs.shape == (E, 33) and obs.dtype == np.float32
    for k in range(E):
        assert np.array_equal(obs[k], envs[k].reset_keyed())
    rng = np.random.default_rng(2)
    ep_ret = np.zeros(E)
    saw_done = False
    for s in range(45):
        acts = rng.integers(0, 5, size=E)
        venv.step_async(acts)
        obs, rews, dones, infos = venv.step_wait()
        assert rews.dtype == np.float32 and dones.dtype == bool and len(infos) == E
        for k in range(E):
            oo, rr, tr = envs[k].step_keyed(int(acts[k]))
            ep_ret[k] += rr
            assert dones[k] == tr and abs(rews[k] - rr) <= 1e-5 * max(1, abs(rr))
            if tr:
                saw_done = True
                info = infos[k]
                assert np.max(np.abs(info["terminal_observation"] - oo)) <= 1e-6
                assert info["TimeLimit.truncated"] is True
                assert info["episode"]["l"] == 20
                assert abs(info["episode"]["r"] - ep_ret[k]) <= 1e-9 * max(1, abs(ep_ret[k]))
                assert info["last_episode_stats"]["num_sensors"] == 10
                assert info["last_episode_stats"]["time_to_coverage"] is None          # dqn.py:302 vs :330
                # the 17 keys of the reference's _get_info() (uav_env.py:676-700) for the TERMINAL step, from the oracle's state
                # before its reset: what BestByMetricCallback reads through infos[0] (dqn.py:1150-1155)
                st = envs[k].state()
                want_ratios = [float(t_ / max(g_, 1e-6)) for t_, g_ in zip(st["tx"], st["gen"])]
                assert np.allclose(info["sensor_collection_ratios"], want_ratios, rtol=1e-12, atol=0) and len(want_ratios) == 10
                assert abs(info["total_data_collected"] - float(st["total_collected"])) <= 1e-9
                assert abs(info["battery"] - float(st["battery"])) <= 1e-9
                assert abs(info["battery_percent"] - float(st["battery"]) / 274.0 * 100) <= 1e-9
                assert info["current_step"] == 20 and info["sensors_collected"] == int(st["visited"].sum())
                assert info["coverage_percentage"] == int(st["visited"].sum()) / 10 * 100
                assert abs(info["total_reward"] - float(st["total_reward"])) <= 1e-9 * max(1, abs(float(st["total_reward"])))
                urg = (st["buffer"] / 2.2).astype(np.float32)
                assert info["max_urgency"] == float(np.max(urg)) and info["avg_urgency"] == float(np.mean(urg))
                assert info["high_urgency_sensors"] == int(np.sum(urg > 0.8))
                assert info["capture_effect_triggers"] == int(st["capture_triggers"]) and info["boundary_hits"] == int(st["boundary_hits"])
                assert info["edge_steps"] == int(st["edge_steps"]) and abs(info["last_step_bytes_collected"] - float(st["last_bytes"])) <= 1e-9
                assert np.array_equal(info["uav_position"], np.array([st["uav_x"], st["uav_y"]], np.float32)) and info["is_alive"] is True
                assert len(info) == 17 + 4
                ep_ret[k] = 0.0
                oo = envs[k].reset_keyed()
            else:
                assert infos[k] == {}
            assert np.max(np.abs(obs[k] - oo)) <= 1e-6
    assert saw_done
    assert venv.get_attr("last_episode_stats")[0]["num_sensors"] == 10
    assert venv.env_is_wrapped(object) == [False] * E
    with pytest.raises(ValueError):
        venv.step_async(np.array([0, 1, 2, 3, 4, 5]))
    venv.close()
    dr = U.UAVVecEnv(4, domain_rand=True, num_sensors=20, max_steps=15)
    assert dr.observation_space.shape == (153,)
    dr.reset()
    dr.env_method("set_curriculum_stage", 3)
    for _ in range(20):
        dr.step(np.zeros(4, dtype=np.int64))
    assert all(g in [(100, 100), (200, 200), (300, 300), (400, 400)] for g in dr.get_attr("grid_size"))
    dr.close()


def test_vec_env_per_worker_sensor_counts_and_zero_copy_host_buffers():
    """dqn.py:1065, :1223-1234: every worker of the reference trainer is pinned to its own sensor count
    (WORKER_SENSOR_COUNTS = [10, 20, 30, 40]); observations keep 50 slots (zero padded).  Also the host_copies=False mode:
    views of rotating pinned buffers hold the same numbers as fresh copies."""
    torch, U, O = _mods()
    E, counts = 8, [10, 20, 30, 40]
    kw = dict(domain_rand=True, num_sensors=counts, max_steps=9, seed=12)
    a = U.UAVVecEnv(E, **kw)
    b = U.UAVVecEnv(E, host_copies=False, host_buffers=3, **kw)
    assert a.observation_space.shape == (153,) and a.get_attr("num_sensors") == [counts[i % 4] for i in range(E)]
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa, ob)
    for i in range(E):
        n = counts[i % 4]
        assert np.all(oa[i, 3 + 3 * n:] == 0.0) and np.any(oa[i, 3:3 + 3 * n] != 0.0)     # ghost slots are zero (dqn.py:286-298)
    rng = np.random.default_rng(1)
    kept = []
    for s in range(30):
        acts = rng.integers(0, 5, size=E)
        oa, ra, da, ia = a.step(acts)
        ob, rb, db, ib = b.step(acts)
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db)
        kept.append((s, ob, oa.copy()))
        for back in (1, 2):                              # a view stays valid for host_buffers - 1 = 2 further steps
            if len(kept) > back:
                assert np.array_equal(kept[-1 - back][1], kept[-1 - back][2])
        for i in np.nonzero(da)[0]:
            n = counts[i % 4]
            assert len(ia[i]["sensor_collection_ratios"]) == n == ia[i]["last_episode_stats"]["num_sensors"]
            assert ia[i]["sensor_collection_ratios"] == ib[i]["sensor_collection_ratios"]
            assert ia[i]["episode"]["l"] == 9
    a.close(); b.close()


def _sb3_frame_stack_step(stacked, obs, done, term, D):
    """numpy restatement of SB3 2.x StackedObservations.update for flat Box observations
    (stable_baselines3/common/vec_env/stacked_observations.py; SB3 itself is not installed here, so this
    row's parity is pinned by this restatement only -- "parity unpinned" by the reference's own files)."""
    stacked = np.roll(stacked, -D, axis=-1)
    terminal = {}
    for i in np.nonzero(done)[0]:
        terminal[i] = np.concatenate([stacked[i, :-D], term[i]])
        stacked[i] = 0
    stacked[:, -D:] = obs
    return stacked, terminal


@pytest.mark.parametrize("k,n", [(4, 50), (10, 50), (4, 20)])
def test_frame_stack_matches_sb3_semantics(k, n):
    torch, U, O = _mods()
    E = 37
    env = U.BatchedUAVEnv(E, num_sensors=n, max_steps=9, seed=2)
    D = env.obs_dim
    fs = U.FrameStack(E, D, k, env.device)
    obs = env.reset()
    got = fs.reset(obs).cpu().numpy()
    ref = np.zeros((E, k * D), np.float32); ref[:, -D:] = obs.cpu().numpy()
    assert np.array_equal(got, ref)
    for s in range(40):
        o, r, d = env.step_random()
        got = fs.step(o, d, env.terminal_obs).cpu().numpy()
        ref, terminal = _sb3_frame_stack_step(ref, o.cpu().numpy(), d.cpu().numpy(), env.terminal_obs.cpu().numpy(), D)
        assert np.array_equal(got, ref), s
        ts = fs.terminal_stacked.cpu().numpy()
        for i, row in terminal.items():
            assert np.array_equal(ts[i], row), (s, i)
    env.close()


def test_replay_ring_with_terminal_pool():
    """Kernel-written observations + terminal pool: sampled transitions are (obs_t, a_t, r_t, obs_{t+1}) with
    next_obs = the TERMINAL observation where the episode ended (SB3 replay semantics), checked against the oracle."""
    torch, U, O = _mods()
    E, steps = 48, 60
    kw = dict(num_sensors=20, max_steps=13, duty_cycle=70.0, grid_size=(90, 90), seed=8)
    env = U.BatchedUAVEnv(E, **kw)
    want = O.trace_keyed(O.default_config(**kw), E, steps)
    ring = U.TransitionRing(steps + 2, E, env.obs_dim, env.device)
    ring.attach(env)
    env.reset(); ring.local_obs_slot().copy_(env.obs)
    ring.commit(torch.zeros(E, device=env.device), torch.zeros(E, device=env.device), torch.zeros(E, device=env.device))
    for s in range(steps):
        env.step_random(obs_out=ring.local_obs_slot())
        ring.commit()                                  # the kernel wrote (action, reward, done, terminal ticket) itself
    torch.cuda.synchronize()
    # slot s+1 holds obs after step s together with (action, reward, done) OF step s
    obs = ring.obs[:, 0].cpu().numpy(); aux = ring.aux[:, 0].cpu().numpy()
    all_e = torch.arange(E, device=env.device)
    assert np.array_equal(obs[0], want["reset_obs"])
    n_term = 0
    for s in range(steps):
        assert np.array_equal(obs[s + 1], want["obs"][s])
        assert np.array_equal(aux[s + 1, :, 0].astype(np.int32), want["actions"][s])
        assert np.array_equal(aux[s + 1, :, 2] > 0.5, want["done"][s].astype(bool))
        tick = ring.tickets_at(s + 1, 0, all_e)
        rows = ring.terminal_at(s + 1, 0, tick.clamp(min=0).long()).cpu().numpy()
        tick = tick.cpu().numpy()
        for k in np.nonzero(want["done"][s])[0]:
            assert tick[k] >= 0
            assert np.array_equal(rows[k], want["term_obs"][s, k]); n_term += 1
        assert np.all(tick[~want["done"][s].astype(bool)] == -1)
    assert n_term >= E * 3
    # and through the sampler: every episode end is a valid transition whose next_obs is its terminal observation
    b = ring.sample(4096, generator=torch.Generator(device=env.device).manual_seed(1))
    assert b["valid"].all() and b["done"].any()
    env.close()


@pytest.mark.parametrize("L,period", [(1, 7), (8, 3)])
def test_replay_ring_all_envs_end_on_the_same_step(L, period):
    """Step-limit episodes end for EVERY environment on the same step (a pool sized for the average rate of
    episode ends hands out overwritten rows).  Terminal sections hold E rows per chunk, so no row is lost with one
    burst per chunk -- and when a chunk sees more episode ends than rows, the overwritten ones come back valid=False,
    never as another environment's observation."""
    torch, U, O = _mods()
    E = 1500                                            # more than the old default pool of 1024 rows
    kw = dict(num_sensors=10, max_steps=period, grid_size=(60, 60), seed=4)
    env = U.BatchedUAVEnv(E, **kw)
    D = env.obs_dim
    ring = U.TransitionRing(32, E, D, env.device, chunk_len=L); ring.attach(env)
    obs = env.reset(); ring.local_obs_slot().copy_(obs)
    z = torch.zeros(E, device=env.device); ring.commit(z, z, z)
    per_step = [None]
    for s in range(27):
        o, r, d = env.step_random(obs_out=ring.local_obs_slot())
        slot = ring.commit()
        per_step.append((slot, d.clone()))
    torch.cuda.synchronize()
    assert all(bool(d.all()) == ((s % period) == 0) for s, (slot, d) in enumerate(per_step[1:], 1))   # all end together
    b = ring.sample(20000, generator=torch.Generator(device=env.device).manual_seed(2))
    done = b["done"]
    assert done.any()
    if L == 1:                                          # one burst per chunk: nothing lost
        assert b["valid"].all()
    else:                                               # a chunk of 8 slots sees two or three bursts of E ends: only the last survives
        assert (~b["valid"]).any() and b["valid"][done].any()
    # every valid terminal transition carries ITS OWN environment's terminal observation: header = position / battery of a
    # UAV that has flown `period` steps, never the (0, 0, 1) header of a freshly reset one
    nxt = b["next_obs"][done & b["valid"]]
    assert (nxt[:, 2] < 1.0).all()
    env.close()


def test_stacked_sampling_equals_frame_stack():
    """TransitionRing.sample_stacked gathers the same stacks the device FrameStack produced while acting."""
    torch, U, O = _mods()
    E, steps, k = 24, 50, 4
    env = U.BatchedUAVEnv(E, num_sensors=10, max_steps=11, seed=3)
    D = env.obs_dim
    ring = U.TransitionRing(steps + 2, E, D, env.device); ring.attach(env)
    fs = U.FrameStack(E, D, k, env.device)
    obs = env.reset(); ring.local_obs_slot().copy_(obs)
    z = torch.zeros(E, device=env.device); ring.commit(z, z, z)
    stacks = [fs.reset(obs).clone()]; terms = [None]
    for s in range(steps):
        o, r, d = env.step_random(obs_out=ring.local_obs_slot())
        ring.commit()
        # (with the ring attached the kernel writes terminal rows into the slot's terminal section, not env.terminal_obs)
        slot = ring.head - 1
        tick = ring.tickets_at(slot, 0, torch.arange(E, device=env.device))
        stacks.append(fs.step(o, d, None).clone()); terms.append((d.clone(), ring.terminal_at(slot, 0, tick.clamp(min=0).long()).clone()))
    g = torch.Generator(device=env.device).manual_seed(0)
    b = ring.sample_stacked(4096, k, generator=g)
    g = torch.Generator(device=env.device).manual_seed(0)          # replay the index draws
    n_slots = ring.sampleable() - 1
    j = torch.randint(0, n_slots, (4096,), generator=g, device=env.device)
    _ = torch.randint(0, 1, (4096,), generator=g, device=env.device)
    e = torch.randint(0, E, (4096,), generator=g, device=env.device)
    checked_term = 0
    for i in range(0, 4096, 7):
        s, ei = int(j[i]), int(e[i])
        assert torch.equal(b["obs"][i], stacks[s][ei]), (i, s, ei)
        d, tix = terms[s + 1]
        if d[ei]:      # SB3's stacked terminal_observation: [old frames shifted | terminal obs]
            want = torch.cat([stacks[s][ei][D:], tix[ei]])
            assert torch.equal(b["next_obs"][i], want); checked_term += 1
        else:
            assert torch.equal(b["next_obs"][i], stacks[s + 1][ei])
    assert checked_term > 5 and b["valid"].all()
    env.close()


def test_vec_env_with_builtin_frame_stack():
    """UAVVecEnv(n_stack=k) == VecFrameStack(k) over UAVVecEnv (SB3 semantics restated in numpy above)."""
    torch, U, O = _mods()
    E, k = 5, 4
    kw = dict(num_sensors=10, grid_size=(80, 80), max_steps=12, seed=6)
    a = U.UAVVecEnv(E, n_stack=k, **kw)
    b = U.UAVVecEnv(E, **kw)
    D = b.observation_space.shape[0]
    assert a.observation_space.shape == (k * D,)
    oa, ob = a.reset(), b.reset()
    ref = np.zeros((E, k * D), np.float32); ref[:, -D:] = ob
    assert np.array_equal(oa, ref)
    rng = np.random.default_rng(0)
    for s in range(40):
        acts = rng.integers(0, 5, size=E)
        oa, ra, da, ia = a.step(acts)
        ob, rb, db, ib = b.step(acts)
        term = np.zeros((E, D), np.float32)
        for i in np.nonzero(db)[0]:
            term[i] = ib[i]["terminal_observation"]
        ref, terminal = _sb3_frame_stack_step(ref, ob, db, term, D)
        assert np.array_equal(oa, ref) and np.array_equal(ra, rb) and np.array_equal(da, db)
        for i, row in terminal.items():
            assert np.array_equal(ia[i]["terminal_observation"], row)
    a.close(); b.close()


@pytest.mark.parametrize("k,sensors,positions", [(5, 10, False), (1, 10, False), (4, 10, False), (12, 10, False), (16, 10, False),
                                                 (4, 50, True), (3, 64, True)])
def test_hip_ring_gather_equals_the_pytorch_statement(k, sensors, positions):
    """uavenv_ring_gather_stacked (one launch per sampled batch) against TransitionRing.stacked_batch_at_torch (the tensor
    expressions it replaces), on a chunked ring that has wrapped around, with a terminal section too small for the episode-end
    bursts (overwritten rows must come back as valid = False) and draws that reach before the ring start.  Stack depths on both
    sides of the kernel's compile-time bounds (4, 10, 16); rows of 33, 253 and 323 floats (the last one takes two passes)."""
    torch, U, O = _mods()
    E = 40
    env = U.BatchedUAVEnv(E, num_sensors=sensors, max_steps=7, seed=11, include_sensor_positions=positions)   # an episode ends every 7 steps
    D = env.obs_dim
    assert D == 3 + sensors * (5 if positions else 3)
    ring = U.TransitionRing(24, E, D, env.device, chunk_len=6, terminal_rows=16); ring.attach(env)
    obs = env.reset(); ring.local_obs_slot().copy_(obs)
    z = torch.zeros(E, device=env.device); ring.commit(z, z, z)
    for s in range(61):                                                    # 2.5 revolutions
        env.step_random(obs_out=ring.local_obs_slot()); ring.commit()
    n, oldest = ring.window_state()
    g = torch.Generator(device=env.device).manual_seed(4)
    B = 3000
    j = torch.randint(0, n - 1, (B,), generator=g, device=env.device)
    j[:40] = torch.arange(40, device=env.device) % max(1, min(4, k - 1))   # stacks that reach before the ring start
    slot = (oldest + j) % ring.capacity
    r = torch.zeros(B, dtype=torch.int64, device=env.device)
    e = torch.randint(0, E, (B,), generator=g, device=env.device)
    got = ring.stacked_batch_at(j, slot, r, e, k)
    want = ring.stacked_batch_at_torch(j, slot, r, e, k)
    for key in ("obs", "next_obs", "action", "reward", "done", "valid"):
        assert got[key].dtype == want[key].dtype and got[key].shape == want[key].shape, key
        assert torch.equal(got[key], want[key]), key
    assert bool(got["done"].any()) and not bool(got["valid"].all()) and bool(got["valid"].any())
    if k > 4:
        assert bool((got["obs"][:40, :D] == 0).all())                      # the frames from before the ring start are zero
    env.close()


def test_keyed_in_kernel_draw_is_uniform_and_equals_the_gather_at_its_indices():
    """uavenv_ring_sample_stacked: the draw made inside the gather kernel.  Same counter -> same batch; another counter -> another
    batch; the batch equals stacked_batch_at() at the indices it reports; ages / ranks / environments cover their ranges evenly."""
    torch, U, O = _mods()
    E, k = 48, 3
    env = U.BatchedUAVEnv(E, num_sensors=10, grid_size=(50, 50), max_steps=11, seed=8)
    ring = U.TransitionRing(40, E, env.obs_dim, env.device, chunk_len=8)
    ring.attach(env)
    ring.local_obs_slot().copy_(env.reset())
    z = torch.zeros(E, device=env.device)
    ring.commit(z, z, z)
    for _ in range(60):
        env.step_random(obs_out=ring.local_obs_slot())
        ring.commit()
    n, oldest = ring.window_state()
    window = torch.tensor([n, oldest], dtype=torch.int64, device=env.device)
    counter = torch.zeros(1, device=env.device)
    a = ring.sample_stacked_keyed(4096, k, window, counter, seed=123)
    a = {key: (v.clone() if torch.is_tensor(v) else tuple(t.clone() for t in v)) for key, v in a.items()}
    b = ring.sample_stacked_keyed(4096, k, window, counter, seed=123)
    assert all(torch.equal(a[key], b[key]) for key in ("obs", "next_obs", "action", "reward", "done", "valid", "index_block"))
    counter.fill_(1.0)
    c = ring.sample_stacked_keyed(4096, k, window, counter, seed=123)
    assert not torch.equal(a["index_block"], c["index_block"])
    j, slot, r, e = a["index"]
    assert int(j.min()) == 0 and int(j.max()) == n - 2 and torch.equal(slot, (oldest + j) % ring.capacity) and int(r.max()) == 0
    assert int(e.min()) == 0 and int(e.max()) == E - 1
    cnt_e = torch.bincount(e, minlength=E).float()
    cnt_j = torch.bincount(j, minlength=n - 1).float()
    assert float(cnt_e.min()) > 0.5 * 4096 / E and float(cnt_j.min()) > 0.4 * 4096 / (n - 1)
    want = ring.stacked_batch_at(j, slot, r, e, k)
    for key in ("obs", "next_obs", "action", "reward", "done", "valid"):
        assert torch.equal(a[key], want[key]), key
    assert bool(a["done"].any())
    env.close()
