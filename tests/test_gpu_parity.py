"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(libuavenv_hip.so via ctypes), against

  1. the golden fixtures captured from the REAL reference (tests/golden/*.npz), under the injected tape;
  2. the CPU oracle in keyed (Philox) mode on seeded random-policy rollouts, for every lane-group width;
  3. size-independent invariants at BASELINE.json's full size (4096 envs x 50 sensors).

Tolerances (north_star: "within 1e-5 fp32"):
  observations  <= 1e-6 absolute (float32 rows; in practice bit-identical),
  rewards       <= 1e-9 relative (float64; the wave butterfly sums differ from the oracle's sequential
                   sums in the last bits), i.e. four orders inside the 1e-5 contract,
  truncation flags, spreading factors, visited sets, counters, UAV positions: identical.
"""
import numpy as np
import pytest

import golden_util as G
import tape as T

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


def _tape_tensors(torch, seed, envs, step, n, stride, device):
    st = np.zeros((len(envs), T.NUM_STEP_SLOTS, stride), np.float32)
    for k, e in enumerate(envs):
        st[k, :, :n] = T.step_tape(seed, e, step, n)
    return torch.from_numpy(st).to(device)


def _reset_tape_tensor(torch, seed, envs, episodes, n, stride, device):
    rt = np.zeros((len(envs), T.NUM_RESET_SLOTS, stride), np.float32)
    for k, (e, ep) in enumerate(zip(envs, episodes)):
        rt[k, :, :n] = T.reset_tape(seed, e, ep, n)
    return torch.from_numpy(rt).to(device)


@pytest.mark.parametrize("auto_reset", [False, True])
@pytest.mark.parametrize("name", G.fixture_names())
def test_hip_replays_reference_fixture(name, auto_reset):
    """The HIP kernel reproduces the real reference's trajectory (obs, reward, truncation, SF, final
    state) step for step.  Run as a batch of 3 identical environments to cover the batch indexing."""
    torch, U, O = _mods()
    fx = G.load(name)
    meta = fx["meta"]
    n, seed = meta["n"], meta["tape_seed"]
    E = 3
    px, py = T.positions(seed, 0, n, meta["grid"][0], meta["grid"][1])
    env = U.BatchedUAVEnv(E, auto_reset=auto_reset, sensor_positions=np.stack([px, py], -1), **G.config_overrides(meta))
    S = env.lane_stride
    dev = env.device
    episode = 0
    env.set_noise_tape(None, _reset_tape_tensor(torch, seed, [0] * E, [0] * E, n, S, dev))
    obs = env.reset().cpu().numpy()
    for k in range(E):
        assert np.array_equal(obs[k], fx["reset_obs"][0])
    for s, a in enumerate(fx["actions"]):
        st = _tape_tensors(torch, seed, [0] * E, s, n, S, dev)
        rt = _reset_tape_tensor(torch, seed, [0] * E, [episode + 1] * E, n, S, dev)
        env.set_noise_tape(st, rt)
        actions = torch.full((E,), int(a), dtype=torch.int32, device=dev)
        o, r, d = env.step(actions)
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        tr = bool(fx["truncated"][s])
        step_obs = env.terminal_obs.cpu().numpy() if (tr and auto_reset) else o
        for k in range(E):
            assert np.max(np.abs(step_obs[k] - fx["obs"][s])) <= OBS_ATOL, (name, s, k)
            assert _rel(r[k], fx["reward"][s]) <= REW_RTOL, (name, s, r[k], fx["reward"][s])
            assert bool(d[k]) == tr, (name, s)
        if not (tr and auto_reset):
            assert np.array_equal(env.sensor_state(0)["sf"], fx["sf"][s]), (name, s)
        if tr:
            episode += 1
            if auto_reset:
                for k in range(E):
                    assert np.array_equal(o[k], fx["reset_obs"][episode]), (name, s)
            else:
                env.set_noise_tape(None, rt)
                o = env.reset().cpu().numpy()
                for k in range(E):
                    assert np.array_equal(o[k], fx["reset_obs"][episode]), (name, s)
    ss = env.sensor_state(1)
    rec = env.records()[1]
    for k in ("sf", "visited", "data_collected"):
        assert np.array_equal(ss[k], fx["final_" + k]), k
    for k in ("buffer", "gen", "tx", "lost"):
        assert np.allclose(ss[k], fx["final_" + k], rtol=1e-12, atol=1e-9), k
    assert np.allclose(ss["avg_rssi"], fx["final_avg_rssi"], rtol=0, atol=4e-5, equal_nan=True)
    assert rec["uav_x"] == fx["final_uav_x"] and rec["uav_y"] == fx["final_uav_y"]
    assert rec["current_step"] == fx["final_step"]
    assert abs(rec["battery"] - fx["final_battery"]) < 1e-9
    assert _rel(rec["total_reward"], float(fx["final_total_reward"])) < REW_RTOL
    assert abs(rec["total_data_collected"] - fx["final_total_collected"]) < 1e-6
    for k in ("capture_triggers", "boundary_hits", "edge_steps"):
        assert rec[k] == fx["final_" + k], k
    env.close()


def test_noise_matches_oracle_bit_for_bit():
    """Philox words and the transcendental-free Box-Muller are integer / IEEE-only: device == oracle."""
    torch, U, O = _mods()
    for n in (10, 20, 50):
        env = U.BatchedUAVEnv(7, num_sensors=n, seed=0x1234ABCD9876, env_index_base=1000)
        env.reset()
        for _ in range(3):
            env.step_random()
        st, rt = env.dump_noise()
        st, rt = st.cpu().numpy(), rt.cpu().numpy()
        rec = env.records()
        L = O.lib()
        for k in range(7):
            want = np.zeros((7, n), np.float32)
            L.orc_noise_step_tape(0x1234ABCD9876, 1000 + k, int(rec["episode"][k]), int(rec["current_step"][k]) + 1, n,
                                  O._fp(want))
            assert np.array_equal(st[k, :, :n], want), (n, k)
            wr = np.zeros((4, n), np.float32)
            L.orc_noise_reset_tape(0x1234ABCD9876, 1000 + k, int(rec["episode"][k]) + 1, n, O._fp(wr))
            assert np.array_equal(rt[k, :, :n], wr), (n, k)
        env.close()


KEYED_CASES = [
    # name, E, oracle/product config overrides, steps, flags
    ("n50_default", 96, dict(num_sensors=50), 260, 0),
    ("n50_trunc", 64, dict(num_sensors=50, max_steps=37, duty_cycle=60.0, grid_size=(120, 120)), 200, 0),
    ("n20_g32", 100, dict(num_sensors=20, max_battery=9.0, duty_cycle=80.0, grid_size=(150, 150)), 250, 0),
    ("n10_g16", 130, dict(num_sensors=10, max_steps=50, duty_cycle=100.0, grid_size=(90, 90)), 250, 0),
    ("n33_pad50_pos", 40, dict(num_sensors=33, pad_sensors=50, include_sensor_positions=1, duty_cycle=70.0,
                               grid_size=(100, 100), max_steps=80), 200, 0),
    ("n64_full_wave", 20, dict(num_sensors=64, duty_cycle=100.0, grid_size=(60, 60), max_steps=90), 200, 0),
    ("n1", 9, dict(num_sensors=1, duty_cycle=100.0, grid_size=(20, 20), start_x=5.0, start_y=5.0, max_steps=70), 200, 0),
    ("domain_rand", 72, dict(num_sensors=20, pad_sensors=50, duty_cycle=50.0, max_steps=60,
                             grid_choices=[(100, 100), (200, 200), (300, 300)]), 250, 1 | 2 | 4 | 8),
    ("shaping_only", 40, dict(num_sensors=10, duty_cycle=100.0, grid_size=(100, 100), max_steps=55), 150, 4 | 8),
    # batches big enough for the 16-wave workgroups with SIMD load balancing, for every lane-group width
    ("n50_big_workgroups", 4096 + 21, dict(num_sensors=50, max_steps=9, duty_cycle=60.0, grid_size=(150, 150)), 14, 0),
    ("n20_big_workgroups", 8192 + 5, dict(num_sensors=20, max_steps=7, duty_cycle=80.0, grid_size=(120, 120)), 10, 0),
    ("n10_big_workgroups", 16384 + 3, dict(num_sensors=10, max_steps=6, duty_cycle=100.0, grid_size=(90, 90)), 8, 0),
    # the DEFAULT constants (only the step limit and the sizes differ: the literal-constant kernels run), small and chip-filling
    # batches (write-through stores), every lane-group width
    ("n50_default_consts", 200, dict(num_sensors=50, max_steps=40), 90, 0),
    ("n20_default_consts", 300, dict(num_sensors=20, max_steps=40), 90, 0),
    ("n10_default_consts", 500, dict(num_sensors=10, max_steps=40), 90, 0),
    ("n50_default_consts_big", 4096 + 21, dict(num_sensors=50, max_steps=9), 14, 0),
    ("n20_default_consts_big", 8192 + 5, dict(num_sensors=20, max_steps=7), 10, 0),
    ("n10_default_consts_big", 16384 + 3, dict(num_sensors=10, max_steps=6), 8, 0),
]


@pytest.mark.parametrize("case", KEYED_CASES, ids=[c[0] for c in KEYED_CASES])
def test_keyed_rollout_matches_oracle(case):
    """Random-policy rollouts with in-kernel Philox noise and auto-reset, compared env by env and step
    by step with the keyed oracle (same seed, same GLOBAL env indices)."""
    torch, U, O = _mods()
    name, E, over, steps, flags = case
    seed, base = 20260704, 5000
    ocfg = O.default_config(seed=seed, flags=flags, **over)
    want = O.trace_keyed(ocfg, E, steps, base=base, auto_reset=True)
    env = U.BatchedUAVEnv(E, env_index_base=base, auto_reset=True, flags=flags, seed=seed, **over)
    assert env.obs_dim == want["obs"].shape[-1]
    o = env.reset().cpu().numpy()
    assert np.array_equal(o, want["reset_obs"]), name
    n_done = 0
    for s in range(steps):
        o, r, d = env.step_random()
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        a = env.actions_taken.cpu().numpy()
        assert np.array_equal(a, want["actions"][s]), (name, s)
        assert np.array_equal(d, want["done"][s]), (name, s)
        assert np.max(np.abs(o - want["obs"][s])) <= OBS_ATOL, (name, s, np.max(np.abs(o - want["obs"][s])))
        assert np.max(_rel(r, want["reward"][s])) <= REW_RTOL, (name, s)
        if d.any():
            t = env.terminal_obs.cpu().numpy()
            m = d.astype(bool)
            assert np.max(np.abs(t[m] - want["term_obs"][s][m])) <= OBS_ATOL, (name, s)
            n_done += int(m.sum())
    assert n_done > 0 or "default" in name, "case should exercise truncation + auto-reset"
    # final state
    ss = env.sensor_state()
    rec = env.records()
    for k in range(E):
        f = want["final"][k]
        for key in ("sf", "visited", "data_collected", "pos_x", "pos_y"):
            assert np.array_equal(ss[key][k], f[key]), (name, k, key)
        for key in ("buffer", "gen", "tx", "lost"):
            assert np.allclose(ss[key][k], f[key], rtol=1e-12, atol=1e-9), (name, k, key)
        assert np.allclose(ss["avg_rssi"][k], f["avg_rssi"], rtol=0, atol=1e-9, equal_nan=True), (name, k)
        assert rec["uav_x"][k] == f["uav_x"] and rec["uav_y"][k] == f["uav_y"]
        assert rec["current_step"][k] == f["step"] and rec["episode"][k] == f["episode"]
        assert rec["grid_w"][k] == f["grid_w"] and rec["grid_h"][k] == f["grid_h"]
        assert abs(rec["battery"][k] - f["battery"]) < 1e-9
        for key in ("capture_triggers", "boundary_hits", "edge_steps", "collisions_total"):
            assert rec[key][k] == f[key], (name, k, key)
    env.close()


@pytest.mark.parametrize("width", ["32", "64"])
def test_wider_lane_group_than_needed_matches_oracle(monkeypatch, width):
    """UAVENV_LANE_GROUP (read by uavenv_create): 10 sensors in 32- and 64-lane groups, 20 sensors in 64 lanes -- the same
    environments, stepped by the kernels of the wider group (for 64: one environment per wavefront, scalar record path)."""
    torch, U, O = _mods()
    monkeypatch.setenv("UAVENV_LANE_GROUP", width)
    env = U.BatchedUAVEnv(8, num_sensors=10, seed=1)
    assert env.lane_stride == int(width)
    env.close()
    test_keyed_rollout_matches_oracle(("n10_wide", 130, dict(num_sensors=10, max_steps=50, duty_cycle=100.0, grid_size=(90, 90)), 200, 0))
    if width == "64":
        test_keyed_rollout_matches_oracle(("n20_wide", 100, dict(num_sensors=20, max_battery=9.0, duty_cycle=80.0, grid_size=(150, 150)), 200, 0))
        test_keyed_rollout_matches_oracle(("n20_default_consts_wide", 300, dict(num_sensors=20, max_steps=40), 90, 0))


def test_given_actions_and_manual_reset_match_oracle():
    """auto_reset=False (single gymnasium.Env semantics): explicit actions, masked reset on truncation."""
    torch, U, O = _mods()
    E, steps, seed = 33, 180, 99
    over = dict(num_sensors=20, max_steps=45, duty_cycle=70.0, grid_size=(110, 110))
    rng = np.random.default_rng(5)
    actions = rng.integers(0, 5, size=(steps, E)).astype(np.int32)
    ocfg = O.default_config(seed=seed, **over)
    envs = [O.OracleEnv(ocfg, k) for k in range(E)]
    env = U.BatchedUAVEnv(E, auto_reset=False, seed=seed, **over)
    o = env.reset().cpu().numpy()
    for k in range(E):
        assert np.array_equal(o[k], envs[k].reset_keyed())
    for s in range(steps):
        o, r, d = env.step(torch.from_numpy(actions[s]).to(env.device))
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy().astype(bool)
        for k in range(E):
            oo, rr, tr = envs[k].step_keyed(int(actions[s, k]))
            assert np.max(np.abs(o[k] - oo)) <= OBS_ATOL and _rel(r[k], rr) <= REW_RTOL and d[k] == tr, (s, k)
        if d.any():
            before = o.copy()
            o2 = env.reset(torch.from_numpy(d.astype(np.uint8))).cpu().numpy()
            for k in range(E):
                if d[k]:
                    assert np.array_equal(o2[k], envs[k].reset_keyed()), (s, k)
                else:
                    assert np.array_equal(o2[k], before[k])      # unmasked rows untouched
    env.close()


def test_sharding_is_invisible():
    """Results depend on the GLOBAL env index only: one 96-env instance == two 48-env shards."""
    torch, U, O = _mods()
    kw = dict(num_sensors=50, max_steps=40, duty_cycle=50.0, seed=7)
    whole = U.BatchedUAVEnv(96, env_index_base=0, **kw)
    lo = U.BatchedUAVEnv(48, env_index_base=0, **kw)
    hi = U.BatchedUAVEnv(48, env_index_base=48, **kw)
    a, b, c = whole.reset().clone(), lo.reset().clone(), hi.reset().clone()
    assert torch.equal(a, torch.cat([b, c]))
    for _ in range(100):
        ow, rw, dw = whole.step_random()
        ol, rl, dl = lo.step_random()
        oh, rh, dh = hi.step_random()
        assert torch.equal(ow, torch.cat([ol, oh])) and torch.equal(rw, torch.cat([rl, rh]))
        assert torch.equal(dw, torch.cat([dl, dh]))
    for e in (whole, lo, hi):
        e.close()


def test_full_size_invariants():
    """BASELINE config (4096 envs x 50 sensors, 500x500): size-independent properties of the domain."""
    torch, U, O = _mods()
    E, n = 4096, 50
    env = U.BatchedUAVEnv(E, num_sensors=n, seed=1)
    obs0 = env.reset().clone()
    assert obs0.shape == (E, 153) and torch.isfinite(obs0).all()
    fill = obs0[:, 3::3]
    assert (fill >= 0.2 - 1e-6).all() and (fill <= 0.6 + 1e-6).all()          # uav_env.py:410 U(0.2, 0.6)
    steps = 300
    total_r = torch.zeros(E, dtype=torch.float64, device=env.device)
    for _ in range(steps):
        o, r, d = env.step_random()
        total_r += r
    assert torch.isfinite(o).all() and (o >= -1 - 1e-6).all() and (o <= 1 + 1e-6).all()   # observation_space Box(-1, 1)
    g = lambda f: env.get_state(f)[:, :n]
    from uavenv_amd import _native as N
    b, gen, tx, lost = g(N.F_BUFFER), g(N.F_GEN), g(N.F_TX), g(N.F_LOST)
    # conservation: everything generated is buffered, transmitted or lost
    assert torch.allclose(b + tx + lost, gen, rtol=1e-12, atol=1e-8)
    assert (b >= 0).all() and (b <= 1000.0).all() and (tx >= 0).all() and (lost >= 0).all()
    rec = env.records()
    assert (rec["current_step"] == steps).all() and (rec["episode"] == 0).all()      # no truncation yet (battery lasts >= 1381 steps)
    # generated = prefill + rate * elapsed time (dt == 1 for every action at collection_duration 1.0)
    gen0 = (obs0[:, 3::3].double() * 1000.0).cpu()
    assert torch.allclose(gen.cpu(), gen0 + 2.2 * steps, rtol=1e-6, atol=1e-3)
    # battery bookkeeping: only three possible per-step drains (uav.py:176,180,204)
    used = 274.0 - rec["battery"]
    assert (used >= steps * 250.0 / 3600 - 1e-9).all() and (used <= steps * 700.0 / 3600 + 1e-9).all()
    assert np.allclose(rec["total_reward"], total_r.cpu().numpy(), rtol=1e-12)
    flags = g(N.F_FLAGS)
    sf = flags & 15
    assert bool(((sf == 7) | (sf == 9) | (sf == 11) | (sf == 12)).all())              # ADR-reachable SFs only
    # a spot check of 8 global env indices against the oracle, 300 steps deep
    ocfg = O.default_config(num_sensors=n, seed=1)
    idx = [0, 1, 63, 64, 1000, 2047, 4000, 4095]
    for k in idx:
        w = O.trace_keyed(ocfg, 1, steps, base=k)
        assert np.max(np.abs(o[k].cpu().numpy() - w["obs"][-1, 0])) <= OBS_ATOL, k
        f = w["final"][0]
        assert np.allclose(b[k].cpu().numpy(), f["buffer"], rtol=1e-12, atol=1e-9), k
        assert np.array_equal(sf[k].cpu().numpy(), f["sf"]), k
    env.close()


def test_invalid_action_is_reported_not_fatal():
    torch, U, O = _mods()
    env = U.BatchedUAVEnv(4, num_sensors=5, auto_reset=False)
    env.reset()
    gen0 = env.sensor_state()["gen"].copy()
    env.step(torch.tensor([0, 9, 4, -1], dtype=torch.int32, device=env.device))
    with pytest.raises(ValueError):
        env.check_actions()
    st = env.sensor_state()
    assert (env.records()["current_step"] == 1).all() and (st["gen"] > gen0).all()    # aged like uav_env.py:439-468
    env.check_actions()      # flag cleared
    env.close()


def test_state_roundtrip_and_host_api():
    import ctypes as C
    torch, U, O = _mods()
    from uavenv_amd import _native as N
    kw = dict(num_sensors=20, seed=3, max_steps=30)
    a = U.BatchedUAVEnv(10, **kw)
    a.reset()
    for _ in range(17):
        a.step_random()
    sd = a.state_dict()
    b = U.BatchedUAVEnv(10, **kw)
    b.load_state_dict(sd)
    for _ in range(40):
        oa, ra, da = a.step_random()
        ob, rb, db = b.step_random()
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    # synchronous host-buffer entry points of the C ABI
    L = N.lib()
    E, D = 10, a.obs_dim
    obs = np.zeros((E, D), np.float32); rew = np.zeros(E, np.float64); done = np.zeros(E, np.uint8)
    acts = np.full(E, 4, np.int32)
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    torch.cuda.synchronize()
    N.check(L.uavenv_step_host(a._h, vp(acts), vp(obs), vp(rew), vp(done), None), a._h)
    ob, rb, db = b.step(torch.from_numpy(acts).to(b.device))
    assert np.array_equal(obs, ob.cpu().numpy()) and np.array_equal(rew, rb.cpu().numpy())
    acts[3] = 11
    rc = L.uavenv_step_host(a._h, vp(acts), vp(obs), vp(rew), vp(done), None)
    assert rc == N.E_ACTION
    a.close(); b.close()


@pytest.mark.parametrize("n,flags", [(50, 0), (20, 0), (10, 1 | 2 | 4 | 8)])
def test_fused_rollout_is_bit_identical_to_single_steps(n, flags):
    """uavenv_rollout (K steps per launch, state in registers / LDS) == K uavenv_step_random launches,
    including auto-resets inside the fused launch, for the random policy and for given actions."""
    torch, U, O = _mods()
    kw = dict(num_sensors=n, max_steps=23, duty_cycle=70.0, grid_size=(130, 130), seed=21, flags=flags)
    if flags:
        kw.update(pad_sensors=50, grid_choices=[(100, 100), (200, 200)])
    E, K = 70, 37
    a = U.BatchedUAVEnv(E, **kw)
    b = U.BatchedUAVEnv(E, **kw)
    assert torch.equal(a.reset(), b.reset())
    for rep in range(2):
        ro = a.rollout(K, with_terminal=True)
        for k in range(K):
            o, r, d = b.step_random()
            assert torch.equal(ro["obs"][k], o) and torch.equal(ro["reward"][k], r) and torch.equal(ro["done"][k], d), (rep, k)
            assert torch.equal(ro["actions"][k], b.actions_taken)
            m = d.bool()
            if m.any():
                assert torch.equal(ro["terminal_obs"][k][m], b.terminal_obs[m])
        assert ro["done"].any()
    acts = torch.randint(0, 5, (K, E), dtype=torch.int32, device=a.device)
    ro = a.rollout(K, actions=acts)
    for k in range(K):
        o, r, d = b.step(acts[k])
        assert torch.equal(ro["obs"][k], o) and torch.equal(ro["reward"][k], r) and torch.equal(ro["done"][k], d), k
    sa, sb = a.state_dict(), b.state_dict()
    for key in sa:
        assert torch.equal(sa[key], sb[key]), key
    a.close(); b.close()


@pytest.mark.parametrize("name", G.policy_fixture_names())
def test_hip_policies_replay_reference_agents(name):
    """On-device NearestSensorGreedy / MaxThroughputGreedyV2 (uavenv_step_policy) choose the action the REAL
    reference agent chose at every step, and the resulting trajectory matches (tape incl. slot zP)."""
    torch, U, O = _mods()
    fx = G.load(name)
    meta = fx["meta"]
    n, seed, pid = meta["n"], meta["tape_seed"], meta["policy_id"]
    E = 2
    px, py = T.positions(seed, 0, n, meta["grid"][0], meta["grid"][1])
    env = U.BatchedUAVEnv(E, auto_reset=True, sensor_positions=np.stack([px, py], -1), **G.config_overrides(meta))
    S, dev = env.lane_stride, env.device
    env.set_noise_tape(None, _reset_tape_tensor(torch, seed, [0] * E, [0] * E, n, S, dev))
    assert np.array_equal(env.reset().cpu().numpy()[1], fx["reset_obs"][0])
    episode = 0
    for s in range(meta["steps"]):
        env.set_noise_tape(_tape_tensors(torch, seed, [0] * E, s, n, S, dev),
                           _reset_tape_tensor(torch, seed, [0] * E, [episode + 1] * E, n, S, dev))
        o, r, d = env.step_policy(pid)
        a = env.actions_taken.cpu().numpy()
        assert a[0] == a[1] == int(fx["actions"][s]), (name, s, a, int(fx["actions"][s]))
        tr = bool(fx["truncated"][s])
        step_obs = (env.terminal_obs if tr else o).cpu().numpy()
        assert np.max(np.abs(step_obs[1] - fx["obs"][s])) <= OBS_ATOL, (name, s)
        assert _rel(r.cpu().numpy()[1], fx["reward"][s]) <= REW_RTOL and bool(d[1]) == tr, (name, s)
        if tr:
            episode += 1
            assert np.array_equal(o.cpu().numpy()[1], fx["reset_obs"][episode])
    env.close()


@pytest.mark.parametrize("policy,n", [(2, 50), (3, 50), (3, 20), (2, 10)])
def test_policy_rollouts_match_oracle_and_fuse(policy, n):
    """Keyed (Philox) policy rollouts: per-step kernel == oracle, and the fused K-step kernel == per-step."""
    torch, U, O = _mods()
    E, steps, seed = 40, 120, 77
    over = dict(num_sensors=n, grid_size=(1500, 1500), max_steps=70, duty_cycle=40.0)      # large grid: the policies must navigate
    want = O.trace_keyed(O.default_config(seed=seed, **over), E, steps, policy=policy)
    a = U.BatchedUAVEnv(E, seed=seed, **over)
    b = U.BatchedUAVEnv(E, seed=seed, **over)
    assert np.array_equal(a.reset().cpu().numpy(), want["reset_obs"]); b.reset()
    ro = b.rollout(steps, policy=policy)
    moves = 0
    for s in range(steps):
        o, r, d = a.step_policy(policy)
        act = a.actions_taken.cpu().numpy()
        assert np.array_equal(act, want["actions"][s]), s
        assert np.array_equal(d.cpu().numpy(), want["done"][s])
        assert np.max(np.abs(o.cpu().numpy() - want["obs"][s])) <= OBS_ATOL
        assert np.max(_rel(r.cpu().numpy(), want["reward"][s])) <= REW_RTOL
        assert torch.equal(ro["obs"][s], o) and torch.equal(ro["reward"][s], r) and torch.equal(ro["actions"][s], a.actions_taken)
        moves += int((act != 4).sum())
    assert moves > 0 and want["done"].any()
    a.close(); b.close()


def test_full_episode_parity_including_natural_truncation():
    """A whole reference-length episode and beyond (2300 steps > max_steps 2100; battery death at 1381-1933 steps)
    for 64 envs x 50 sensors under the random policy: the float64 state must not drift (observations stay within
    1e-6, rewards within 1e-9 relative, every truncation and auto-reset on the same step as the oracle)."""
    torch, U, O = _mods()
    E, steps, seed = 64, 2300, 4242
    want = O.trace_keyed(O.default_config(num_sensors=50, seed=seed), E, steps, base=123)
    env = U.BatchedUAVEnv(E, num_sensors=50, seed=seed, env_index_base=123)
    assert np.array_equal(env.reset().cpu().numpy(), want["reset_obs"])
    K = 100
    for s0 in range(0, steps, K):
        ro = env.rollout(K, with_terminal=True)
        o, r, d = ro["obs"].cpu().numpy(), ro["reward"].cpu().numpy(), ro["done"].cpu().numpy()
        sl = slice(s0, s0 + K)
        assert np.array_equal(d, want["done"][sl]), s0
        assert np.array_equal(ro["actions"].cpu().numpy(), want["actions"][sl]), s0
        assert np.max(np.abs(o - want["obs"][sl])) <= OBS_ATOL, (s0, np.max(np.abs(o - want["obs"][sl])))
        assert np.max(_rel(r, want["reward"][sl])) <= REW_RTOL, s0
        m = d.astype(bool)
        if m.any():
            assert np.max(np.abs(ro["terminal_obs"].cpu().numpy()[m] - want["term_obs"][sl][m])) <= OBS_ATOL
    assert want["done"].sum() >= E                      # every environment finished at least one episode
    lens = env.episode_stats()["length"]
    assert lens.min() >= 1381 and lens.max() <= 2100    # all-hover 1381 ... max_steps 2100
    env.close()


def test_mixed_grid_and_sensor_count_in_one_handle():
    """BASELINE config 5 interleaves (grid, N) combinations: per-environment grid and sensor count inside ONE
    handle (uavenv_set_env_params), every environment checked against its own oracle instance."""
    torch, U, O = _mods()
    E, steps, seed = 48, 150, 31
    ns = np.array([10, 20, 33, 50])[np.arange(E) % 4].astype(np.int32)
    grids = np.array([250, 500, 1000])[np.arange(E) % 3].astype(np.int32)
    rng = np.random.default_rng(0)
    pos = np.zeros((E, 50, 2), np.float32)
    for k in range(E):
        pos[k, :ns[k]] = (rng.random((ns[k], 2)) * grids[k]).astype(np.float32)
    env = U.BatchedUAVEnv(E, num_sensors=50, seed=seed, max_steps=60, duty_cycle=50.0, sensor_positions=pos)
    env.set_env_params(grid_w=grids, grid_h=grids, num_sensors=ns)
    orcs = []
    for k in range(E):
        cfg = O.default_config(num_sensors=int(ns[k]), pad_sensors=50, grid_size=(int(grids[k]), int(grids[k])), seed=seed,
                               max_steps=60, duty_cycle=50.0)
        orcs.append(O.OracleEnv(cfg, k, pos[k, :ns[k], 0], pos[k, :ns[k], 1]))
    o = env.reset().cpu().numpy()
    for k in range(E):
        assert np.array_equal(o[k], orcs[k].reset_keyed()), k
    for s in range(steps):
        o, r, d = env.step_random()
        o, r, d, a = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy(), env.actions_taken.cpu().numpy()
        for k in range(E):
            assert a[k] == orcs[k].next_random_action(), (s, k)
            oo, rr, tr = orcs[k].step_keyed(int(a[k]))
            if tr:
                oo = orcs[k].reset_keyed()
            assert np.max(np.abs(o[k] - oo)) <= OBS_ATOL and _rel(r[k], rr) <= REW_RTOL and bool(d[k]) == tr, (s, k)
    env.close()


EXOTIC = [
    dict(use_ema_adr=0, duty_cycle=60.0),                                   # iot_sensors.py:245 non-EMA branch
    dict(data_generation_rate=0.0, duty_cycle=80.0, fill_lo=0.5, fill_hi=0.9),   # rate 0: AoI urgency branch uav_env.py:390-393
    dict(collection_duration=3.0, duty_cycle=100.0, power_hover=900.0, power_move=350.0),
    dict(fill_lo=0.9, fill_hi=1.4, duty_cycle=100.0, max_buffer_size=300.0),     # np.clip(fill, 0, 1) at iot_sensors.py:308
    dict(sf_thresholds=[-50.0, -64.0, -75.0, -90.0], rssi_threshold=-92.0, capture_threshold_db=1.5, duty_cycle=100.0,
         shadowing_std_db=7.0, noise_floor_dbm=-100.0),
    dict(alive_fraction=0.6, duty_cycle=50.0, penalty_unvisited=-77.0, penalty_starved=-13.0, starvation_cr_threshold=0.6),
    dict(capture_threshold_db=-3.0, duty_cycle=100.0),                      # negative margin: the strongest always captures
    dict(uav_altitude=30.0, sensor_height=1.5, wavelength=0.125, freq_mhz=2400.0, tx_power_dbm=20.0, duty_cycle=70.0),
]


@pytest.mark.parametrize("idx", range(len(EXOTIC)))
def test_exotic_configurations_match_oracle(idx):
    """Every constant of the path is a runtime parameter: unusual values of the IoTSensor / UAV /
    reward parameters, HIP vs oracle (no reference fixture reaches these through the env's kwargs)."""
    torch, U, O = _mods()
    E, steps, seed = 32, 140, 900 + idx
    over = dict(num_sensors=20, grid_size=(140, 140), max_steps=45, **EXOTIC[idx])
    want = O.trace_keyed(O.default_config(seed=seed, **over), E, steps)
    env = U.BatchedUAVEnv(E, seed=seed, **over)
    assert np.array_equal(env.reset().cpu().numpy(), want["reset_obs"])
    ro = env.rollout(steps, with_terminal=True)
    assert np.array_equal(ro["done"].cpu().numpy(), want["done"])
    assert np.array_equal(ro["actions"].cpu().numpy(), want["actions"])
    assert np.max(np.abs(ro["obs"].cpu().numpy() - want["obs"])) <= OBS_ATOL
    assert np.max(_rel(ro["reward"].cpu().numpy(), want["reward"])) <= REW_RTOL
    assert want["done"].any() and np.isfinite(want["reward"]).all()
    env.close()
