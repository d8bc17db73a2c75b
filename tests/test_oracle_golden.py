"""CPU: the C oracle must reproduce every golden vector captured from the real reference
(tests/golden/make_golden.py) -- observations bit-exact, rewards to 1e-12 relative, truncation flags
and spreading factors identical, final state equal."""
import json
import os

import numpy as np
import pytest

import golden_util as G
import tape as T
from oracle import oracle as O

NAMES = G.fixture_names()


def test_fixtures_present():
    assert len(NAMES) >= 18


@pytest.mark.parametrize("name", NAMES)
def test_oracle_replays_reference_fixture(name):
    fx = G.load(name)
    meta = fx["meta"]
    n, seed = meta["n"], meta["tape_seed"]
    cfg = O.default_config(**G.config_overrides(meta))
    px, py = T.positions(seed, 0, n, meta["grid"][0], meta["grid"][1])
    env = O.OracleEnv(cfg, 0, px, py)
    episode = 0
    obs = env.reset_tape(T.reset_tape(seed, 0, episode, n))
    assert np.array_equal(obs, fx["reset_obs"][0])
    for s, a in enumerate(fx["actions"]):
        obs, r, tr = env.step_tape(int(a), T.step_tape(seed, 0, s, n))
        assert np.array_equal(obs, fx["obs"][s]), (name, s)
        assert abs(r - fx["reward"][s]) <= 1e-12 * max(1.0, abs(fx["reward"][s])), (name, s, r, fx["reward"][s])
        assert tr == bool(fx["truncated"][s]), (name, s)
        assert np.array_equal(env.state()["sf"], fx["sf"][s]), (name, s)
        if tr:
            episode += 1
            obs = env.reset_tape(T.reset_tape(seed, 0, episode, n))
            assert np.array_equal(obs, fx["reset_obs"][episode])
    st = env.state()
    for k in ("sf", "visited", "data_collected", "uav_x", "uav_y", "step", "capture_triggers", "boundary_hits",
              "edge_steps"):
        assert np.array_equal(st[k], fx["final_" + k]), k
    for k in ("buffer", "gen", "tx", "lost", "battery", "total_reward", "total_collected", "last_bytes"):
        assert np.allclose(st[k], fx["final_" + k], rtol=1e-12, atol=1e-9), k
    # EMA RSSI carries the reference platform's 1-ulp float32 log10 fuzz (oracle/uavenv_oracle.h)
    assert np.allclose(st["avg_rssi"], fx["final_avg_rssi"], rtol=0, atol=4e-5, equal_nan=True)


def test_scalar_known_answers():
    kats = json.load(open(os.path.join(G.GOLDEN, "kats.json")))
    cfg = O.default_config()
    L = O.lib()
    for k in kats["rssi_sigma0"]:
        got = L.orc_rssi_deterministic(cfg, float(k["grid_dist"]), 0.0, 0.0, 0.0)
        # <= 1 float32 ulp of the path loss (3.8e-6 dB below 64 dB, 7.6e-6 below 128, 1.5e-5 above)
        assert abs(got - k["rssi"]) <= 1.6e-5, (k, got)


def test_move_and_collect_reward_kats():
    """SURVEY 8a a3/a12 known answers, through the oracle's step on a hand-built state."""
    kats = json.load(open(os.path.join(G.GOLDEN, "kats.json")))
    cfg = O.default_config(num_sensors=3, grid_size=(500, 500), shadowing_std_db=0.0)
    far = np.array([400.0, 410.0, 420.0], np.float32)
    env = O.OracleEnv(cfg, 0, far, far)
    env.reset_tape(np.zeros((3, 3), np.float32))
    zero = np.zeros((6, 3), np.float32)
    # blocked move (DOWN at y=0) then a legal one: rewards minus the (zero) data-loss term
    _, r_blocked, _ = env.step_tape(1, zero)
    _, r_ok, _ = env.step_tape(0, zero)
    assert abs(r_blocked - kats["move_reward_blocked"]) < 1e-12
    assert abs(r_ok - kats["move_reward_ok"]) < 1e-12
    # nothing collected with buffers (500, 250, 1000): ageing adds 2.2 first, so pre-load b - 2.2
    e = env.e
    for i, b in enumerate((500.0, 250.0, 1000.0)):
        e.buffer[i] = b - 2.2 if b < 1000 else b - 2.2
        e.lost[i] = 0.0
    _, r, _ = env.step_tape(4, np.ones((6, 3), np.float32))   # u = 1 -> nobody transmits
    assert abs(r - kats["collect_reward_nothing_500_250_1000"]) < 1e-9


def test_invalid_action_raises_after_ageing():
    cfg = O.default_config(num_sensors=2)
    env = O.OracleEnv(cfg, 0, np.zeros(2, np.float32), np.zeros(2, np.float32))
    env.reset_tape(np.zeros((3, 2), np.float32))
    gen0 = env.state()["gen"].copy()
    with pytest.raises(ValueError):
        env.step_tape(7, np.zeros((6, 2), np.float32))
    st = env.state()
    assert st["step"] == 1 and np.all(st["gen"] > gen0)      # uav_env.py:439-468 order


@pytest.mark.parametrize("name", G.policy_fixture_names())
def test_oracle_policies_replay_reference_agents(name):
    """The oracle's restatement of NearestSensorGreedy / MaxThroughputGreedyV2 picks the action the REAL
    reference agent picked at every step (fixtures from tests/golden/make_golden_policies.py)."""
    fx = G.load(name)
    meta = fx["meta"]
    n, seed, pid = meta["n"], meta["tape_seed"], meta["policy_id"]
    cfg = O.default_config(**G.config_overrides(meta))
    px, py = T.positions(seed, 0, n, meta["grid"][0], meta["grid"][1])
    env = O.OracleEnv(cfg, 0, px, py)
    episode = 0
    assert np.array_equal(env.reset_tape(T.reset_tape(seed, 0, 0, n)), fx["reset_obs"][0])
    for s in range(meta["steps"]):
        a, obs, r, tr = env.step_policy_tape(pid, T.step_tape(seed, 0, s, n))
        assert a == int(fx["actions"][s]), (name, s)
        assert np.array_equal(obs, fx["obs"][s]) and tr == bool(fx["truncated"][s]), (name, s)
        assert abs(r - fx["reward"][s]) <= 1e-12 * max(1.0, abs(fx["reward"][s])), (name, s)
        if tr:
            episode += 1
            assert np.array_equal(env.reset_tape(T.reset_tape(seed, 0, episode, n)), fx["reset_obs"][episode])
    assert len(G.policy_fixture_names()) >= 5


# ---------------------------------------------------------------------------------------------------------------
# SURVEY rows a17 + f4: the REAL DomainRandEnv (tests/golden/make_golden_domainrand.py), keyed noise
# ---------------------------------------------------------------------------------------------------------------
def test_domainrand_fixtures_present():
    names = G.domainrand_fixture_names()
    assert len(names) >= 12
    finished = sum(int(G.load(n)["truncated"].sum()) for n in names)
    assert finished >= 24                      # every fixture holds >= 2 finished episodes


def check_domainrand_stats(got, row, keys, rtol=1e-12):
    """`got`: dict with the dqn.py:316-331 keys; `row`: the fixture's values in `keys` order."""
    want = dict(zip(keys, row))
    for k in ("total_generated", "total_collected", "total_lost", "battery_remaining", "ndr", "fairness_std", "jains_index",
              "data_efficiency", "bytes_per_wh"):
        assert abs(got[k] - want[k]) <= rtol * max(1.0, abs(want[k])), (k, got[k], want[k])
    assert tuple(got["grid_size"]) == (int(want["grid_w"]), int(want["grid_h"]))
    assert got["num_sensors"] == int(want["num_sensors"])


@pytest.mark.parametrize("name", G.domainrand_fixture_names())
def test_oracle_replays_real_domainrand(name):
    """reset (grid draw, inherited SF, fresh layout, far start, padded obs), step (shaping + Jain bonus), truncation and
    `last_episode_stats` of the real class, from nothing but (seed, env_index, actions)."""
    fx = G.load(name)
    meta = fx["meta"]
    env = O.OracleEnv(O.default_config(**G.domainrand_overrides(meta)), meta["env_index"])
    ep = 0

    def open_episode():
        obs = env.reset_keyed()
        st = env.state()
        assert np.array_equal(obs, fx["ep_reset_obs"][ep]), (name, ep)
        assert (st["grid_w"], st["grid_h"]) == tuple(fx["ep_grid"][ep])
        assert (st["start_x"], st["start_y"]) == tuple(fx["ep_start"][ep]) == (st["uav_x"], st["uav_y"])
        assert np.array_equal(np.stack([st["pos_x"], st["pos_y"]], -1), fx["ep_pos"][ep])

    open_episode()
    for s, a in enumerate(fx["actions"]):
        obs, r, tr = env.step_keyed(int(a))
        assert np.array_equal(obs, fx["obs"][s]), (name, s)
        assert abs(r - fx["reward"][s]) <= 1e-12 * max(1.0, abs(fx["reward"][s])), (name, s, r, fx["reward"][s])
        assert tr == bool(fx["truncated"][s]), (name, s)
        assert np.array_equal(env.state()["sf"], fx["sf"][s]), (name, s)
        if tr:
            got = env.episode_stats()
            check_domainrand_stats(got, fx["ep_stats"][ep], meta["stat_keys"])
            ep += 1
            open_episode()
    assert ep == len(fx["ep_stats"]) >= 2
    st = env.state()
    for k in ("sf", "visited", "data_collected", "uav_x", "uav_y", "step", "capture_triggers", "boundary_hits", "edge_steps"):
        assert np.array_equal(st[k], fx["final_" + k]), k
    for k in ("buffer", "gen", "tx", "lost", "battery", "total_reward", "total_collected", "last_bytes"):
        assert np.allclose(st[k], fx["final_" + k], rtol=1e-12, atol=1e-9), k


def test_oracle_episode_stats_without_rates():
    """dqn.py:322-329 guards: no sensor generated data -> fairness_std 0.0, `_jains([])` = 1.0, data_efficiency 0.0; and
    bytes_per_wh 0.0 when no battery was used."""
    cfg = O.default_config(num_sensors=4, grid_size=(100, 100), data_generation_rate=0.0, flags=1)
    env = O.OracleEnv(cfg, 0)
    env.reset_keyed()
    st = env.episode_stats()
    assert st["rated"] == 0 and st["fairness_std"] == 0.0 and st["jains_index"] == 1.0
    assert st["data_efficiency"] == 0.0 and st["bytes_per_wh"] == 0.0 and st["ndr"] == 0.0
