"""SURVEY rows a17 + f4 on the GPU: the HIP path replays trajectories recorded from the REAL `DomainRandEnv`
(agents/dqn/dqn.py:177-451, tests/golden/make_golden_domainrand.py) from nothing but (seed, global env index, actions) --
no tape: grid draw, inherited SF, fresh layout, far start, padded observation, proximity shaping, Jain bonus, truncation
and every `last_episode_stats` value come out of the kernels' own Philox stream.

Fixed sensor counts 10 / 20 / 30 / 40 run in lane groups of 16 / 32 / 32 / 64 lanes.  Tolerances as in test_gpu_parity.py:
observations <= 1e-6 (bit-identical in practice), rewards and statistics <= 1e-9 relative, everything discrete identical.
"""
import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu

OBS_ATOL = 1e-6
RTOL = 1e-9
STAT_FLOATS = ("total_generated", "total_collected", "total_lost", "battery_remaining", "ndr", "fairness_std", "jains_index",
               "data_efficiency", "bytes_per_wh")


def _mods():
    import torch
    import uavenv_amd as U
    return torch, U


def _rel(a, b):
    return abs(a - b) / max(1.0, abs(b))


def _check_stats(got, row, keys, where):
    want = dict(zip(keys, row))
    for k in STAT_FLOATS:
        assert _rel(float(got[k]), want[k]) <= RTOL, (where, k, got[k], want[k])
    assert tuple(got["grid_size"]) == (int(want["grid_w"]), int(want["grid_h"])), where
    assert got["num_sensors"] == int(want["num_sensors"]), where
    assert got["time_to_coverage"] is None          # dqn.py:302 clears the attribute before :330 reads it


def _check_info(info, fx, s, where):
    """The 17 keys of the reference's _get_info() (uav_env.py:676-700) as the REAL step() returned them at step s."""
    keys = fx["meta"]["info_keys"]
    want = dict(zip(keys, fx["info_scalars"][s]))
    for k in ("sensors_collected", "current_step", "high_urgency_sensors", "capture_effect_triggers", "boundary_hits", "edge_steps"):
        assert info[k] == int(want[k]), (where, k, info[k], want[k])
    assert bool(info["is_alive"]) == bool(want["is_alive"]), where
    for k in ("battery", "battery_percent", "total_reward", "total_data_collected", "coverage_percentage", "last_step_bytes_collected"):
        assert _rel(float(info[k]), want[k]) <= RTOL, (where, k, info[k], want[k])
    for k in ("max_urgency", "avg_urgency"):                      # float32 AoI urgencies (uav_env.py:386-394)
        assert abs(float(info[k]) - want[k]) <= 1e-6 * max(1.0, abs(want[k])), (where, k, info[k], want[k])
    assert np.array_equal(np.asarray(info["uav_position"], np.float32), fx["info_uav_position"][s]), where
    assert np.allclose(info["sensor_collection_ratios"], fx["info_sensor_collection_ratios"][s], rtol=1e-12, atol=1e-15), where


def _kernel_stats_as_dict(st, max_battery):
    """UavEnvEpisodeStats (include/uavenv.h) -> the dqn.py:316-331 keys, the way vec_env.py derives them."""
    tg, tc = float(st["total_generated"]), float(st["total_collected"])
    used = max_battery - float(st["battery_remaining"])
    return {"total_generated": tg, "total_collected": tc, "total_lost": float(st["total_lost"]),
            "battery_remaining": float(st["battery_remaining"]),
            "ndr": int(st["sensors_visited"]) / int(st["num_sensors"]) * 100,
            "fairness_std": float(st["fairness_std"]), "jains_index": float(st["jains_index"]),
            "grid_size": (int(st["grid_w"]), int(st["grid_h"])), "num_sensors": int(st["num_sensors"]),
            "data_efficiency": (tc / tg * 100) if tg > 0 else 0.0, "bytes_per_wh": (tc / used) if used > 0 else 0.0,
            "time_to_coverage": None}


@pytest.mark.parametrize("auto_reset", [False, True])
@pytest.mark.parametrize("name", G.domainrand_fixture_names())
def test_hip_replays_real_domainrand(name, auto_reset):
    """Batch of 3 environments with consecutive global indices; the middle one is the fixture's.  Manual reset goes through
    uav_reset_kernel, auto-reset through the step kernel's own reset block (which also writes UavEnvEpisodeStats)."""
    torch, U = _mods()
    fx = G.load(name)
    meta = fx["meta"]
    n, E, k = meta["n"], 3, 1
    env = U.BatchedUAVEnv(E, auto_reset=auto_reset, env_index_base=meta["env_index"] - k, **G.domainrand_overrides(meta))
    fps = 5 if meta["base"].get("include_sensor_positions") else 3
    assert env.lane_stride == (16 if n <= 16 else 32 if n <= 32 else 64) and env.obs_dim == 3 + 50 * fps
    dev = env.device
    ep = 0

    def check_episode_start(obs):
        rec = env.records()[k]
        ss = env.sensor_state(k)
        assert np.array_equal(obs, fx["ep_reset_obs"][ep]), (name, ep)
        assert (rec["grid_w"], rec["grid_h"]) == tuple(fx["ep_grid"][ep]), (name, ep)
        assert (rec["start_x"], rec["start_y"]) == tuple(fx["ep_start"][ep]) == (rec["uav_x"], rec["uav_y"]), (name, ep)
        assert np.array_equal(np.stack([ss["pos_x"], ss["pos_y"]], -1), fx["ep_pos"][ep]), (name, ep)
        assert rec["episode"] == ep and rec["current_step"] == 0

    check_episode_start(env.reset().cpu().numpy()[k])
    for s, a in enumerate(fx["actions"]):
        o, r, d = env.step(torch.full((E,), int(a), dtype=torch.int32, device=dev))
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        tr = bool(fx["truncated"][s])
        step_obs = env.terminal_obs.cpu().numpy()[k] if (tr and auto_reset) else o[k]
        assert np.max(np.abs(step_obs - fx["obs"][s])) <= OBS_ATOL, (name, s)
        assert _rel(r[k], fx["reward"][s]) <= RTOL, (name, s, r[k], fx["reward"][s])
        assert bool(d[k]) == tr, (name, s)
        if not (tr and auto_reset):
            assert np.array_equal(env.sensor_state(k)["sf"], fx["sf"][s]), (name, s)
        if tr:
            if auto_reset:
                st = env.episode_stats()[k]
                assert st["valid"] == 1 and st["episode"] == ep
                want = dict(zip(meta["stat_keys"], fx["ep_stats"][ep]))
                assert st["length"] == int(want["length"]) and st["first_full_coverage_step"] == int(want["first_full_coverage_step"])
                _check_stats(_kernel_stats_as_dict(st, env.cfg.max_battery), fx["ep_stats"][ep], meta["stat_keys"], (name, ep))
                ep += 1
                check_episode_start(o[k])
            else:
                ep += 1
                check_episode_start(env.reset(mask=torch.from_numpy(d).to(dev)).cpu().numpy()[k])
        elif d.any() and not auto_reset:                     # a neighbour ended its episode (battery-limited case)
            env.reset(mask=torch.from_numpy(d).to(dev))
    assert ep == len(fx["ep_stats"])
    ss, rec = env.sensor_state(k), env.records()[k]
    for key in ("sf", "visited", "data_collected"):
        assert np.array_equal(ss[key], fx["final_" + key]), key
    for key in ("buffer", "gen", "tx", "lost"):
        assert np.allclose(ss[key], fx["final_" + key], rtol=1e-12, atol=1e-9), key
    assert rec["uav_x"] == fx["final_uav_x"] and rec["uav_y"] == fx["final_uav_y"] and rec["current_step"] == fx["final_step"]
    assert abs(rec["battery"] - fx["final_battery"]) < 1e-9 and _rel(rec["total_reward"], float(fx["final_total_reward"])) < RTOL
    for key in ("capture_triggers", "boundary_hits", "edge_steps"):
        assert rec[key] == fx["final_" + key], key
    env.close()


@pytest.mark.parametrize("name", ["domainrand_s2_n20", "domainrand_s4_n40", "domainrand_s3_n30_lowbatt", "domainrand_s2_n20_fps5"])
def test_gym_domainrand_env_matches_the_real_class(name):
    """The single-environment mirror (`gym_env.DomainRandEnv`, the reference's own constructor and call sequence):
    reset() / step() values and `last_episode_stats` as the real class produced them."""
    torch, U = _mods()
    fx = G.load(name)
    meta = fx["meta"]
    env = U.DomainRandEnv(fixed_num_sensors=meta["n"], curriculum_stage=meta["stage"], base_config=meta["base"],
                          seed=meta["seed"], env_index=meta["env_index"])
    assert env.observation_space.shape == (253 if meta["base"].get("include_sensor_positions") else 153,) and env.last_episode_stats is None
    ep = 0
    obs, _ = env.reset()
    assert np.array_equal(obs, fx["ep_reset_obs"][0])
    for s, a in enumerate(fx["actions"]):
        obs, r, term, trunc, info = env.step(int(a))
        assert term is False and trunc == bool(fx["truncated"][s])
        assert np.max(np.abs(obs - fx["obs"][s])) <= OBS_ATOL and _rel(r, fx["reward"][s]) <= RTOL, (name, s)
        assert len(info) == 17
        _check_info(info, fx, s, (name, s))
        if trunc:
            obs, _ = env.reset()
            _check_stats(env.last_episode_stats, fx["ep_stats"][ep], meta["stat_keys"], (name, ep))
            ep += 1
            assert np.array_equal(obs, fx["ep_reset_obs"][ep])
            assert tuple(env.grid_size) == tuple(fx["ep_grid"][ep])
    assert ep == len(fx["ep_stats"])
    env.close()


@pytest.mark.parametrize("name", ["domainrand_s0_n10", "domainrand_s4_n20", "domainrand_s3_n30_lowbatt"])
def test_vec_env_last_episode_stats_match_the_real_class(name):
    """What CurriculumCallback reads (dqn.py:921-984): `infos[i]["last_episode_stats"]`, `get_attr("last_episode_stats")`
    and Monitor's `info["episode"]` of the SB3-style vectorised environment, against the real class's values."""
    torch, U = _mods()
    fx = G.load(name)
    meta = fx["meta"]
    over = G.domainrand_overrides(meta)
    E, k = 4, 2
    for drop in ("flags", "pad_sensors", "grid_choices", "grid_size"):      # domain_rand=True sets these itself
        over.pop(drop)
    venv = U.UAVVecEnv(E, domain_rand=True, curriculum_stage=meta["stage"], env_index_base=meta["env_index"] - k, **over)
    obs = venv.reset()
    assert np.array_equal(obs[k], fx["ep_reset_obs"][0])
    ep, ret = 0, 0.0
    for s, a in enumerate(fx["actions"]):
        obs, rew, dones, infos = venv.step(np.full(E, int(a)))
        ret += fx["reward"][s]
        assert bool(dones[k]) == bool(fx["truncated"][s])
        if dones[k]:
            info = infos[k]
            _check_stats(info["last_episode_stats"], fx["ep_stats"][ep], meta["stat_keys"], (name, ep))
            assert venv.get_attr("last_episode_stats", k)[0] is info["last_episode_stats"]
            assert np.max(np.abs(info["terminal_observation"] - fx["obs"][s])) <= OBS_ATOL
            assert info["TimeLimit.truncated"] is True
            want = dict(zip(meta["stat_keys"], fx["ep_stats"][ep]))
            assert info["episode"]["l"] == int(want["length"]) and _rel(info["episode"]["r"], ret) <= RTOL
            # the terminal step's own _get_info() keys (uav_env.py:676-700) that BestByMetricCallback reads (dqn.py:1150-1155)
            assert _rel(info["total_data_collected"], want["total_collected"]) <= RTOL
            assert _rel(info["battery"], want["battery_remaining"]) <= RTOL
            assert _rel(info["coverage_percentage"], want["ndr"]) <= RTOL
            _check_info(info, fx, s, (name, s))                       # all 17 keys of the TERMINAL step, from the terminal snapshot
            ep += 1
            ret = 0.0
            assert np.array_equal(obs[k], fx["ep_reset_obs"][ep])
        else:
            assert np.max(np.abs(obs[k] - fx["obs"][s])) <= OBS_ATOL
    assert ep == len(fx["ep_stats"])
    venv.close()


def test_episode_stats_without_rates():
    """dqn.py:322-329 guards on the device: no sensor generated data -> fairness_std 0, `_jains([])` = 1; the derived
    data_efficiency is 0."""
    torch, U = _mods()
    env = U.BatchedUAVEnv(2, auto_reset=True, num_sensors=4, grid_size=(100, 100), data_generation_rate=0.0, max_steps=3,
                          flags=1, seed=5)
    env.reset()
    a = torch.full((2,), 4, dtype=torch.int32, device=env.device)
    for _ in range(3):
        _, _, d = env.step(a)
    assert d.cpu().numpy().all()
    for st in env.episode_stats():
        got = _kernel_stats_as_dict(st, env.cfg.max_battery)
        assert st["valid"] == 1 and got["fairness_std"] == 0.0 and got["jains_index"] == 1.0
        assert got["data_efficiency"] == 0.0 and got["total_generated"] == 0.0 and got["ndr"] == 0.0
        assert _rel(got["battery_remaining"], 274.0 - 3 * 700.0 / 3600) <= 1e-12
    env.close()
