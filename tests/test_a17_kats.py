"""SURVEY row a17 (DomainRandEnv, agents/dqn/dqn.py:286-451) against KNOWN ANSWERS, not only oracle-vs-kernel.

`dqn.py` does not parse at this snapshot, so no run of the reference pins a17 ("parity unpinned").
tests/golden/kats_a17.json holds what can be pinned (made by tests/golden/make_kats_a17.py in the build container):
the SF a fresh DomainRandEnv sensor inherits and keeps, produced by the REAL IoTSensor class going through the calls of
dqn.py:340-358; the REAL RewardFunction's move reward that a zero-rate step must return without a Jain bonus
(`if rates:`, dqn.py:434-442); and the closed forms of `_jains` / the bonus / the proximity shaping.

Both the CPU oracle (not gpu) and the HIP path through the C ABI (gpu) are checked against the same numbers.
"""
import json
import os

import numpy as np
import pytest

KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats_a17.json")))
RANDOM_LAYOUT, FAR_START, PROX_SHAPING, JAIN_BONUS = 1, 2, 4, 8
N_INH, GRID_INH = 24, 1000          # 24 fresh sensors on a 1000 x 1000 grid: most are > 212 cells from the start (0, 0)


def _inherit_cfg(flags=RANDOM_LAYOUT, max_steps=3):
    return dict(num_sensors=N_INH, grid_size=(GRID_INH, GRID_INH), shadowing_std_db=0.0, flags=flags, max_steps=max_steps,
                seed=77)


def _old_layout(d0):
    """Sensor 0 at (d0, 0); the others anywhere (they do not matter for the inheritance)."""
    px = np.linspace(600.0, 900.0, N_INH).astype(np.float32)
    py = np.linspace(100.0, 950.0, N_INH).astype(np.float32)
    px[0], py[0] = d0, 0.0
    return px, py


def _check_inherited(case, pos_x, pos_y, sf, ux=0.0, uy=0.0):
    dist = np.hypot(pos_x - ux, pos_y - uy)
    far = dist > case["out_of_range_beyond_cells"]
    assert far.sum() >= 5, "layout has too few out-of-range sensors for this KAT"
    assert (sf[far] == case["fresh_out_of_range_sf_after_first_obs"]).all(), (case, sf[far])
    assert (sf[dist <= 5.0] == case["fresh_at_5_cells_sf_after_first_obs"]).all()


# ---------------------------------------------------------------------------------------------------------------
# CPU oracle
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", KATS["sf_inheritance_sigma0"], ids=lambda c: f"d0={c['old_sensor0_grid_dist']:g}")
def test_oracle_fresh_sensors_inherit_old_sensor0_sf(case):
    from oracle import oracle as O
    px, py = _old_layout(case["old_sensor0_grid_dist"])
    env = O.OracleEnv(O.default_config(**_inherit_cfg()), 5, px, py)
    env.reset_keyed()
    st = env.state()
    assert not np.array_equal(st["pos_x"], px), "RANDOM_LAYOUT must have replaced the layout"
    _check_inherited(case, st["pos_x"], st["pos_y"], st["sf"])
    # second episode: put the old sensor 0 back at the KAT distance, run to truncation, reset again
    env.e.pos_x[0], env.e.pos_y[0] = case["old_sensor0_grid_dist"], 0.0
    for _ in range(3):
        _, _, tr = env.step_keyed(4)
    assert tr
    env.reset_keyed()
    st = env.state()
    _check_inherited(case, st["pos_x"], st["pos_y"], st["sf"])


def _oracle_env(n, pos, **over):
    from oracle import oracle as O
    px = np.array([p[0] for p in pos], np.float32); py = np.array([p[1] for p in pos], np.float32)
    return O.OracleEnv(O.default_config(num_sensors=n, shadowing_std_db=0.0, seed=3, **over), 0, px, py)


def test_oracle_zero_rate_domain_rand_step_has_no_jain_bonus():
    env = _oracle_env(4, [(400, 400)] * 4, grid_size=(500, 500), data_generation_rate=0.0,
                      flags=RANDOM_LAYOUT | JAIN_BONUS)
    env.reset_keyed()
    _, r, _ = env.step_keyed(3)
    assert r == pytest.approx(KATS["zero_rate_domain_rand_step_reward"], rel=1e-13)


@pytest.mark.parametrize("k", range(len(KATS["jain_cases"])))
def test_oracle_jain_bonus_closed_form(k):
    case = KATS["jain_cases"][k]
    env = _oracle_env(4, [(400, 400), (410, 400), (420, 400), (430, 400)], grid_size=(500, 500),
                      data_generation_rate=2.0, flags=JAIN_BONUS)
    env.reset_keyed()
    for i in range(4):
        env.e.tx[i], env.e.gen[i], env.e.buffer[i] = case["tx"][i], case["gen_after_step"][i] - 2.0, 10.0
    _, r, _ = env.step_keyed(3)
    assert r == pytest.approx(case["step_reward"], rel=1e-13)


@pytest.mark.parametrize("k", range(len(KATS["shaping_cases"])))
def test_oracle_proximity_shaping_is_eta_times_distance_change(k):
    case = KATS["shaping_cases"][k]
    env = _oracle_env(1, [tuple(case["sensor"])], grid_size=(500, 500), flags=PROX_SHAPING)
    env.reset_keyed()
    assert env.e.prev_dist_nearest == case["d_prev"]
    _, r, _ = env.step_keyed(case["action"])
    assert r == pytest.approx(case["step_reward"], rel=1e-13)
    assert env.e.prev_dist_nearest == case["d_now"]


# ---------------------------------------------------------------------------------------------------------------
# HIP path (C ABI)
# ---------------------------------------------------------------------------------------------------------------
def _mods():
    import torch
    import uavenv_amd as U
    from uavenv_amd import _native as N
    return torch, U, N


def _set_sensor0(torch, env, N, d0):
    px = env.get_state(N.F_POS_X); py = env.get_state(N.F_POS_Y)
    px[:, 0] = d0; py[:, 0] = 0.0
    env.set_state(N.F_POS_X, px); env.set_state(N.F_POS_Y, py)


@pytest.mark.gpu
@pytest.mark.parametrize("auto_reset", [False, True])
@pytest.mark.parametrize("case", KATS["sf_inheritance_sigma0"], ids=lambda c: f"d0={c['old_sensor0_grid_dist']:g}")
def test_hip_fresh_sensors_inherit_old_sensor0_sf(case, auto_reset):
    """dqn.py:340-351 on the device, through uav_reset_kernel (manual reset) and through the in-step auto-reset."""
    torch, U, N = _mods()
    E = 5
    px, py = _old_layout(case["old_sensor0_grid_dist"])
    env = U.BatchedUAVEnv(E, auto_reset=auto_reset, sensor_positions=np.stack([px, py], -1), **_inherit_cfg())
    env.reset()
    st = env.sensor_state()
    for e in range(E):
        assert not np.array_equal(st["pos_x"][e], px)
        _check_inherited(case, st["pos_x"][e], st["pos_y"][e], st["sf"][e])
    # second episode: old sensor 0 back at the KAT distance, three collect steps to the step limit
    _set_sensor0(torch, env, N, case["old_sensor0_grid_dist"])
    acts = torch.full((E,), 4, dtype=torch.int32, device=env.device)
    for _ in range(3):
        _, _, done = env.step(acts)
    assert done.cpu().numpy().all()
    if not auto_reset:
        env.reset()
    rec = env.records()
    assert (rec["episode"] == 1).all() and (rec["current_step"] == 0).all()
    st = env.sensor_state()
    for e in range(E):
        _check_inherited(case, st["pos_x"][e], st["pos_y"][e], st["sf"][e])
    env.close()


@pytest.mark.gpu
def test_hip_zero_rate_domain_rand_step_has_no_jain_bonus():
    torch, U, N = _mods()
    env = U.BatchedUAVEnv(3, auto_reset=True, num_sensors=4, grid_size=(500, 500), data_generation_rate=0.0,
                          shadowing_std_db=0.0, flags=RANDOM_LAYOUT | JAIN_BONUS, seed=3)
    env.reset()
    _, r, _ = env.step(torch.full((3,), 3, dtype=torch.int32, device=env.device))
    assert np.allclose(r.cpu().numpy(), KATS["zero_rate_domain_rand_step_reward"], rtol=1e-13, atol=0)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_slots", [4, 20, 50])         # lane groups of 16, 32 and 64 lanes (per-env sensor count 4)
@pytest.mark.parametrize("k", range(len(KATS["jain_cases"])))
def test_hip_jain_bonus_closed_form(k, n_slots):
    torch, U, N = _mods()
    case = KATS["jain_cases"][k]
    E = 3
    pos = np.zeros((n_slots, 2), np.float32); pos[:, 0] = 400 + np.arange(n_slots); pos[:, 1] = 400
    env = U.BatchedUAVEnv(E, auto_reset=True, sensor_positions=pos, grid_size=(500, 500), data_generation_rate=2.0,
                          shadowing_std_db=0.0, flags=JAIN_BONUS, seed=3)
    if n_slots != 4:
        env.set_env_params(num_sensors=4)
    env.reset()
    S = env.lane_stride
    for field, vals in ((N.F_TX, case["tx"]), (N.F_GEN, [g - 2.0 for g in case["gen_after_step"]]), (N.F_BUFFER, [10.0] * 4)):
        t = env.get_state(field)
        t[:, :4] = torch.tensor(vals, dtype=torch.float64, device=env.device)
        env.set_state(field, t)
    assert S in (16, 32, 64)
    _, r, _ = env.step(torch.full((E,), 3, dtype=torch.int32, device=env.device))
    assert np.allclose(r.cpu().numpy(), case["step_reward"], rtol=1e-12, atol=0)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(KATS["shaping_cases"])))
def test_hip_proximity_shaping_is_eta_times_distance_change(k):
    torch, U, N = _mods()
    case = KATS["shaping_cases"][k]
    env = U.BatchedUAVEnv(2, auto_reset=True, sensor_positions=np.array([case["sensor"]], np.float32), grid_size=(500, 500),
                          shadowing_std_db=0.0, flags=PROX_SHAPING, seed=3)
    env.reset()
    assert (env.records()["prev_dist_nearest"] == case["d_prev"]).all()
    _, r, _ = env.step(torch.full((2,), case["action"], dtype=torch.int32, device=env.device))
    assert np.allclose(r.cpu().numpy(), case["step_reward"], rtol=1e-13, atol=0)
    assert (env.records()["prev_dist_nearest"] == case["d_now"]).all()
    env.close()
