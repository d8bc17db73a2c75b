"""Build-container-only: known answers for SURVEY row a17 (DomainRandEnv, agents/dqn/dqn.py:286-451).

`dqn.py` itself cannot be imported at this snapshot (unterminated module docstring -> SyntaxError at :136, and
stable_baselines3 is absent), so a17 stays "parity unpinned" by a run of the reference.  What CAN be pinned is pinned
here with the reference's own classes, which DomainRandEnv only composes:

  * SF inheritance (dqn.py:340-351): the REAL `IoTSensor` goes through exactly the calls the discarded
    `super().reset()` makes on the old sensor 0 -- `reset()` (iot_sensors.py:305), then the reset observation's
    `update_spreading_factor(uav.start_position)` (uav_env.py:654) -- and a fresh REAL `IoTSensor(spreading_factor=
    s0.spreading_factor)` far outside radio range then runs its first `update_spreading_factor`: the sticky branch
    (iot_sensors.py:251-255) must keep the inherited value.  sigma = 0 makes it deterministic.
  * the move reward a zero-rate DomainRand step must return unchanged (`if rates:` guard, dqn.py:434-442): the REAL
    `RewardFunction.calculate_movement_reward`.
  * Jain's index / the bonus and the proximity shaping: closed forms evaluated by hand below, each a transcription of
    three lines of dqn.py text (`_jains` :446-451, bonus :442, shaping :419-425) in plain Python / numpy float32.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as R  # noqa: E402

uav_env, iot = R.import_reference()
from rewards.reward_function import RewardFunction  # noqa: E402

out = {}

# ---- SF inheritance ---------------------------------------------------------------------------------------
cases = []
for d0 in (5.0, 30.0, 100.0, 150.0, 400.0):
    s0 = iot.IoTSensor(position=(d0, 0.0), sensor_id=0, shadowing_std_db=0.0)
    s0.reset(initial_buffer_fill=0.37)                      # uav_env.py:409-410 (the fill is irrelevant here)
    s0.update_spreading_factor((0.0, 0.0))                  # uav_env.py:654 inside the discarded reset observation
    inherited = int(s0.spreading_factor)
    far = iot.IoTSensor(position=(900.0, 900.0), sensor_id=1, spreading_factor=s0.spreading_factor,
                        shadowing_std_db=0.0)               # dqn.py:346-358
    far.update_spreading_factor((0.0, 0.0))                 # first observation of the new episode
    near = iot.IoTSensor(position=(3.0, 4.0), sensor_id=2, spreading_factor=s0.spreading_factor, shadowing_std_db=0.0)
    near.update_spreading_factor((0.0, 0.0))
    cases.append(dict(old_sensor0_grid_dist=d0, inherited_sf=inherited,
                      fresh_out_of_range_sf_after_first_obs=int(far.spreading_factor),
                      fresh_at_5_cells_sf_after_first_obs=int(near.spreading_factor),
                      out_of_range_beyond_cells=212))       # kats.json rssi_sigma0: 211 cells -> -85.011 dB < -85
out["sf_inheritance_sigma0"] = cases

# ---- zero-rate step: no Jain bonus, the plain move reward -------------------------------------------------
rf = RewardFunction(penalty_data_loss=-1.0, reward_urgency_reduction=20.0, penalty_battery=-0.5, reward_movement=10.0)
move_ok = rf.calculate_movement_reward(True, 274.0 - (274.0 - 500.0 / 3600))
out["move_reward_ok"] = move_ok
out["zero_rate_domain_rand_step_reward"] = move_ok          # rates == [] -> `if rates:` false -> nothing added


# ---- Jain's index closed form (dqn.py:446-451) and the bonus (dqn.py:442) ----------------------------------
def jains(rates):                                           # transcription of the 4-line static method
    n = len(rates)
    s1 = sum(rates)
    s2 = sum(x ** 2 for x in rates)
    return (s1 ** 2) / (n * s2) if n > 0 and s2 > 0 else 1.0


jc = []
for tx, gen in (([0.0, 0.0, 0.0, 0.0], [100.0, 200.0, 400.0, 800.0]),         # nothing transmitted: s2 == 0 -> J = 1
                ([50.0, 50.0, 100.0, 0.0], [100.0, 200.0, 400.0, 800.0]),      # rates 50, 25, 25, 0 % -> J = 2/3
                ([10.0, 20.0, 30.0, 40.0], [100.0, 200.0, 300.0, 400.0]),      # all 10 % -> J = 1
                ([128.0, 0.0, 0.0, 0.0], [256.0, 512.0, 64.0, 32.0])):         # one sensor only -> J = 1/4
    rates = [t / g for t, g in zip(tx, gen)]
    j = jains([r * 100 for r in rates])
    n = len(tx)
    jc.append(dict(tx=tx, gen_after_step=gen, jain=j, bonus=0.5 * (j - 0.5) / n, step_reward=move_ok + 0.5 * (j - 0.5) / n))
out["jain_cases"] = jc

# ---- proximity shaping (dqn.py:406-425): reward += 2.0 * (d_prev - d_now), distances = float32 norms -----------
sc = []
sensor = np.array([30.0, 0.0], dtype=np.float32)
for name, act, pos in (("RIGHT", 3, (1.0, 0.0)), ("UP", 0, (0.0, 1.0))):
    d_prev = float(np.linalg.norm(sensor - np.array([0.0, 0.0], dtype=np.float32)))
    d_now = float(np.linalg.norm(sensor - np.array(pos, dtype=np.float32)))
    sc.append(dict(action=act, name=name, sensor=[30.0, 0.0], d_prev=d_prev, d_now=d_now,
                   step_reward=move_ok + 2.0 * (d_prev - d_now)))
out["shaping_cases"] = sc

json.dump(out, open(os.path.join(HERE, "kats_a17.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
