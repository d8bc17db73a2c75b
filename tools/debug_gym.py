import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uavenv_amd as U
from oracle import oracle as O
kw = dict(grid_size=(120, 120), num_sensors=20, max_steps=60, sensor_duty_cycle=70.0)
env = U.UAVEnvironment(seed=5, env_index=3, **kw)
orc = O.OracleEnv(O.default_config(grid_size=(120, 120), num_sensors=20, max_steps=60, duty_cycle=70.0, seed=5), 3)
obs, info = env.reset(); oo = orc.reset_keyed()
rng = np.random.default_rng(0)
for s in range(12):
    a = int(rng.integers(0, 5))
    obs, r, term, trunc, info = env.step(a)
    oo, rr, tr = orc.step_keyed(a)
    st = orc.state()
    print(s, 'a', a, 'batt', info['battery'], st['battery'], 'pos', info['uav_position'], st['uav_x'], st['uav_y'], 'r', r, rr, 'bh', info['boundary_hits'], st['boundary_hits'])
