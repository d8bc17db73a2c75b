"""Dev tool (GPU box): count device-vs-oracle noise mismatches over many (env, step, lane) draws."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import uavenv_amd as U  # noqa: E402
from oracle import oracle as O  # noqa: E402

env = U.BatchedUAVEnv(4096, num_sensors=50, seed=77)
env.reset()
bad = tot = 0
L = O.lib()
for it in range(5):
    env.step_random()
    st, rt = env.dump_noise()
    st = st.cpu().numpy()
    rec = env.records()
    for k in range(0, 4096, 8):
        want = np.zeros((6, 50), np.float32)
        L.orc_noise_step_tape(77, k, int(rec["episode"][k]), int(rec["current_step"][k]) + 1, 50, O._fp(want))
        bad += int((st[k, :, :50] != want).sum())
        tot += 300
print("noise mismatches", bad, "of", tot)
