"""Dev tool (GPU box): host-side cost of one bench step (Python + ctypes + hipLaunchKernel), measured on a tiny batch so
the GPU is never the limit, and the enqueue time of the headline loop (before the final synchronise)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uavenv_amd as U
from uavenv_amd.replay import TransitionRing

for E in (16, 4096):
    env = U.BatchedUAVEnv(E, num_sensors=50, seed=0)
    ring = TransitionRing(64, E, env.obs_dim, env.device)
    ring.attach(env)
    env.reset()
    for _ in range(300):
        env.step_random(obs_out=ring.local_obs_slot()); ring.commit()
    torch.cuda.synchronize()
    K = 3000
    t0 = time.perf_counter()
    for _ in range(K):
        env.step_random(obs_out=ring.local_obs_slot()); ring.commit()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"E={E}: enqueue {1e6 * (t1 - t0) / K:.2f} us/step, total {1e6 * (t2 - t0) / K:.2f} us/step")
    t0 = time.perf_counter()
    for _ in range(K):
        env.step_random()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"E={E}: bare step_random enqueue {1e6 * (t1 - t0) / K:.2f} us/step, total {1e6 * (t2 - t0) / K:.2f} us/step")
    env.close()
