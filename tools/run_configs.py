"""GPU box: BASELINE.json configs 2 and 5 on one GPU.

  config 2: 4096 envs x N=20, 500x500, random policy, step()-only throughput + parity of the first 256 envs x
            500 steps against the CPU oracle;
  config 5: the domain-randomised sweep grid in {250,500,1000} x N in {10,20,50}, 4096 envs per combination here
            (the 8-GPU run shards 65536 envs over these nine combinations), per-combination steps/s.

Prints one JSON document (also written to gpurun_out/configs.json)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import uavenv_amd as U  # noqa: E402
from oracle import oracle as O  # noqa: E402


def throughput(E, n, grid, steps=1500, fused=16, **kw):
    env = U.BatchedUAVEnv(E, num_sensors=n, grid_size=(grid, grid), seed=0, **kw)
    env.reset()
    for _ in range(100):
        env.step_random()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        env.step_random()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = env.time_steps(500)
    slab = torch.empty(fused, E, env.obs_dim, dtype=torch.float32, device=env.device)
    env.rollout(fused, obs_out=slab)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    reps = max(1, steps // fused)
    for _ in range(reps):
        env.rollout(fused, obs_out=slab)
    torch.cuda.synchronize()
    dtf = time.perf_counter() - t1
    out = dict(envs=E, sensors=n, grid=grid, lane_group=env.lane_stride, obs_dim=env.obs_dim,
               env_steps_per_s=E * steps / dt, us_per_vector_step=dt / steps * 1e6, kernel_us=kern_ms * 1e3,
               fused_env_steps_per_s=E * fused * reps / dtf, fused_steps_per_launch=fused)
    env.close()
    return out


def parity_c2():
    E, n, steps = 256, 20, 500
    env = U.BatchedUAVEnv(E, num_sensors=n, grid_size=(500, 500), seed=0)
    want = O.trace_keyed(O.default_config(num_sensors=n, grid_size=(500, 500), seed=0), E, steps)
    obs = env.reset().cpu().numpy()
    max_obs = float(np.max(np.abs(obs - want["reset_obs"])))
    max_rew = 0.0
    flags_equal = True
    for s in range(steps):
        o, r, d = env.step_random()
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        max_obs = max(max_obs, float(np.max(np.abs(o - want["obs"][s]))))
        max_rew = max(max_rew, float(np.max(np.abs(r - want["reward"][s]) / np.maximum(1.0, np.abs(want["reward"][s])))))
        flags_equal &= bool(np.array_equal(d, want["done"][s]))
    env.close()
    return dict(envs=E, sensors=n, steps=steps, max_abs_obs_diff=max_obs, max_rel_reward_diff=max_rew,
                truncation_flags_identical=flags_equal, tolerance="1e-5 (north_star)")


if __name__ == "__main__":
    res = dict(config2=dict(throughput=throughput(4096, 20, 500), parity_vs_oracle=parity_c2()), config5=[])
    for grid in (250, 500, 1000):
        for n in (10, 20, 50):
            res["config5"].append(throughput(4096, n, grid, steps=800))
    res["config3_shape"] = throughput(4096, 50, 500, steps=800, pad_sensors=50)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))
