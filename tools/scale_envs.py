"""GPU box: throughput of the single-step and fused paths as a function of the number of environments."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uavenv_amd as U
out = []
for E in (256, 1024, 4096, 16384, 65536, 262144):
    env = U.BatchedUAVEnv(E, num_sensors=50, seed=0)
    env.reset()
    for _ in range(50): env.step_random()
    ms = env.time_steps(300)
    F = 16
    slab = torch.empty(F, E, env.obs_dim, dtype=torch.float32, device=env.device)
    env.rollout(F, obs_out=slab); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps): env.rollout(F, obs_out=slab)
    torch.cuda.synchronize(); dtf = time.perf_counter() - t0
    out.append(dict(envs=E, kernel_us=ms * 1e3, env_steps_per_s=E / (ms * 1e-3), fused_env_steps_per_s=E * F * reps / dtf,
                    state_MB=E * 64 * 52 / 1e6))
    env.close(); del slab
print(json.dumps(out))
