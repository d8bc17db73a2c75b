"""GPU box: single-step and fused throughput for small sensor counts under each admissible lane-group width
(UAVENV_LANE_GROUP), at several batch sizes.  One child process per setting (the knob is read at create)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, os, json
sys.path.insert(0, %r)
import torch, uavenv_amd as U
n, E = int(sys.argv[1]), int(sys.argv[2])
env = U.BatchedUAVEnv(E, num_sensors=n, seed=0)
env.reset()
for _ in range(100): env.step_random()
torch.cuda.synchronize()
ms = sorted(env.time_steps(500) for _ in range(3))[1]
F = 16
slab = torch.empty(F, E, env.obs_dim, dtype=torch.float32, device=env.device)
for _ in range(3): env.rollout(F, obs_out=slab)
torch.cuda.synchronize()
f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
f0.record()
for _ in range(30): env.rollout(F, obs_out=slab)
f1.record(); torch.cuda.synchronize()
print(json.dumps(dict(sensors=n, envs=E, lane_group=env.lane_stride, step_us=ms * 1e3, step_M=E / ms / 1e3,
                      fused_us_per_step=f0.elapsed_time(f1) / 30 / F * 1e3, fused_M=E * F * 30 / f0.elapsed_time(f1) / 1e3)))
''' % ROOT
for n in (10, 20):
    for E in (4096, 16384, 65536):
        for g in ([16, 32, 64] if n <= 16 else [32, 64]):
            env = dict(os.environ, UAVENV_LANE_GROUP=str(g))
            out = subprocess.check_output([sys.executable, "-c", child, str(n), str(E)], env=env).decode().strip().splitlines()[-1]
            d = json.loads(out)
            print("N=%2d E=%6d G=%2d  step %7.2f us %7.1f M/s   fused %7.2f us/step %7.1f M/s" % (
                d["sensors"], d["envs"], d["lane_group"], d["step_us"], d["step_M"], d["fused_us_per_step"], d["fused_M"]), flush=True)
