"""GPU box: the plainest possible driver of the step kernel for profiler passes -- E environments x 50 sensors, random
policy, `steps` back-to-back launches (uavenv_time_steps) -- so that counter averages are not mixed with bench.py's
other launches.   python3 tools/step_driver.py <envs> [steps] [sensors]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import uavenv_amd as U  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50
env = U.BatchedUAVEnv(E, num_sensors=n, seed=0)
env.reset()
for _ in range(50):
    env.step_random()
ms = env.time_steps(steps)
torch.cuda.synchronize()
print(f"envs {E} sensors {n}: {ms * 1e3:.3f} us per launch, {E / (ms * 1e-3) / 1e6:.1f} M env-steps/s")
env.close()
