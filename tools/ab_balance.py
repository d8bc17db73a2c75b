"""Dev tool (GPU box): A/B the SIMD load balancing of the step kernel (UAVENV_NO_BALANCE) on the built library."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, os
sys.path.insert(0, %r)
import torch, uavenv_amd as U
E = int(os.environ.get("E", 4096))
env = U.BatchedUAVEnv(E, num_sensors=int(os.environ.get("NS", 50)), seed=0)
env.reset()
for _ in range(300): env.step_random()
torch.cuda.synchronize()
ts = [env.time_steps(1000) for _ in range(3)]
a = torch.randint(0, 5, (E,), dtype=torch.int32, device=env.device)
for _ in range(50): env.step(a)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(1000): env.step(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 1000
print("%%-12s random-policy kernel us: %%s   given-actions step us (wall): %%.2f" %% (sys.argv[1], " ".join("%%.2f" %% (t * 1e3) for t in ts), dt * 1e6))
''' % ROOT
for name, val in (("balanced", "0"), ("home", "1")):
    subprocess.check_call([sys.executable, "-c", child, name], env=dict(os.environ, UAVENV_NO_BALANCE=val))
