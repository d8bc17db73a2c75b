"""GPU box half of tools/exp.sh: time every tools/_exp/lib_<name>.so at 256 and 4096 environments."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, os
sys.path.insert(0, %r)
import uavenv_amd
from uavenv_amd import _native as N
N.LIB_PATH = sys.argv[1]
import torch, uavenv_amd as U
out = []
for E in [int(x) for x in os.environ.get("ES", "256,4096").split(",")]:
    env = U.BatchedUAVEnv(E, num_sensors=int(os.environ.get("NS", 50)), seed=0)
    env.reset()
    for _ in range(300): env.step_random()
    torch.cuda.synchronize()
    ts = sorted(env.time_steps(1000) for _ in range(5))
    F = 16
    slab = torch.empty(F, E, env.obs_dim, dtype=torch.float32, device=env.device)
    for _ in range(3): env.rollout(F, obs_out=slab)
    torch.cuda.synchronize()
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0.record()
    for _ in range(50): env.rollout(F, obs_out=slab)
    f1.record(); torch.cuda.synchronize()
    # given actions (the acting path of a learner): 64 launches replayed as one HIP graph, a fresh action array per step
    acts = torch.randint(0, 5, (64, E), dtype=torch.int32, device=env.device)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for k in range(3): env.step(acts[k])
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for k in range(64): env.step(acts[k])
        for _ in range(3): g.replay()
        side.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(side)
        for _ in range(20): g.replay()
        a1.record(side); side.synchronize()
    t_act = a0.elapsed_time(a1) / 20 / 64 * 1e3
    out.append("E=%%5d step %%6.2f us (min %%.2f)  given actions %%6.2f us  rollout/16 %%6.2f us/step" %% (E, ts[2] * 1e3, ts[0] * 1e3, t_act, f0.elapsed_time(f1) / 50 / F * 1e3))
    env.close()
print("%%-14s %%s" %% (sys.argv[2], "   |   ".join(out)))
''' % ROOT
for name in sys.argv[1:]:
    lib = os.path.join(ROOT, "tools", "_exp", f"lib_{name}.so")
    subprocess.check_call([sys.executable, "-c", child, lib, name])
