"""Dev tool (GPU box half of `EXP_CMD="python3 tools/obs_stride_probe.py" tools/exp.sh a= b=-DUAV_OBS_STRIDE=160`): step launches
writing their observations into 64 different slots of a slab (as the replay ring does), rows 160 floats apart in the slab
either way (the library decides whether it uses that stride), replayed as one HIP graph."""
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
for E in (256, 4096):
    env = U.BatchedUAVEnv(E, num_sensors=50, seed=0)
    env.reset()
    for _ in range(100): env.step_random()
    torch.cuda.synchronize()
    S = 64
    slab = torch.zeros(S, E, 160, dtype=torch.float32, device=env.device)
    res = []
    for mode in ("same", "slots"):
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            f, po, st = env._hot_args(env.obs)
            def launch(k):
                ptr = slab[k if mode == "slots" else 0].data_ptr()
                rc = env.L.uavenv_step_random(env._h, f[5], ptr, f[1], f[2], f[3], f[4], torch.cuda.current_stream().cuda_stream)
                assert rc == 0
            for k in range(3): launch(k)
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for k in range(S): launch(k)
            for _ in range(3): g.replay()
            side.synchronize()
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(side)
            for _ in range(30): g.replay()
            a1.record(side); side.synchronize()
        res.append("%%s %%6.2f us" %% (mode, a0.elapsed_time(a1) / 30 / S * 1e3))
    out.append("E=%%5d " %% E + "  ".join(res))
    env.close()
print("%%-10s %%s" %% (sys.argv[2], "   |   ".join(out)))
''' % ROOT
for name in sys.argv[1:]:
    lib = os.path.join(ROOT, "tools", "_exp", f"lib_{name}.so")
    subprocess.check_call([sys.executable, "-c", child, lib, name])
