"""Dev tool (GPU box): timing-only ablation builds of the step kernel (each variant computes WRONG
numbers on purpose; only its kernel time is read).  Prints average launch time per variant."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "-reinforcement-learning-for-dynamic-uav-energy-efficient-path-planning-in-iot-sensor-networks._amd")
sys.path.insert(0, PKG)
import build as _build  # noqa: E402  (the package's build.py: one place for the compiler flags)
HIPCC_FLAGS = [f for f in _build.FLAGS if f != "-Wall"]
OUT = os.path.join(ROOT, "gpurun_out", "ablate")
os.makedirs(OUT, exist_ok=True)
VARIANTS = [("base", []), ("no_log10", ["-DUAV_ABL_LOG10"]), ("philox6", ["-DUAV_ABL_PHILOX=6"]), ("no_normal", ["-DUAV_ABL_NORMAL"])]
VARIANTS += [(n, f.split()) for n, f in (x.split("=", 1) for x in sys.argv[1:])]
procs = []
for name, flags in VARIANTS:
    lib = os.path.join(OUT, f"lib_{name}.so")
    procs.append((name, lib, subprocess.Popen(
        ["hipcc"] + HIPCC_FLAGS + flags +
        ["-o", lib, os.path.join(PKG, "csrc", "uavenv_kernels.hip"), os.path.join(PKG, "csrc", "uavenv_capi.hip"), os.path.join(PKG, "csrc", "uavenv_attention.hip"), os.path.join(PKG, "csrc", "uavenv_replay.hip")])))
for name, lib, p in procs:
    assert p.wait() == 0, name
child = r'''
import sys, os
sys.path.insert(0, %r)
import uavenv_amd
from uavenv_amd import _native as N
N.LIB_PATH = sys.argv[1]
import torch, uavenv_amd as U
env = U.BatchedUAVEnv(int(os.environ.get("E", 4096)), num_sensors=int(os.environ.get("NS", 50)), seed=0)
env.reset()
for _ in range(300): env.step_random()
torch.cuda.synchronize()
ts = [env.time_steps(1000) for _ in range(3)]
print("%%-12s kernel us: %%s" %% (sys.argv[2], " ".join("%%.2f" %% (t * 1e3) for t in ts)))
''' % ROOT
for name, lib, _ in procs:
    subprocess.check_call([sys.executable, "-c", child, lib, name])
