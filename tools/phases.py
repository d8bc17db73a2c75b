"""Dev tool (GPU box): phase timeline of one step launch from the diagnostic (-DUAVENV_STAMPS) build.
Phases (shader-clock cycles since the wave's first instruction): 0 step_once entry (unit known), 1 action known (record
arrived), 2 sensor rows arrived + ageing reduced, 3 path loss, 4 noise drawn, 5 collect/truncation done, 6 observation
written, 7 record epilogue done, end = after the state stores were issued."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "-reinforcement-learning-for-dynamic-uav-energy-efficient-path-planning-in-iot-sensor-networks._amd")
sys.path.insert(0, PKG)
import build as _build  # noqa: E402  (the package's build.py: one place for the compiler flags)
HIPCC_FLAGS = [f for f in _build.FLAGS if f != "-Wall"]
LIB = os.path.join(ROOT, "gpurun_out", "libuavenv_hip_stamps.so")
os.makedirs(os.path.dirname(LIB), exist_ok=True)
subprocess.check_call(["hipcc"] + HIPCC_FLAGS + [
                       "-DUAVENV_STAMPS", "-o", LIB] + [os.path.join(PKG, "csrc", f) for f in
                       ("uavenv_kernels.hip", "uavenv_capi.hip", "uavenv_attention.hip", "uavenv_replay.hip")])
import uavenv_amd  # noqa: E402
from uavenv_amd import _native as N  # noqa: E402
N.LIB_PATH = LIB
import torch  # noqa: E402
import uavenv_amd as U  # noqa: E402

for E in [int(x) for x in os.environ.get("ES", "256,4096").split(",")]:
    env = U.BatchedUAVEnv(E, num_sensors=50, seed=0)
    L = N.lib()
    L.uavenv_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    assert E * 8 <= (1 << 20)
    stamps = torch.zeros((1 << 20) + (E + 64) * 16, dtype=torch.int64, device=env.device)   # padded environments stamp too
    env.reset()
    for _ in range(200):
        env.step_random()
    torch.cuda.synchronize()
    L.uavenv_debug_set_stamps(env._h, C.c_void_p(stamps.data_ptr()))
    env.step_random()
    torch.cuda.synchronize()
    raw = stamps.cpu().numpy()
    w = raw[: E * 8].reshape(E, 8)            # per wave slot (block*16 + wave): t0, t1, r0, r1, hwid, xcc, action
    ph = raw[(1 << 20): (1 << 20) + E * 16].reshape(E, 16)    # per environment
    # map env -> wave slot start time: the wave that stepped env e recorded phases for env e; its t0 is in w[slot] where
    # slot is unknown under balancing, so match through the end stamp: use per-block sets instead (same block)
    wpb = 16 if E >= 4096 else 4               # waves per workgroup (launch_step: 16-wave workgroups once every CU gets 16)
    t0_blk = w[:, 0].reshape(-1, wpb).min(axis=1)
    t0 = np.repeat(t0_blk, wpb)[:E]
    act = np.zeros(E, dtype=np.int64)
    # the action per env: recover from the aux-free path -> use env.last actions via records? use the wave record instead
    rel = ph[:, :8] - t0[:, None]
    print(f"E={E}: phase medians (cycles after the earliest wave start of the workgroup; ~2.3 cycles/ns)")
    names = ["entry", "action", "rows+age", "pathloss", "noise", "collect", "observe", "epilogue"]
    for i, nme in enumerate(names):
        print("   %-9s p50 %6.0f  p90 %6.0f  max %6.0f" % (nme, np.median(rel[:, i]), np.percentile(rel[:, i], 90), rel[:, i].max()))
    life = w[:, 1] - w[:, 0]
    print("   wave lifetime p50 %.0f p90 %.0f max %.0f" % (np.median(life), np.percentile(life, 90), life.max()))
    env.close()
