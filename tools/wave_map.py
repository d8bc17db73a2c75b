"""Dev tool (GPU box): where do the wavefronts of a workgroup land?  Builds the diagnostic library with a given
workgroup size (UAV_BLOCK) and prints, from HW_ID, the wave -> (XCC, CU, SIMD) mapping pattern of the step kernel."""
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
BLOCK = int(os.environ.get("UAV_BLOCK", 1024))
LIB = os.path.join(ROOT, "gpurun_out", f"libuavenv_hip_stamps_b{BLOCK}.so")
os.makedirs(os.path.dirname(LIB), exist_ok=True)
subprocess.check_call(["hipcc"] + HIPCC_FLAGS + [
                       "-DUAVENV_STAMPS", f"-DUAV_BLOCK={BLOCK}", "-o", LIB] + [os.path.join(PKG, "csrc", f) for f in
                       ("uavenv_kernels.hip", "uavenv_capi.hip", "uavenv_attention.hip", "uavenv_replay.hip")])
import uavenv_amd  # noqa: E402
from uavenv_amd import _native as N  # noqa: E402
N.LIB_PATH = LIB
import torch  # noqa: E402
import uavenv_amd as U  # noqa: E402

E = 4096
env = U.BatchedUAVEnv(E, num_sensors=50, seed=0)
L = N.lib()
L.uavenv_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
stamps = torch.zeros(E * 8 + 64, dtype=torch.int64, device=env.device)
env.reset()
for _ in range(100):
    env.step_random()
torch.cuda.synchronize()
ts = [env.time_steps(500) for _ in range(3)]
print("block", BLOCK, "kernel us:", " ".join("%.2f" % (t * 1e3) for t in ts))
L.uavenv_debug_set_stamps(env._h, C.c_void_p(stamps.data_ptr()))
env.step_random()
torch.cuda.synchronize()
s = stamps.cpu().numpy()[: E * 8].reshape(E, 8).astype(np.uint64)
hw = s[:, 4]; xcc = (s[:, 5] & 0xF).astype(np.int64)
simd = ((hw >> 4) & 0x3).astype(np.int64); cu = ((hw >> 8) & 0xF).astype(np.int64)
sh = ((hw >> 12) & 1).astype(np.int64); se = ((hw >> 13) & 0x7).astype(np.int64)
cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu
wpb = BLOCK // 64
nb = E // wpb
same_cu = sum(len(set(cukey[b * wpb:(b + 1) * wpb])) == 1 for b in range(nb))
print("blocks", nb, "blocks entirely on one CU:", same_cu)
pat = {}
for b in range(nb):
    key = tuple(int(x) for x in simd[b * wpb:(b + 1) * wpb])
    pat[key] = pat.get(key, 0) + 1
for k, v in sorted(pat.items(), key=lambda kv: -kv[1])[:6]:
    print("  simd pattern by wave-in-block", k, "x", v)
blocks_per_cu = {}
for b in range(nb):
    blocks_per_cu.setdefault(int(cukey[b * wpb]), []).append(b)
cnt = np.array([len(v) for v in blocks_per_cu.values()])
print("CUs used", len(blocks_per_cu), "blocks per CU min/max", cnt.min(), cnt.max())
for k in list(blocks_per_cu)[:4]:
    print("  CU", hex(k), "blocks", blocks_per_cu[k])
print("xcc of blocks 0..15:", [int(xcc[b * wpb]) for b in range(min(16, nb))])
