"""GPU box: PCIe-inclusive rates of the host-buffer boundaries (never used for bench `value`):
uavenv_step_host (C ABI, synchronous numpy in/out) and UAVVecEnv.step (SB3 contract)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import uavenv_amd as U  # noqa: E402
from uavenv_amd import _native as N  # noqa: E402

E, n, steps = 4096, 50, 300
env = U.BatchedUAVEnv(E, num_sensors=n, seed=0)
env.reset()
L = N.lib()
obs = np.zeros((E, env.obs_dim), np.float32); rew = np.zeros(E); done = np.zeros(E, np.uint8)
acts = np.random.default_rng(0).integers(0, 5, size=(steps, E)).astype(np.int32)
# (plain integer addresses: building ctypes pointer objects per call costs more than the call)
po, pr, pd, pa, stride = obs.ctypes.data, rew.ctypes.data, done.ctypes.data, acts.ctypes.data, acts.strides[0]
torch.cuda.synchronize()
for s in range(20):
    N.check(L.uavenv_step_host(env._h, pa + s * stride, po, pr, pd, None), env._h)
t0 = time.perf_counter()
for s in range(steps):
    rc = L.uavenv_step_host(env._h, pa + s * stride, po, pr, pd, None)
    if rc:
        N.check(rc, env._h)
dt = time.perf_counter() - t0
out = {"uavenv_step_host": {"env_steps_per_s": E * steps / dt, "us_per_vector_step": dt / steps * 1e6,
                            "bytes_over_pcie_per_step": E * (4 + env.obs_dim * 4 + 8 + 1)}}
env.close()
for name, kw in (("UAVVecEnv.step", {}), ("UAVVecEnv.step(host_copies=False)", dict(host_copies=False))):
    venv = U.UAVVecEnv(E, num_sensors=n, seed=0, **kw)
    venv.reset()
    for s in range(20):
        venv.step_async(acts[s]); venv.step_wait()
    t0 = time.perf_counter()
    for s in range(steps):
        venv.step_async(acts[s]); venv.step_wait()
    dt = time.perf_counter() - t0
    out[name] = {"env_steps_per_s": E * steps / dt, "us_per_vector_step": dt / steps * 1e6}
    venv.close()
print(json.dumps(out))
