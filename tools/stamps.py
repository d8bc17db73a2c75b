"""Dev tool (GPU box): build the diagnostic (-DUAVENV_STAMPS) library, run the headline workload and
print the in-kernel timeline of the step kernel: per-wave lifetime, start skew, clock, concurrency."""
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
                       "-DUAVENV_STAMPS", "-o", LIB, os.path.join(PKG, "csrc", "uavenv_kernels.hip"),
                       os.path.join(PKG, "csrc", "uavenv_capi.hip"), os.path.join(PKG, "csrc", "uavenv_attention.hip"), os.path.join(PKG, "csrc", "uavenv_replay.hip")])
import uavenv_amd  # noqa: E402
from uavenv_amd import _native as N  # noqa: E402
N.LIB_PATH = LIB
import torch  # noqa: E402
import uavenv_amd as U  # noqa: E402

E, n = int(os.environ.get("E", 4096)), int(os.environ.get("NS", 50))
env = U.BatchedUAVEnv(E, num_sensors=n, seed=0)
L = N.lib()
L.uavenv_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
nw = (E * env.lane_stride + 63) // 64
stamps = torch.zeros((1 << 20) + (E + 64) * 16, dtype=torch.int64, device=env.device)   # per-wave records, then the phase stamps at word 1 << 20 (UAV_PHASE)
assert nw * 8 <= (1 << 20)
env.reset()
for _ in range(300):
    env.step_random()
torch.cuda.synchronize()
L.uavenv_debug_set_stamps(env._h, C.c_void_p(stamps.data_ptr()))
env.step_random()
torch.cuda.synchronize()
s = stamps.cpu().numpy()[: nw * 8].reshape(nw, 8).astype(np.uint64)
t0, t1, r0, r1 = s[:, 0].astype(np.int64), s[:, 1].astype(np.int64), s[:, 2].astype(np.int64), s[:, 3].astype(np.int64)
life = t1 - t0
rl = (r1 - r0)
T0 = r0.min()
print("waves", nw, "shader-clock lifetime cycles: mean %.0f p50 %.0f p95 %.0f max %.0f" % (life.mean(), np.median(life), np.percentile(life, 95), life.max()))
print("realtime(100MHz) lifetime us: mean %.2f p50 %.2f max %.2f" % (rl.mean() / 100, np.median(rl) / 100, rl.max() / 100))
print("kernel span (first start -> last end) us: %.2f" % ((r1.max() - T0) / 100))
print("start skew us: p50 %.2f p95 %.2f max %.2f" % (np.median(r0 - T0) / 100, np.percentile(r0 - T0, 95) / 100, (r0 - T0).max() / 100))
clk = life.sum() / max(1, rl.sum()) * 100e6
print("effective shader clock ~ %.2f GHz" % (clk / 1e9))
act = s[:, 6].astype(np.int64)
for a in range(5):
    m = act == a
    if m.any():
        print(" action", a, "waves", int(m.sum()), "mean life cycles %.0f" % life[m].mean())
xcc = s[:, 5] & 0xF
hw = s[:, 4]
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
key = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
uk, cnt = np.unique(key, return_counts=True)
print("distinct (xcc,se,sh,cu):", len(uk), "waves per CU: min %d max %d" % (cnt.min(), cnt.max()))
# concurrency profile
ev = np.concatenate([np.stack([r0 - T0, np.ones_like(r0)], 1), np.stack([r1 - T0, -np.ones_like(r1)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1])
for t_us in (0.5, 1, 2, 4, 6, 8, 10, 12, 14, 16):
    i = np.searchsorted(ev[:, 0], t_us * 100)
    print("  t=%5.1f us  resident waves %d" % (t_us, conc[min(i, len(conc) - 1)]))

# per-SIMD composition: does the launch tail come from SIMDs that host several collect-action waves?
wave_in_cu = (hw >> 0) & 0xF
skey = key * 4 + simd.astype(np.int64)
import collections
by = collections.defaultdict(list)
for i in range(nw):
    by[int(skey[i])].append(i)
rows = []
for kx, idxs in by.items():
    nc = int(sum(act[i] == 4 for i in idxs))
    rows.append((nc, len(idxs), max((r1[i] - T0) / 100 for i in idxs), np.mean([life[i] for i in idxs])))
rows = np.array(rows)
print("SIMDs", len(rows), "waves/SIMD min %d max %d" % (rows[:, 1].min(), rows[:, 1].max()))
for nc in range(5):
    m = rows[:, 0] == nc
    if m.any():
        print("  SIMDs with %d collect waves: %4d   finish time us: mean %.2f max %.2f   mean wave life %.0f cycles" %
              (nc, m.sum(), rows[m, 2].mean(), rows[m, 2].max(), rows[m, 3].mean()))
late = np.argsort(r1)[-12:]
for i in late:
    print("  late wave: end %.2f us life %d cycles action %d  simd-collects %d" %
          ((r1[i] - T0) / 100, life[i], act[i], sum(act[j] == 4 for j in by[int(skey[i])])))
