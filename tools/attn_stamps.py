"""GPU box: per-wavefront phase timeline of workgroup 0 of the attention kernel (library built with -DATTN_STAMPS, see
tools/attn_exp.sh): shader-clock cycles (s_memtime; ~2.4 GHz: the whole kernel is ~31 k cycles = 13 us) from the workgroup's first stamp.
    UAVENV_LIB=tools/_exp/lib_stamps.so python3 tools/attn_stamps.py [batch]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, uavenv_amd as U
from uavenv_amd import _native as N
from uavenv_amd.learner import AttentionFeatures
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = 10
m = AttentionFeatures(k).cuda().eval()
fused = U.FusedAttentionFeatures(m, k, "cuda:0")
x = torch.rand(B, k * 153, device="cuda")
x[:, -150:] *= (torch.rand(B, 150, device="cuda") > 0.3)
for _ in range(5): fused(x)
torch.cuda.synchronize()
raw = (C.c_ulonglong * 256)()
L = C.CDLL(os.environ["UAVENV_LIB"])
assert L.uavenv_debug_attn_stamps(raw) == 0
t = np.array(raw, dtype=np.int64).reshape(16, 16)[:, :10]
t0 = t[:, 0].min()
names = ["start", "inputs", "uav+LN1", "Wq+QKfold", "scores", "softmax", "mix", "barrier", "V,Wo,LN2", "fusion+store"]
print("batch", B, " (shader-clock cycles from the first wavefront's start)")
print("wave " + " ".join(f"{n:>12s}" for n in names))
for w in range(16):
    print(f"{w:4d} " + " ".join(f"{(v - t0):12d}" for v in t[w]))
