"""GPU box: microseconds per launch of each of the fused DQN update's eight product launches (reference shapes: batch 256,
612 -> 512 -> 512 -> 256 -> 5), each replayed 200 times back to back inside one HIP graph; next to it the same launch with the
reduction length cut to 16 (what the launch, the epilogue and the store cost without the products).  UAVENV_LIB selects a build."""
import copy, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from uavenv_amd import _native as N
from uavenv_amd.learner import QNetwork
from uavenv_amd.mlp_update import FusedMLPUpdate

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, D, k = 256, 153, 4
q, qt = QNetwork(D, k).to(dev), QNetwork(D, k).to(dev)
U = FusedMLPUpdate(q, qt, B, 0.99, 10.0)
batch = dict(obs=torch.randn(B, D * k, device=dev), next_obs=torch.randn(B, D * k, device=dev), action=torch.randint(0, 5, (B,), device=dev),
             reward=torch.randn(B, device=dev), valid=torch.ones(B, dtype=torch.bool, device=dev))
launches = []
orig = U._launch


def clone(p):
    c = N.UavGemm(); C.memmove(C.byref(c), C.byref(p), C.sizeof(N.UavGemm)); return c



def rec(first, second, stream):
    launches.append((clone(first), None if second is None else clone(second)))
    orig(first, second, stream)
U._launch = rec
U.backward(batch)
torch.cuda.synchronize()
L = N.lib()
REPS = 200


def time_pair(first, second):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        st = C.c_void_p(s.cuda_stream)
        for _ in range(3):
            L.uavenv_gemm_f32(C.byref(first), None if second is None else C.byref(second), st)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for _ in range(REPS):
                rc = L.uavenv_gemm_f32(C.byref(first), None if second is None else C.byref(second), st)
                assert rc == 0
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    return best


def short(p):
    if p is None:
        return None
    c = clone(p); c.K = min(p.K, 16); return c


names = ["fwd l0 (online|target)", "fwd l1", "fwd l2", "fwd l3", "bwd l3 (dW|dx)", "bwd l2", "bwd l1", "bwd l0 (dW)"]
rows, total = [], 0.0
for name, (a, b) in zip(names, launches):
    t, t0 = time_pair(a, b), time_pair(short(a), short(b))
    fl = 2.0 * a.M * a.N * a.K + (0 if b is None else 2.0 * b.M * b.N * b.K)
    rows.append(dict(launch=name, first=[a.M, a.N, a.K], second=None if b is None else [b.M, b.N, b.K], us=round(t, 2), us_k16=round(t0, 2),
                     tflops=round(fl / t * 1e-6, 1)))
    total += t
    print(rows[-1], flush=True)
print(json.dumps(dict(total_us=round(total, 1), lib=os.environ.get("UAVENV_LIB", "default"))))

# ---- what a row stride that is not a power of two buys (fwd l1: both operands 512 floats per row) -----------------------------
a, b = launches[1]
for pad in (0, 16, 32, 100):
    ld = 512 + pad
    bufs = []
    def padded(p):
        c = clone(p)
        A = torch.randn(p.M, ld, device=dev); Bm = torch.randn(p.N, ld, device=dev); bufs.extend([A, Bm])
        c.A, c.B, c.a_sm, c.b_sn = A.data_ptr(), Bm.data_ptr(), ld, ld
        return c
    print(dict(launch="fwd l1, operand rows padded", row_floats=ld, us=round(time_pair(padded(a), padded(b)), 2)), flush=True)
