"""GPU box: fused UAVAttentionExtractor forward vs the eager PyTorch module (same weights, batch 4096)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import uavenv_amd as U
from train_dqn import AttentionFeatures
out = {}
for k in (4, 10):
    m = AttentionFeatures(k).cuda().eval()
    fused = U.FusedAttentionFeatures(m, k, "cuda:0")
    x = torch.rand(4096, k * 153, device="cuda")
    def t(f, n=200):
        for _ in range(20): f(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f(x)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    with torch.no_grad():
        out[f"n_stack_{k}"] = {"torch_eager_us": t(m), "fused_hip_us": t(fused)}
print(json.dumps(out))
