"""GPU box: do the gradients of AttentionFeatures.forward come out of a HIP-graph replay the way eager autograd computes them, on
inputs that changed since the capture?  (Found with this: the gradient of a bias passed to torch.baddbmm -- a [B, T, E] -> [E]
reduction -- replays STALE on PyTorch 2.10 / ROCm 7; the first replay on unchanged inputs looks right.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, uavenv_amd as U
from uavenv_amd import learner as LR
torch.manual_seed(0)
m = LR.AttentionFeatures(10).cuda()
m.fused_core = os.environ.get("FUSED_CORE", "1") == "1"
params = list(m.parameters()); names = [n for n, _ in m.named_parameters()]
n = sum(p.numel() for p in params)
BATCH = int(os.environ.get("BATCH", 256))
x = torch.rand(BATCH, 1530, device="cuda"); up = torch.randn(BATCH, 128, device="cuda")
flat = torch.zeros(n, device="cuda")
def step():
    for p in params: p.grad = None
    (m(x) * up).sum().backward()
    torch.cat([p.grad.reshape(-1) for p in params], out=flat)
for _ in range(3): step()
torch.cuda.synchronize()
ref = flat.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
flat.zero_()
g.replay(); torch.cuda.synchronize()
off = 0
for nm, p in zip(names, params):
    k = p.numel(); a, b = flat[off:off + k], ref[off:off + k]; off += k
    rel = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
    if rel > 1e-5: print("graph vs eager:", nm, rel)
print("done; total rel", float((flat - ref).abs().max()))
x.mul_(0.5)                       # other inputs through the same graph
for p in params: p.grad = None
(m(x) * up).sum().backward(); ref2 = torch.cat([p.grad.reshape(-1) for p in params])
g.replay(); torch.cuda.synchronize()
print("second replay max abs diff", float((flat - ref2).abs().max()))
off = 0
for nm, p in zip(names, params):
    k = p.numel(); a, b = flat[off:off + k], ref2[off:off + k]; off += k
    rel = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
    if rel > 1e-5: print("second replay:", nm, rel, float(b.abs().max()))
# third: replay twice in a row on yet another input
x.add_(0.1)
for p in params: p.grad = None
(m(x) * up).sum().backward(); ref3 = torch.cat([p.grad.reshape(-1) for p in params])
g.replay(); g.replay(); torch.cuda.synchronize()
print("third max abs diff", float((flat - ref3).abs().max()))
