"""GPU box half of tools/attn_exp.sh: time uavenv_attention_features of every tools/_exp/lib_<name>.so (one child process per
library, UAVENV_LIB override): HIP-graph replays of 40 back-to-back launches, n_stack 10, microseconds per launch."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, os
sys.path.insert(0, %r)
import torch, uavenv_amd as U
from uavenv_amd.learner import AttentionFeatures
k = 10
m = AttentionFeatures(k).cuda().eval()
fused = U.FusedAttentionFeatures(m, k, "cuda:0")
out = []
for B in (16, 256, 4096, 16384):
    x = torch.rand(B, k * 153, device="cuda")
    if os.environ.get("ATTN_REAL", "1") == "1":        # realistic key masks: 20 sensors padded to 50 slots, most of them out of range
        tok = x[:, -150:].view(B, 50, 3)
        tok[:, 20:] = 0.0
        tok[:, :20, 2] *= (torch.rand(B, 20, device="cuda") > 0.6)
    else:
        x[:, -150:] *= (torch.rand(B, 150, device="cuda") > 0.3)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3): fused(x)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(40): y = fused(x)
        for _ in range(3): g.replay()
        side.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(5): g.replay()
            e1.record(side); side.synchronize()
            ts.append(e0.elapsed_time(e1) / 200 * 1e3)
    out.append("%%6d: %%6.2f us" %% (B, sorted(ts)[2]))
print("  ".join(out))
''' % ROOT
for name in sys.argv[1:]:
    env = dict(os.environ, UAVENV_LIB=os.path.join(ROOT, "tools", "_exp", f"lib_{name}.so"))
    r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
    print(f"{name:>12s}  {r.stdout.strip()}" + (("  ERR " + r.stderr.strip()[-300:]) if r.returncode else ""), flush=True)
