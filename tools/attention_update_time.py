"""GPU box: microseconds per captured DQN update with the attention extractor (4096 x 50, n_stack 10, reference hyper-parameters),
for the folded training forward and for the module form (nn.MultiheadAttention), hybrid update and pure PyTorch update."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uavenv_amd as U
from uavenv_amd import learner as LR

out = {}
for form in ("folded", "module"):
    for fused in (True, False):
        if form == "module":
            LR.AttentionFeatures.forward_folded = LR.AttentionFeatures.__dict__.get("forward_folded", LR.AttentionFeatures.forward)
            LR.AttentionFeatures.forward = LR.AttentionFeatures.forward_module
        elif "forward_folded" in LR.AttentionFeatures.__dict__:
            LR.AttentionFeatures.forward = LR.AttentionFeatures.forward_folded
        env = U.BatchedUAVEnv(4096, num_sensors=50, pad_sensors=50, grid_size=(500, 500), seed=0)
        hp = dict(LR.REFERENCE_HYPERPARAMS, n_stack=10, extractor="attention", learning_starts=0, total_timesteps=10**9)
        L = LR.DQNLearner(env, seed=0, use_graphs=True, fused_update=fused, **hp)
        for _ in range(8):
            L.collect(L.train_freq); L.train()
        torch.cuda.synchronize()
        assert L._train_graph is not None
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            L.train(50)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 50 * 1e6)
        out[f"{form}_{'hybrid' if fused else 'torch'}_update_us"] = round(best, 1)
        print(form, fused, round(best, 1), flush=True)
        env.close()
print(json.dumps(out))
