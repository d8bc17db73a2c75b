"""Dev tool (GPU box): which part of the learner's update breaks HIP graph capture?  Each stage runs in its own process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys
sys.path.insert(0, %r)
import torch, torch.nn as nn
import uavenv_amd as U
from uavenv_amd import learner as LR
stage = int(sys.argv[1])
env = U.BatchedUAVEnv(96, num_sensors=10, max_steps=9, grid_size=(60, 60), seed=5)
L = LR.DQNLearner(env, learning_rate=1e-2, buffer_size=96 * 40, batch_size=64, gamma=0.9, learning_starts=0,
                  target_update_interval=96 * 7, train_freq=2, gradient_steps=1, net_arch=(32, 16), n_stack=3,
                  total_timesteps=10**6, max_grad_norm=0.5, seed=3, reward_scale=1e-3, use_graphs=False, fused_update=False)   # (bisects the PyTorch update)
L.collect(20)
for _ in range(4): L.train(1)
torch.cuda.synchronize()
win = (torch.zeros((), dtype=torch.int64, device=env.device), torch.zeros((), dtype=torch.int64, device=env.device))
n, oldest = L.ring.window_state(); win[0].fill_(n); win[1].fill_(oldest)
params = list(L.q.parameters())
out = torch.zeros((), device=env.device)
g = torch.cuda.CUDAGraph()
if stage != 9: g.register_generator_state(L.gen)
gen = L.gen if stage != 9 else None
kw = {}
if stage in (31, 34):
    L.opt.zero_grad(set_to_none=True)
if stage in (32, 34):
    kw["capture_error_mode"] = "thread_local"
if stage in (33, 34, 35):
    s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        for _ in range(3):
            b_ = L.ring.sample_stacked(64, 3, generator=gen, window=win)
            l_ = LR.td_loss(L.q, L.q_target, b_, 0.9, 1e-3)
            L.opt.zero_grad(set_to_none=True); l_.backward(); L.opt.step()
    torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize()
    L.opt.zero_grad(set_to_none=True)
    kw["stream"] = s_
if stage == 35:
    kw["capture_error_mode"] = "relaxed"
real = stage
if stage > 30: stage = 3
D = 3 * env.obs_dim
SB = dict(obs=torch.randn(64, D, device=env.device), next_obs=torch.randn(64, D, device=env.device),
          action=torch.randint(0, 5, (64,), device=env.device), reward=torch.randn(64, device=env.device),
          valid=torch.ones(64, dtype=torch.bool, device=env.device))
Q2 = LR.QNetwork(env.obs_dim, 3, (32, 16)).to(env.device)
if real >= 41:
    s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        for _ in range(3):
            Q2(SB["obs"]).sum().backward(); L.q(SB["obs"]).sum().backward()
    torch.cuda.current_stream().wait_stream(s2); torch.cuda.synchronize()
    Q2.zero_grad(set_to_none=True); L.opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g, **kw):
    if stage == 0:
        j, slot, r, e = L.ring._draw(64, gen, win); out.copy_(slot.sum().float())
    elif real == 41:      # static random batch, the learner's loss
        loss = LR.td_loss(L.q, L.q_target, SB, 0.9, 1e-3); loss.backward()
    elif real == 42:      # ring batch, trivial loss
        batch = L.ring.sample_stacked(64, 3, generator=gen, window=win)
        loss = L.q(batch["obs"]).sum(); loss.backward()
    elif real == 43:      # static batch, trivial loss, the learner's network
        loss = L.q(SB["obs"]).sum(); loss.backward()
    elif real == 44:      # fresh network of the same shape, static batch
        loss = Q2(SB["obs"]).sum(); loss.backward()
    else:
        batch = L.ring.sample_stacked(64, 3, generator=gen, window=win)
        if stage == 1: out.copy_(batch["obs"].sum())
        if stage >= 2:
            loss = LR.td_loss(L.q, L.q_target, batch, 0.9, 1e-3)
            if stage == 2: out.copy_(loss.detach())
        if stage >= 3:
            L.opt.zero_grad(set_to_none=True); loss.backward()
        if stage >= 4:
            nn.utils.clip_grad_norm_(params, 0.5)
        if stage >= 5:
            L.opt.step()
        if stage >= 6:
            out.copy_(loss.detach())
torch.cuda.synchronize()
g.replay(); g.replay()
torch.cuda.synchronize()
print("stage", real, "ok", float(out))
''' % ROOT
for st in [int(a) for a in sys.argv[1:]] or [43, 3, 6]:
    r = subprocess.run([sys.executable, "-c", child, str(st)], capture_output=True, text=True)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "Fatal" in l][:2]
    print("stage %d rc %d %s %s" % (st, r.returncode, tail, err), flush=True)
