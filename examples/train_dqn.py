#!/usr/bin/env python3
"""BASELINE config 3 in one process: 4096 environments x 50 sensor slots driving a DQN learner with the
reference's hyper-parameters (agents/dqn/dqn.py:1077-1099), everything resident on one MI355X:

    HIP step kernel -> device frame stack (uavenv_frame_stack) -> epsilon-greedy Q-network (torch) -> actions
                    -> transition ring (observations written in place, terminal pool) -> stacked sampling -> learner

This is the SB3-free form of `DQN("MlpPolicy", VecFrameStack(DummyVecEnv([...]), 4))` (dqn.py:1276-1288); with
stable-baselines3 installed the same environments are available to SB3 itself through `uavenv_amd.UAVVecEnv`.
`--extractor attention` uses the architecture of the reference's UAVAttentionExtractor (dqn.py:548-650: UAV-state
MLP over all frames + one-query cross-attention over the 50 sensor slots of the newest frame, ghost and
out-of-range slots masked).  Model code is plain PyTorch-ROCm; it is not part of the hot path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import uavenv_amd as U  # noqa: E402


class AttentionFeatures(nn.Module):
    """Same layer shapes as dqn.py:548-650 (embed 64, 4 heads, 128 features)."""

    def __init__(self, n_stack, frame=153, slots=50, embed=64, heads=4, features=128):
        super().__init__()
        self.n_stack, self.frame, self.slots = n_stack, frame, slots
        self.uav = nn.Sequential(nn.Linear(3 * n_stack, embed), nn.LayerNorm(embed), nn.ReLU())
        self.sensor = nn.Linear(3, embed)
        self.attn = nn.MultiheadAttention(embed, heads, batch_first=True)
        self.norm = nn.LayerNorm(embed)
        self.fuse = nn.Sequential(nn.Linear(2 * embed, features), nn.ReLU())
        self.features_dim = features

    def forward(self, obs):
        B = obs.shape[0]
        fr = obs.view(B, self.n_stack, self.frame)
        q = self.uav(fr[:, :, :3].reshape(B, -1))
        sens = fr[:, -1, 3:].view(B, self.slots, 3)
        mask = (sens.abs().sum(-1) < 1e-6) | (sens[:, :, 2] < 1e-6)
        mask = mask & ~mask.all(1, keepdim=True)
        kv = F.relu(self.sensor(sens))
        ctx, _ = self.attn(q.unsqueeze(1), kv, kv, key_padding_mask=mask)
        return self.fuse(torch.cat([q, self.norm(ctx.squeeze(1))], -1))


class QNet(nn.Module):
    def __init__(self, obs_dim, n_stack, extractor, arch=(512, 512, 256), n_actions=5):
        super().__init__()
        self.features = AttentionFeatures(n_stack) if extractor == "attention" else nn.Flatten()
        d = self.features.features_dim if extractor == "attention" else obs_dim * n_stack
        layers = []
        for h in arch:
            layers += [nn.Linear(d, h), nn.ReLU()]
            d = h
        self.head = nn.Sequential(*layers, nn.Linear(d, n_actions))

    def forward(self, x):
        return self.head(self.features(x))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--sensors", type=int, default=20)
    ap.add_argument("--n-stack", type=int, default=4)
    ap.add_argument("--extractor", choices=["mlp", "attention"], default="mlp")
    ap.add_argument("--vector-steps", type=int, default=400)
    ap.add_argument("--learning-starts", type=int, default=50, help="vector steps before learning")
    ap.add_argument("--train-freq", type=int, default=4, help="vector steps per learner phase (dqn.py:1089)")
    ap.add_argument("--updates", type=int, default=4, help="gradient steps per learner phase")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--ring", type=int, default=256, help="replay capacity in vector steps (x envs transitions)")
    ap.add_argument("--domain-rand", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    torch.manual_seed(args.seed)
    flags = 0
    kw = dict(num_sensors=args.sensors, pad_sensors=50, seed=args.seed)
    if args.domain_rand:
        flags = U.FLAG_RANDOM_LAYOUT | U.FLAG_FAR_START | U.FLAG_PROX_SHAPING | U.FLAG_JAIN_BONUS
        kw.update(grid_choices=U.CURRICULUM_STAGES[4][0], grid_size=(100, 100))
    else:
        kw.update(grid_size=(500, 500))
    env = U.BatchedUAVEnv(args.envs, flags=flags, **kw)
    dev, E, D, k = env.device, args.envs, env.obs_dim, args.n_stack
    fs = U.FrameStack(E, D, k, dev)
    ring = U.TransitionRing(args.ring, E, D, dev)
    ring.attach(env)
    q, q_tgt = QNet(D, k, args.extractor).to(dev), QNet(D, k, args.extractor).to(dev)
    q_tgt.load_state_dict(q.state_dict())
    opt = torch.optim.Adam(q.parameters(), lr=3e-4)
    gamma, eps_final, eps_frac = 0.99, 0.03, 0.25
    target_every = max(1, 5000 // E)            # dqn.py:1088 counts env timesteps

    obs = env.reset()
    ring.local_obs_slot().copy_(obs)
    zero = torch.zeros(E, device=dev)
    ring.commit(zero, zero, zero)
    stacked = fs.reset(obs)
    ep_returns, losses = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(args.vector_steps):
        eps = max(eps_final, 1.0 - (1.0 - eps_final) * step / max(1, eps_frac * args.vector_steps))
        with torch.no_grad():
            greedy = q(stacked).argmax(1).to(torch.int32)
        rnd = torch.randint(0, 5, (E,), device=dev, dtype=torch.int32)
        actions = torch.where(torch.rand(E, device=dev) < eps, rnd, greedy)
        o, r, d = env.step(actions, obs_out=ring.local_obs_slot())
        ring.commit()
        stacked = fs.step(o, d, None)
        if step >= args.learning_starts and step % args.train_freq == 0:
            for _ in range(args.updates):
                b = ring.sample_stacked(args.batch, k)
                with torch.no_grad():
                    tgt = b["reward"] + gamma * q_tgt(b["next_obs"]).max(1).values      # terminated is always False
                qa = q(b["obs"]).gather(1, b["action"].unsqueeze(1)).squeeze(1)
                loss = (F.smooth_l1_loss(qa, tgt, reduction="none") * b["valid"]).sum() / b["valid"].sum().clamp(min=1)
                opt.zero_grad(set_to_none=True)
                loss.backward()
                nn.utils.clip_grad_norm_(q.parameters(), 10.0)
                opt.step()
            losses.append(loss.detach())
        if step % target_every == 0:
            q_tgt.load_state_dict(q.state_dict())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = env.episode_stats()
    done_eps = st[st["valid"] == 1]
    out = {"envs": E, "sensors": args.sensors, "n_stack": k, "extractor": args.extractor, "vector_steps": args.vector_steps,
           "timesteps": E * args.vector_steps, "timesteps_per_s": E * args.vector_steps / dt,
           "ms_per_vector_step": dt / args.vector_steps * 1e3, "gradient_steps": len(losses) * args.updates,
           "last_loss": float(losses[-1]) if losses else None,
           "finished_episodes_seen": int(len(done_eps)),
           "mean_episode_return": float(done_eps["episode_return"].mean()) if len(done_eps) else None}
    print(json.dumps(out))
    env.close()


if __name__ == "__main__":
    main()
