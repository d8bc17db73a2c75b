"""GPU box: does the packaged DQN learner learn?  Small grid, short episodes; prints random vs greedy evaluation returns."""
import json, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uavenv_amd as U
from uavenv_amd.learner import DQNLearner

cfgs = [
    dict(E=256, env=dict(num_sensors=5, grid_size=(20, 20), max_steps=80, seed=1),
         lrn=dict(learning_rate=1e-3, buffer_size=100_000, learning_starts=2_000, target_update_interval=2_000, train_freq=1,
                  gradient_steps=2, net_arch=(128, 128), n_stack=2, total_timesteps=150_000, exploration_fraction=0.5, reward_scale=1e-3)),
    dict(E=256, env=dict(num_sensors=5, grid_size=(20, 20), max_steps=80, seed=1),
         lrn=dict(learning_rate=1e-3, buffer_size=100_000, learning_starts=2_000, target_update_interval=2_000, train_freq=1,
                  gradient_steps=4, net_arch=(128, 128), n_stack=2, total_timesteps=300_000, exploration_fraction=0.5, reward_scale=1e-4)),
    dict(E=512, env=dict(num_sensors=10, grid_size=(30, 30), max_steps=120, seed=2),
         lrn=dict(learning_rate=5e-4, buffer_size=200_000, learning_starts=5_000, target_update_interval=5_000, train_freq=1,
                  gradient_steps=4, net_arch=(256, 256), n_stack=2, total_timesteps=600_000, exploration_fraction=0.4, reward_scale=1e-3)),
]
for c in cfgs:
    for seed in (0, 1):
        env = U.BatchedUAVEnv(c["E"], **c["env"])
        ev = U.BatchedUAVEnv(256, **dict(c["env"], seed=99))
        L = DQNLearner(env, seed=seed, **c["lrn"])
        r0, n0 = L.evaluate(ev, 1, "random")
        g0, _ = L.evaluate(ev, 1, "greedy")
        t0 = time.perf_counter()
        L.learn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        g1, n1 = L.evaluate(ev, 1, "greedy")
        print(json.dumps(dict(cfg=c["env"], seed=seed, steps=c["lrn"]["total_timesteps"], random=r0, greedy_before=g0, greedy_after=g1,
                              episodes=n1, seconds=dt, updates=L.n_updates, loss=float(L.last_loss.detach()))))
        env.close(); ev.close()
