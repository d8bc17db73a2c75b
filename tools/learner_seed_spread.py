"""GPU box: how much does the outcome of the short learning run of tests/test_gpu_learner.py depend on the seed, with the
eager loops and with the graph replays?  Greedy / random return ratio for seeds 0-3 x (eager, graphs) per configuration.
Found with it: a constant learning rate of 1e-3 ends anywhere between -0.9 x and 1.4 x (the final policy is whatever the last
updates left), a rate that decays linearly to 0 ends at 1.21-1.47 x in all eight runs, with clipping at 1.0 at 1.26-1.53 x."""
import sys, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import uavenv_amd as U
from uavenv_amd import learner as LR
kw = dict(num_sensors=5, grid_size=(20, 20), max_steps=80)
cfgs = dict(
    B_decay=dict(learning_rate=lambda p: 1e-3 * p, total_timesteps=300_000),
    A_lr3e4=dict(learning_rate=3e-4, total_timesteps=400_000),
    C_clip1=dict(learning_rate=lambda p: 1e-3 * p, total_timesteps=300_000, max_grad_norm=1.0),
)
for name, c in cfgs.items():
    res = []
    for seed in (0, 1, 2, 3):
        for mode in (False, True):
            env = U.BatchedUAVEnv(256, seed=1, **kw); held = U.BatchedUAVEnv(256, seed=99, **kw)
            hp = dict(buffer_size=100_000, learning_starts=2_000, target_update_interval=2_000, train_freq=1, gradient_steps=4,
                      net_arch=(128, 128), n_stack=2, exploration_fraction=0.5, reward_scale=1e-4, seed=seed, use_graphs=mode)
            hp.update(c)
            L = LR.DQNLearner(env, **hp)
            r0, _ = L.evaluate(held, 1, "random")
            L.learn()
            g, _ = L.evaluate(held, 1, "greedy")
            res.append(round(g / r0, 2))
            env.close(); held.close()
    print(name, res, flush=True)
