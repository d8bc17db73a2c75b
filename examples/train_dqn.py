#!/usr/bin/env python3
"""BASELINE config 3 in one process: 4096 environments x 50 sensor slots driving the packaged DQN learner
(uavenv_amd.learner.DQNLearner: the reference's hyper-parameters, agents/dqn/dqn.py:1077-1099, with SB3's semantics),
everything resident on one MI355X:

    HIP step kernel -> device frame stack (uavenv_frame_stack) -> epsilon-greedy Q-network (torch) -> actions
                    -> transition ring (observations + terminal rows written in place) -> stacked sampling -> learner

`--extractor attention` uses the architecture of the reference's UAVAttentionExtractor (dqn.py:548-650, N_STACK = 10).
With stable-baselines3 installed the same environments are available to SB3 itself through `uavenv_amd.UAVVecEnv`.
Prints one JSON line: timesteps/s, finished episodes and their mean return, last loss.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import uavenv_amd as U  # noqa: E402
from uavenv_amd.learner import DQNLearner, REFERENCE_HYPERPARAMS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--sensors", type=int, default=50)
    ap.add_argument("--n-stack", type=int, default=4)
    ap.add_argument("--extractor", choices=["mlp", "attention"], default="mlp")
    ap.add_argument("--timesteps", type=int, default=3_000_000, help="total env transitions (TRAINING_CONFIG: 3 M)")
    ap.add_argument("--gradient-steps", type=int, default=1, help="updates per rollout of train_freq vector steps (SB3 default 1)")
    ap.add_argument("--reward-scale", type=float, default=1.0)
    ap.add_argument("--domain-rand", action="store_true")
    ap.add_argument("--tune-gemms", action="store_true", help="let PyTorch's TunableOp pick the update's GEMM kernels even without a cache (seconds of tuning, persisted under $UAVENV_CACHE_DIR; by default a cache is used when present)")
    ap.add_argument("--updates-per-transition", type=float, default=None, help="e.g. 0.0625 = the reference's one update per 16 transitions (overrides --gradient-steps)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--save", default=None, help="write the learner's checkpoint here at the end (networks, Adam state, counters: the reference's model.save)")
    ap.add_argument("--load", default=None, help="resume from a checkpoint written by --save (--timesteps is then the TOTAL to reach)")
    args = ap.parse_args()

    flags = 0
    kw = dict(num_sensors=args.sensors, pad_sensors=50, seed=args.seed)
    if args.domain_rand:
        flags = U.FLAG_RANDOM_LAYOUT | U.FLAG_FAR_START | U.FLAG_PROX_SHAPING | U.FLAG_JAIN_BONUS
        kw.update(grid_choices=U.CURRICULUM_STAGES[4][0], grid_size=(100, 100))
    else:
        kw.update(grid_size=(500, 500))
    env = U.BatchedUAVEnv(args.envs, flags=flags, **kw)
    hp = dict(REFERENCE_HYPERPARAMS, n_stack=args.n_stack, total_timesteps=args.timesteps, gradient_steps=args.gradient_steps)
    learner = DQNLearner(env, extractor=args.extractor, seed=args.seed, reward_scale=args.reward_scale, updates_per_transition=args.updates_per_transition, tune_gemms=True if args.tune_gemms else None, **hp)
    if args.load:
        learner.load(args.load)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    half = {}

    def mark(l):            # the second half of the run is free of one-time costs (library warm-up, graph captures)
        if "t" not in half and l.num_timesteps >= args.timesteps // 2:
            torch.cuda.synchronize()
            half.update(t=time.perf_counter(), n=l.num_timesteps, calls=l.n_calls)
    learner.learn(callback=mark)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if args.save:
        learner.save(args.save)
    dt = t1 - t0
    st = env.episode_stats()
    done_eps = st[st["valid"] == 1]
    out = {"envs": args.envs, "sensors": args.sensors, "n_stack": args.n_stack, "extractor": args.extractor,
           "timesteps": learner.num_timesteps, "vector_steps": learner.n_calls, "timesteps_per_s": learner.num_timesteps / dt,
           "ms_per_vector_step": dt / max(1, learner.n_calls) * 1e3, "gradient_steps": learner.n_updates,
           "second_half_timesteps_per_s": (learner.num_timesteps - half["n"]) / (t1 - half["t"]) if "t" in half and t1 > half["t"] else None,
           "second_half_ms_per_vector_step": (t1 - half["t"]) / max(1, learner.n_calls - half["calls"]) * 1e3 if "t" in half else None,
           "graph_replay": learner._act_graphs is not None and learner._train_graph is not None, "tune_gemms": bool(learner.tune_gemms),
           "updates_per_transition": learner.updates_per_transition,
           "last_loss": None if learner.last_loss is None else float(learner.last_loss.detach()),
           "learning_rate_now": learner.lr_schedule(learner.progress_remaining()), "epsilon_now": learner.exploration_rate(),
           "replay_slots": learner.ring.capacity, "replay_chunk": learner.ring.L,
           "finished_episodes_seen": int(len(done_eps)),
           "mean_episode_return": float(done_eps["episode_return"].mean()) if len(done_eps) else None,
           "mean_episode_length": float(done_eps["length"].mean()) if len(done_eps) else None}
    print(json.dumps(out))
    env.close()


if __name__ == "__main__":
    main()
