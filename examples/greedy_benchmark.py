#!/usr/bin/env python3
"""The curriculum gate's greedy benchmark (agents/dqn/dqn.py:456-543 `_run_greedy_benchmark`) on device:
MaxThroughputGreedyV2 (or NearestSensorGreedy) runs full episodes on thousands of environments at once -- the
policy is evaluated inside the step kernel and K steps are fused per launch -- and the mean NDR (sensors
visited) and Jain's index over the first finished episode of every environment are reported.  The
reference runs 20 sequential Python episodes per stage for the same numbers."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import uavenv_amd as U  # noqa: E402
from uavenv_amd import _native as N  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--sensors", type=int, default=20)          # GREEDY_BENCHMARK["sensor_count"], dqn.py:103
    ap.add_argument("--stage", type=int, default=4)
    ap.add_argument("--policy", choices=["max_throughput_v2", "nearest"], default="max_throughput_v2")
    ap.add_argument("--fuse", type=int, default=50)
    args = ap.parse_args()
    grid = U.CURRICULUM_STAGES[args.stage][0][-1]               # hardest grid of the stage (dqn.py:476)
    policy = {"max_throughput_v2": N.POLICY_MAX_THROUGHPUT_V2, "nearest": N.POLICY_NEAREST}[args.policy]
    env = U.BatchedUAVEnv(args.envs, num_sensors=args.sensors, grid_size=grid, flags=U.FLAG_FAR_START, seed=0)
    env.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 0
    slab = torch.empty(args.fuse, args.envs, env.obs_dim, dtype=torch.float32, device=env.device)
    while True:
        env.rollout(args.fuse, policy=policy, obs_out=slab)
        steps += args.fuse
        if steps >= 2100 + args.fuse:                           # every env has finished at least one episode
            break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = env.episode_stats()
    assert (st["valid"] == 1).all()
    out = {"policy": args.policy, "grid": list(grid), "sensors": args.sensors, "episodes": int(args.envs),
           "ndr": float((st["sensors_visited"] / st["num_sensors"] * 100).mean()),
           "jains": float(st["jains_index"].mean()),
           "mean_episode_length": float(st["length"].mean()), "mean_collected_bytes": float(st["total_collected"].mean()),
           "env_steps": args.envs * steps, "env_steps_per_s": args.envs * steps / dt, "seconds": dt}
    print(json.dumps(out))
    env.close()


if __name__ == "__main__":
    main()
