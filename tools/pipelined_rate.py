"""Dev tool (GPU box): double-buffered stepping.  The 4096 environments are split into P independent groups (own handle,
own stream, env_index_base so that the union is bit-identical to one 4096-environment handle); each group's steps form
a dependent chain, the chains run concurrently, so the slow tail of one group's launch overlaps the body of another's.
Prints the aggregate env-steps/s for P = 1, 2, 4, 8 with HIP graphs of 64 steps per chain."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uavenv_amd as U

E, N, L, REPS = 4096, 50, 64, 20
dev = torch.device("cuda", 0)
obs = torch.zeros(L, E, 153, dtype=torch.float32, device=dev)
for P in (1, 2, 4, 8):
    Eg = E // P
    envs = [U.BatchedUAVEnv(Eg, num_sensors=N, seed=0, env_index_base=g * Eg) for g in range(P)]
    streams = [torch.cuda.Stream(dev) for _ in range(P)]
    for e in envs:
        e.reset()
    for _ in range(50):
        for g, e in enumerate(envs):
            with torch.cuda.stream(streams[g]):
                e.step_random()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cap = torch.cuda.current_stream()
        for g, e in enumerate(envs):
            streams[g].wait_stream(cap)
            with torch.cuda.stream(streams[g]):
                for k in range(L):
                    e.step_random(obs_out=obs[k, g * Eg:(g + 1) * Eg])
        for g in range(P):
            cap.wait_stream(streams[g])
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        graph.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"P={P}: {E * L * REPS / dt / 1e6:7.1f} M env-steps/s   {dt / (L * REPS) * 1e6:6.2f} us per vector step of {E} envs")
    for e in envs:
        e.close()
