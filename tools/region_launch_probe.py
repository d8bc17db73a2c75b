"""Dev tool (GPU box): what is the cheapest way to launch a SHORT region of K single-step launches onto an idle GPU between two
device synchronisations (the shape of bench.py's timed region at the driver's --steps 20)?  One graph of K nodes, or the first
m launches eagerly (the GPU starts after one packet) followed by a graph of the other K - m."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, uavenv_amd as U
E, K = 4096, 20
env = U.BatchedUAVEnv(E, num_sensors=50, seed=0)
ring = U.TransitionRing(2 * K, E, env.obs_dim, env.device, chunk_len=K); ring.attach(env)
env.reset()
for _ in range(100): env.step_random()
torch.cuda.synchronize()
def launch(k):
    ring._point_env(k)
    env.step_random(obs_out=ring.local_obs_slot(k))
for m in (0, 1, 2, 4, K):
    g = None
    if m < K:
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(m, K): launch(k)
    def region():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(m): launch(k)
        if g is not None: g.replay()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    for _ in range(5): region()
    ts = sorted(region() for _ in range(41))
    print("first %2d eager + graph of %2d: median %.2f us/step (min %.2f)" % (m, K - m, ts[20] / K * 1e6, ts[0] / K * 1e6), flush=True)
