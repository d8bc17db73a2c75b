"""GPU box: where a vector step of the packaged DQN learner goes (config 3 shape: 4096 x 50, reference hyper-parameters).
Each part is timed with a device synchronize on both sides over `reps` iterations (so launch latency is included and nothing
overlaps): the sum is an upper bound of the pipelined loop, which is timed as well."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uavenv_amd as U
from uavenv_amd.learner import DQNLearner, REFERENCE_HYPERPARAMS

E = int(os.environ.get("ENVS", 4096))
ext = os.environ.get("EXTRACTOR", "mlp")
k = 4 if ext == "mlp" else 10
env = U.BatchedUAVEnv(E, num_sensors=50, pad_sensors=50, grid_size=(500, 500), seed=0)
hp = dict(REFERENCE_HYPERPARAMS, n_stack=k, total_timesteps=10**9)
L = DQNLearner(env, extractor=ext, seed=0, use_graphs=False, **hp)
L.collect(64)
torch.cuda.synchronize()


def timed(f, reps=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


out = {"envs": E, "extractor": ext, "n_stack": k}
st = L._stacked
out["act_us"] = timed(lambda: L.act(st, 0.5))
acts = L.act(st, 0.5)
with torch.no_grad():
    out["q_forward_us"] = timed(lambda: L.q(st))
out["env_step_us"] = timed(lambda: L.env.step(acts, obs_out=L.ring.local_obs_slot()))
o, _, d = L.env.step(acts, obs_out=L.ring.local_obs_slot())
out["frame_stack_us"] = timed(lambda: L.fs.step(o, d, None))


def one():
    L.collect(1)
out["collect_1_us"] = timed(one, 64)
out["train_1_us"] = timed(lambda: L.train(1), 30)
out["sample_us"] = timed(lambda: L.ring.sample_stacked(L.batch_size, L.k, generator=L.gen), 30)
t0 = time.perf_counter(); n0 = L.n_calls
for _ in range(100):
    L.collect(L.train_freq); L.train()
torch.cuda.synchronize()
out["loop_us_per_vector_step"] = (time.perf_counter() - t0) / (L.n_calls - n0) * 1e6
env2 = U.BatchedUAVEnv(E, num_sensors=50, pad_sensors=50, grid_size=(500, 500), seed=0)
G = DQNLearner(env2, extractor=ext, seed=0, use_graphs=True, tune_gemms=os.environ.get("TUNE_GEMMS") == "1", **dict(hp, learning_starts=0))
out["tune_gemms"] = os.environ.get("TUNE_GEMMS") == "1"
for _ in range(12):
    G.collect(G.train_freq); G.train()
torch.cuda.synchronize()
assert G._act_graphs is not None and G._train_graph is not None
t0 = time.perf_counter(); n0 = G.n_calls
for _ in range(200):
    G.collect(G.train_freq); G.train()
torch.cuda.synchronize()
out["graph_loop_us_per_vector_step"] = (time.perf_counter() - t0) / (G.n_calls - n0) * 1e6
out["graph_collect_1_us"] = timed(lambda: G.collect(1), 64)
out["graph_train_1_us"] = timed(lambda: G.train(1), 30)
out["graph_timesteps_per_s"] = E / (out["graph_loop_us_per_vector_step"] * 1e-6)
print(json.dumps(out))
