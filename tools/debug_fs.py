import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, uavenv_amd as U
E, steps, k = 24, 12, 4
env = U.BatchedUAVEnv(E, num_sensors=10, max_steps=11, seed=3)
D = env.obs_dim
ring = U.TransitionRing(steps + 1, E, D, env.device); ring.attach(env)
fs = U.FrameStack(E, D, k, env.device)
obs = env.reset(); ring.local_obs_slot().copy_(obs)
z = torch.zeros(E, device=env.device); ring.commit(z, z, z)
st = fs.reset(obs).clone()
print('reset stack frame norms', st[0].view(k, D).abs().sum(1).tolist())
for s in range(steps):
    o, r, d = env.step_random(obs_out=ring.local_obs_slot())
    print(s, 'o norm', float(o[0].abs().sum()), 'done', int(d[0]), 'ptr', o.data_ptr(), o.is_contiguous(), o.shape)
    ring.commit(env.actions_taken, env.reward32, d)
    st = fs.step(o, d, env.terminal_obs).clone()
    print('   stack frame norms', [round(x, 3) for x in st[0].view(k, D).abs().sum(1).tolist()], 'ring slot norm', float(ring.obs[s + 1, 0, 0].abs().sum()))
