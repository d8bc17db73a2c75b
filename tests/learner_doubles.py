"""Test doubles for running `DQNLearner` WITHOUT the HIP extension (CPU, torch.distributed over gloo): a deterministic
stand-in environment that fills the transition ring the way the step kernel does (observation rows written in place, aux
rows = (action, reward, done, terminal ticket), terminal rows claimed from the chunk's counter) and a torch frame stack with
SB3 `VecFrameStack` semantics.  They exist so that the learner's HOST logic -- schedules, the ring exchange across ranks,
the batch split and the gradient averaging -- can be exercised where there is no GPU; the product classes they stand in for
(BatchedUAVEnv, FrameStack) have no CPU path."""
import torch


class ToyEnv:
    """E environments per rank; observation row of env e at step t: obs[0] = 1000 * rank + e + t / 1000 (identifies the
    transition), the rest a fixed pseudo-random function of (rank, e, t).  Environment e ends an episode whenever
    (t + e + rank) % period == 0; reward = 0.01 * action + obs[1]."""

    def __init__(self, num_envs, obs_dim, rank=0, period=7, device="cpu"):
        self.num_envs, self.obs_dim, self.rank, self.period = int(num_envs), int(obs_dim), int(rank), int(period)
        self.device = torch.device(device)
        self.launch_epoch = 0
        self.t = 0
        self.reward32 = torch.zeros(self.num_envs)
        self.done = torch.zeros(self.num_envs, dtype=torch.uint8)
        self._pool = self._aux = None

    def _obs(self, t, terminal=False):
        e = torch.arange(self.num_envs, dtype=torch.float32)
        g = torch.Generator().manual_seed(1000003 * self.rank + 7919 * t + (1 if terminal else 0))
        o = torch.rand(self.num_envs, self.obs_dim, generator=g)
        o[:, 0] = 1000.0 * self.rank + e + t / 1000.0 + (0.5 if terminal else 0.0)
        return o

    def set_terminal_pool(self, pool, counter, index_out):
        self._pool = (pool, counter)

    def set_aux_output(self, aux):
        self._aux = aux

    def reset(self):
        self.t = 0
        return self._obs(0)

    def step(self, actions, obs_out=None):
        self.t += 1
        t = self.t
        e = torch.arange(self.num_envs)
        done = ((t + e + self.rank) % self.period == 0)
        obs = self._obs(t)
        reward = 0.01 * actions.to(torch.float32) + obs[:, 1]
        if obs_out is not None:
            obs_out.copy_(obs)
        aux = self._aux
        aux[:, 0] = actions.to(torch.float32); aux[:, 1] = reward; aux[:, 2] = done.to(torch.float32)
        tick = aux.view(torch.int32)[:, 3]
        tick.fill_(-1)
        idx = torch.nonzero(done).flatten()
        if idx.numel():
            pool, counter = self._pool
            base = int(counter[0])
            tk = base + torch.arange(idx.numel())
            tick[idx] = tk.to(torch.int32)
            pool[tk % pool.shape[0]] = self._obs(t, terminal=True)[idx]
            counter[0] = base + idx.numel()
        self.reward32, self.done = reward, done.to(torch.uint8)
        return (obs_out if obs_out is not None else obs), reward.double(), self.done


class TorchFrameStack:
    """SB3 StackedObservations on torch tensors (what csrc's uavenv_frame_stack does on the device)."""

    def __init__(self, num_envs, obs_dim, n_stack, device):
        self.E, self.D, self.k = num_envs, obs_dim, n_stack
        self.stacked = torch.zeros(num_envs, n_stack * obs_dim)
        self.terminal_stacked = torch.zeros_like(self.stacked)

    def reset(self, obs):
        self.stacked.zero_()
        self.stacked[:, -self.D:] = obs
        return self.stacked

    def step(self, obs, done=None, terminal_obs=None):
        self.stacked = torch.roll(self.stacked, -self.D, dims=1)
        if done is not None:
            self.stacked[done.bool()] = 0.0
        self.stacked[:, -self.D:] = obs
        return self.stacked
