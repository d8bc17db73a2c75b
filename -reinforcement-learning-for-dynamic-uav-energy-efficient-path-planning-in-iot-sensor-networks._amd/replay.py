"""Shared replay ring fed by an all-gather of transition blocks (BASELINE config 4).

One process per GPU owns a shard of environments.  Each vector step produces, per rank, ONE contiguous
transition block   [ obs: E x D float32 | aux: E x 4 float32 = (action, reward, done, terminal row) ].
The step kernel writes both parts DIRECTLY into this rank's block of the current ring slot
(`local_obs_slot()` / `local_aux_slot()`, uavenv_set_aux_output), so inserting into the replay buffer
costs no copy and no pack kernel; the other ranks' blocks arrive through one in-place
`all_gather_into_tensor` per step (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for the
multi-process tests), issued on a side stream so that it overlaps the next environment step.

next_obs is not stored: it is the following slot's obs, except where `done` (the env was auto-reset,
so the following slot holds the first observation of the NEXT episode): there the terminal observation
is looked up in a compact pool that the step kernel fills directly (uavenv_set_terminal_pool; SB3's
"terminal_observation" semantics, `terminated` is always False in this environment so every episode end
bootstraps from its terminal observation).  The pool is local to a rank: transitions of other ranks that
end an episode are returned with valid=False (about 1 in 1500) and must be masked out of the loss.
"""
import torch
import torch.distributed as dist


class TransitionRing:
    def __init__(self, capacity, envs_per_rank, obs_dim, device, world_size=1, rank=0, group=None, terminal_rows=None):
        self.capacity, self.E, self.D = int(capacity), int(envs_per_rank), int(obs_dim)
        self.world, self.rank, self.group = int(world_size), int(rank), group
        self.device = torch.device(device)
        # [slot][rank][block]: a rank's block (obs then aux) is contiguous => ONE in-place all-gather per step
        self.block = self.E * (self.D + 4)
        self.store = torch.zeros(self.capacity, self.world, self.block, dtype=torch.float32, device=self.device)
        self.obs = self.store[:, :, :self.E * self.D].view(self.capacity, self.world, self.E, self.D)
        self.aux = self.store[:, :, self.E * self.D:].view(self.capacity, self.world, self.E, 4)
        # terminal-observation pool (rows recycle; sized so that a row outlives the ring slot that refers to it:
        # ~E/1400 episodes end per step)
        self.terminal_rows = int(terminal_rows) if terminal_rows is not None else max(1024, (self.capacity * self.E) // 256)
        self.term_pool = torch.zeros(self.terminal_rows, self.D, dtype=torch.float32, device=self.device)
        self.term_counter = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._env = None
        self.head = 0                 # next slot to write
        self.size = 0                 # number of valid slots
        self._pending = [None] * self.capacity
        self._comm_stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def attach(self, env):
        """Let `env` (BatchedUAVEnv) write terminal observations into this ring's pool and the (action, reward,
        done, terminal row) part of every transition straight into the ring."""
        env.set_terminal_pool(self.term_pool, self.term_counter, None)
        self._env = env
        env.set_aux_output(self.local_aux_slot())

    # ---- producer side -----------------------------------------------------------------------
    def local_obs_slot(self, slot=None):
        """[E, D] view of THIS rank's part of a slot: pass it as `obs_out` to BatchedUAVEnv.step*()."""
        return self.obs[self.head if slot is None else slot, self.rank]

    def local_aux_slot(self, slot=None):
        return self.aux[self.head if slot is None else slot, self.rank]

    def wait_slot(self, slot):
        w = self._pending[slot]
        if w is not None:
            w.wait()
            self._pending[slot] = None

    def commit(self, actions=None, reward=None, done=None):
        """Publish this rank's block of the current slot.  With an attached env the kernel has already written
        the aux part; otherwise (tests, foreign producers) pass actions / reward / done to fill it here."""
        slot = self.head
        if actions is not None:
            aux = self.aux[slot, self.rank]
            aux[:, 0] = actions.to(torch.float32)
            aux[:, 1] = reward.to(torch.float32)
            aux[:, 2] = done.to(torch.float32)
            aux[:, 3] = -1.0
        if self.world > 1:
            out, inp = self.store[slot].view(-1), self.store[slot, self.rank]
            if self._comm_stream is not None:
                self._comm_stream.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self._comm_stream):
                    w = dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)
            else:   # CPU / gloo (tests)
                w = dist.all_gather_into_tensor(out, inp.clone(), group=self.group, async_op=True)
            self._pending[slot] = w
        self.head = (slot + 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)
        # the slot about to be overwritten next must have finished its previous gather
        self.wait_slot(self.head)
        if self._env is not None:
            self._env.set_aux_output(self.local_aux_slot())
        return slot

    # ---- launch-bound producer loops: one HIP graph per ring revolution ---------------------------------
    def capture_revolution(self, step_fn):
        """Capture `capacity` consecutive `step_fn(obs_slot); commit()` pairs -- one full revolution of the ring,
        starting at the current head -- into ONE HIP graph and return it.  `replay_revolution(graph)` then costs a
        single graph launch instead of `capacity` Python -> ctypes -> hipLaunchKernel round trips (about 12 us each,
        more than the step kernel itself at 4096 environments).  Single-rank rings only: the collective of a shared
        ring is issued from the host.  `step_fn(obs_out)` must only enqueue work (e.g. `env.step_random`)."""
        assert self.world == 1 and self.device.type == "cuda" and self._env is not None
        head0, size0 = self.head, self.size
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(self.capacity):
                step_fn(self.local_obs_slot())
                self.commit()
        # capturing executed nothing and the revolution ends where it began
        assert self.head == head0
        self.size = size0
        return g

    def replay_revolution(self, graph):
        graph.replay()
        self.size = self.capacity

    def drain(self):
        for s in range(self.capacity):
            self.wait_slot(s)
        if self._comm_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._comm_stream)

    # ---- consumer side -----------------------------------------------------------------------
    def _draw(self, batch_size, generator):
        n_slots = self.size - 1
        oldest = (self.head - self.size) % self.capacity
        j = torch.randint(0, n_slots, (batch_size,), generator=generator, device=self.device)
        r = torch.randint(0, self.world, (batch_size,), generator=generator, device=self.device)
        e = torch.randint(0, self.E, (batch_size,), generator=generator, device=self.device)
        return j, (oldest + j) % self.capacity, r, e

    def _next_frame(self, slot, r, e):
        """(action, reward, done, newest frame of next_obs, valid) of the transitions slot -> slot+1."""
        nxt = (slot + 1) % self.capacity
        aux = self.aux[nxt, r, e]                  # the aux row stored WITH an observation describes the step INTO it
        done = aux[:, 2] > 0.5
        tidx = aux[:, 3].long()
        have_term = done & (tidx >= 0) & (r == self.rank)
        last = torch.where(have_term.unsqueeze(1), self.term_pool[tidx.clamp(min=0)], self.obs[nxt, r, e])
        return aux[:, 0].long(), aux[:, 1], done, last, ~done | have_term

    def sample(self, batch_size, generator=None):
        """Uniform sample of transitions (obs, action, reward, done, next_obs, valid) over all ranks' envs.
        Slot s holds the observation s_t together with (a, r, done) of the step that PRODUCED it, so the
        transition out of slot s reads its action / reward / done from slot s+1."""
        assert self.size >= 2
        self.drain()
        _, slot, r, e = self._draw(batch_size, generator)
        action, reward, done, last, valid = self._next_frame(slot, r, e)
        return dict(obs=self.obs[slot, r, e], action=action, reward=reward, done=done, next_obs=last, valid=valid)

    def sample_stacked(self, batch_size, n_stack, generator=None):
        """Like sample(), but observations are frame stacks of `n_stack` frames gathered from the ring on the
        fly (SB3 VecFrameStack layout: [oldest | ... | newest], frames from before the episode start zeroed),
        so the replay stores each frame ONCE instead of n_stack times (dqn.py:1085 budgets 2 x 612 floats per
        transition for the stacked copies)."""
        assert self.size >= n_stack + 1
        self.drain()
        k = int(n_stack)
        j, slot, r, e = self._draw(batch_size, generator)
        back = torch.arange(k - 1, -1, -1, device=self.device)                      # k-1 ... 0 (oldest first)
        fs = (slot.unsqueeze(1) - back.unsqueeze(0)) % self.capacity                # [B, k] frame slots
        in_ring = (j.unsqueeze(1) - back.unsqueeze(0)) >= 0                         # frame older than the ring start?
        rr, ee = r.unsqueeze(1).expand(-1, k), e.unsqueeze(1).expand(-1, k)
        frames = self.obs[fs, rr, ee]                                               # [B, k, D]
        dn = self.aux[fs, rr, ee, 2] > 0.5                                          # frame is the first of an episode
        # frame i (i < k-1) is valid iff no episode start among frames i+1 .. k-1
        later_start = torch.flip(torch.cumsum(torch.flip(dn[:, 1:], [1]).int(), 1), [1]) > 0     # [B, k-1]
        valid_f = torch.cat([~later_start, torch.ones(batch_size, 1, dtype=torch.bool, device=self.device)], 1) & in_ring
        frames = frames * valid_f.unsqueeze(2)
        action, reward, done, last, valid = self._next_frame(slot, r, e)
        # next stack = [frames 1..k-1 | newest]; after an auto-reset the real next state is the terminal observation
        next_obs = torch.cat([frames[:, 1:].reshape(batch_size, (k - 1) * self.D), last], 1)
        return dict(obs=frames.reshape(batch_size, k * self.D), action=action, reward=reward, done=done,
                    next_obs=next_obs, valid=valid)
