"""Shared replay ring fed by an all-gather of transition batches (BASELINE config 4).

One process per GPU owns a shard of environments.  Each vector step produces, per rank, a
transition block  obs[E_local, D] float32 + aux[E_local, 4] float32 (action, reward, done, pad).
The step kernel writes its observations DIRECTLY into this rank's slice of the ring slot
(`slot_obs(slot)[rank]`), so inserting into the replay buffer costs no copy; the other ranks'
slices arrive through one in-place `all_gather_into_tensor` per tensor (RCCL over xGMI when the
backend is "nccl"; "gloo" on CPU for the multi-process tests).  The collective is issued on a side
stream so that it overlaps the next environment step.

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
        # [slot][rank][env][...]: a rank's block is contiguous => in-place all-gather
        self.obs = torch.zeros(self.capacity, self.world, self.E, self.D, dtype=torch.float32, device=self.device)
        self.aux = torch.zeros(self.capacity, self.world, self.E, 4, dtype=torch.float32, device=self.device)
        # terminal-observation pool (rows recycle; sized so that a row outlives the ring slot that refers to it:
        # ~E/1400 episodes end per step)
        self.terminal_rows = int(terminal_rows) if terminal_rows is not None else max(1024, (self.capacity * self.E) // 256)
        self.term_pool = torch.zeros(self.terminal_rows, self.D, dtype=torch.float32, device=self.device)
        self.term_counter = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.term_index = torch.full((self.E,), -1, dtype=torch.int32, device=self.device)
        self._attached = False
        self.head = 0                 # next slot to write
        self.size = 0                 # number of valid slots
        self._pending = [None] * self.capacity
        self._comm_stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def attach(self, env):
        """Let `env` (BatchedUAVEnv) write terminal observations straight into this ring's pool."""
        env.set_terminal_pool(self.term_pool, self.term_counter, self.term_index)
        self._attached = True

    # ---- producer side -----------------------------------------------------------------------
    def local_obs_slot(self, slot=None):
        """[E, D] view of THIS rank's slice of a slot: pass it as `obs_out` to BatchedUAVEnv.step*()."""
        return self.obs[self.head if slot is None else slot, self.rank]

    def wait_slot(self, slot):
        w = self._pending[slot]
        if w is not None:
            for x in w:
                x.wait()
            self._pending[slot] = None

    def commit(self, actions, reward, done):
        """Record (action, reward, done) of the step whose observations were written into
        local_obs_slot(), then publish this rank's block of the slot to every rank."""
        slot = self.head
        aux = self.aux[slot, self.rank]
        aux[:, 0] = actions.to(torch.float32)
        aux[:, 1] = reward.to(torch.float32)
        aux[:, 2] = done.to(torch.float32)
        aux[:, 3] = self.term_index.to(torch.float32) if self._attached else -1.0
        if self.world > 1:
            if self._comm_stream is not None:
                self._comm_stream.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self._comm_stream):
                    w1 = dist.all_gather_into_tensor(self.obs[slot].view(-1), self.obs[slot, self.rank].reshape(-1),
                                                     group=self.group, async_op=True)
                    w2 = dist.all_gather_into_tensor(self.aux[slot].view(-1), self.aux[slot, self.rank].reshape(-1),
                                                     group=self.group, async_op=True)
            else:   # CPU / gloo (tests): same calls, synchronous streams
                w1 = dist.all_gather_into_tensor(self.obs[slot].view(-1), self.obs[slot, self.rank].reshape(-1).clone(),
                                                 group=self.group, async_op=True)
                w2 = dist.all_gather_into_tensor(self.aux[slot].view(-1), self.aux[slot, self.rank].reshape(-1).clone(),
                                                 group=self.group, async_op=True)
            self._pending[slot] = (w1, w2)
        self.head = (slot + 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)
        # the slot about to be overwritten next must have finished its previous gather
        self.wait_slot(self.head)
        return slot

    def drain(self):
        for s in range(self.capacity):
            self.wait_slot(s)
        if self._comm_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._comm_stream)

    # ---- consumer side -----------------------------------------------------------------------
    def sample(self, batch_size, generator=None):
        """Uniform sample of transitions (obs, action, reward, done, next_obs) over all ranks' envs.
        Only slots whose successor slot is valid are eligible (next_obs = successor's obs)."""
        assert self.size >= 2
        self.drain()
        n_slots = self.size - 1
        newest = (self.head - 1) % self.capacity
        oldest = (self.head - self.size) % self.capacity
        k = torch.randint(0, n_slots, (batch_size,), generator=generator, device=self.device)
        slot = (oldest + k) % self.capacity
        nxt = (slot + 1) % self.capacity
        r = torch.randint(0, self.world, (batch_size,), generator=generator, device=self.device)
        e = torch.randint(0, self.E, (batch_size,), generator=generator, device=self.device)
        aux = self.aux[slot, r, e]
        assert newest != oldest or self.size == 1
        done = aux[:, 2] > 0.5
        next_obs = self.obs[nxt, r, e]
        tidx = aux[:, 3].long()
        have_term = done & (tidx >= 0) & (r == self.rank)
        next_obs = torch.where(have_term.unsqueeze(1), self.term_pool[tidx.clamp(min=0)], next_obs)
        valid = ~done | have_term
        return dict(obs=self.obs[slot, r, e], action=aux[:, 0].long(), reward=aux[:, 1], done=done,
                    next_obs=next_obs, valid=valid)


    def sample_stacked(self, batch_size, n_stack, generator=None):
        """Like sample(), but observations are frame stacks of `n_stack` frames gathered from the ring on the
        fly (SB3 VecFrameStack layout: [oldest | ... | newest], frames from before the episode start zeroed),
        so the replay stores each frame ONCE instead of n_stack times (dqn.py:1085 budgets 2 x 612 floats per
        transition for the stacked copies)."""
        assert self.size >= n_stack + 1
        self.drain()
        k = int(n_stack)
        n_slots = self.size - 1
        oldest = (self.head - self.size) % self.capacity
        j = torch.randint(0, n_slots, (batch_size,), generator=generator, device=self.device)
        slot = (oldest + j) % self.capacity
        r = torch.randint(0, self.world, (batch_size,), generator=generator, device=self.device)
        e = torch.randint(0, self.E, (batch_size,), generator=generator, device=self.device)
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
        obs = frames.reshape(batch_size, k * self.D)
        nxt = (slot + 1) % self.capacity
        aux = self.aux[nxt, r, e]                  # (action, reward, done) OF the transition slot -> nxt
        done = aux[:, 2] > 0.5
        tidx = aux[:, 3].long()
        have_term = done & (tidx >= 0) & (r == self.rank)
        last = torch.where(have_term.unsqueeze(1), self.term_pool[tidx.clamp(min=0)], self.obs[nxt, r, e])
        # next stack = [frames 1..k-1 | newest]; after an auto-reset the real next state is the terminal observation
        next_obs = torch.cat([frames[:, 1:].reshape(batch_size, (k - 1) * self.D), last], 1)
        return dict(obs=obs, action=aux[:, 0].long(), reward=aux[:, 1], done=done, next_obs=next_obs,
                    valid=~done | have_term)
