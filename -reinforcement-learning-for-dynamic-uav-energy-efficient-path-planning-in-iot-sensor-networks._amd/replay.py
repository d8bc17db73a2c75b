"""Shared replay ring fed by an all-gather of transition blocks (BASELINE config 4).

One process per GPU owns a shard of environments.  Each vector step produces, per rank, ONE contiguous
transition block   [ obs: E x D float32 | aux: E x 4 = (action, reward, done, terminal ticket) ].
The step kernel writes both parts DIRECTLY into this rank's block of the current ring slot
(`local_obs_slot()` / `local_aux_slot()`, uavenv_set_aux_output), so inserting into the replay buffer
costs no copy and no pack kernel; the other ranks' blocks arrive through one in-place
`all_gather_into_tensor` per chunk of L slots (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for
the multi-process tests), issued on a side stream so that it overlaps the following environment steps.

next_obs is not stored: it is the following slot's obs, except where `done` (the env was auto-reset, so the
following slot holds the first observation of the NEXT episode).  There the step kernel has written the
TERMINAL observation (SB3's "terminal_observation"; `terminated` is always False in this environment, so every
episode end bootstraps from it) into the TERMINAL SECTION of the same chunk: every (chunk, rank) part of the
ring is

    [ L transition blocks | T terminal rows of D floats | count ]

and travels as a whole in the chunk's collective, so an episode end is a valid transition on EVERY rank -- these
are exactly the transitions that carry the -5000 * unvisited / -1000 * starved terminal penalties
(reward_function.py:59-67).  A terminal row is addressed by its ticket = the value of the chunk's counter when
the kernel claimed it (row = ticket mod T; `count` = the counter's final value travels with the section): a
row has been overwritten by a later episode end of the same chunk iff count - ticket > T, which `_next_frame`
reports as valid = False instead of handing out another environment's observation.  The counter is never reset:
it keeps counting through every recycling of its chunk (the slots holding tickets of an earlier cycle have left the
sampling window by then), so a chunk's graph is nothing but its L step launches.  T defaults to E: every
environment of the shard can end inside one chunk (they all start together: with the base configuration all
environments truncate within ~60 steps of each other) without losing a row.
"""
import torch
import torch.distributed as dist


def _pad4(n):
    return (int(n) + 3) & ~3


class _ChunkGraphs(list):
    """The graphs capture_chunks returns + the environment's launch epoch at capture time."""
    epoch = 0


class TransitionRing:
    def __init__(self, capacity, envs_per_rank, obs_dim, device, world_size=1, rank=0, group=None, terminal_rows=None,
                 chunk_len=1, always_exchange=False):
        """chunk_len: slots per exchange.  1 = every commit() all-gathers its slot in place.  L > 1 = the ring is
        split into capacity/L chunks and a chunk is gathered (one in-place collective of L blocks + the terminal
        section per rank) when its last slot is committed: L times fewer, L times larger collectives -- at ~10 us
        per step the per-call host cost of a collective is several steps long -- and the natural unit for HIP-graph
        replay (`capture_chunks`).  terminal_rows: T, terminal rows per (chunk, rank); default E (see above).

        A chunk is recycled as a whole: once the head enters it, its older slots stop being sampled (their terminal
        rows are about to be overwritten), so the ring holds between capacity - L and capacity - 1 slots."""
        self.capacity, self.E, self.D = int(capacity), int(envs_per_rank), int(obs_dim)
        self.L = int(chunk_len)
        assert self.L >= 1 and self.capacity % self.L == 0
        self.n_chunks = self.capacity // self.L
        assert self.n_chunks >= 2 or self.L == 1, "a chunked ring needs at least two chunks"
        self.world, self.rank, self.group = int(world_size), int(rank), group
        self.exchange = self.world > 1 or bool(always_exchange)   # always_exchange: run the collective even alone (self-test)
        self.device = torch.device(device)
        self.T = int(terminal_rows) if terminal_rows is not None else self.E
        assert self.T >= 1
        # [chunk][rank][ L blocks | terminal rows | count ]: a rank's part of a chunk is contiguous and the chunk is the
        # concatenation of the ranks' parts => ONE in-place all-gather per chunk.  Every piece starts 16-byte aligned
        # (the kernel stores aux rows as float4).
        self.obs_floats = _pad4(self.E * self.D)
        self.block = self.obs_floats + self.E * 4
        self.term_off = self.L * self.block
        self.count_off = self.term_off + _pad4(self.T * self.D)
        self.section = self.count_off + 4
        self.store = torch.zeros(self.n_chunks, self.world, self.section, dtype=torch.float32, device=self.device)
        blocks = self.store[..., :self.term_off].view(self.n_chunks, self.world, self.L, self.block)
        self._obs5 = blocks[..., :self.E * self.D].view(self.n_chunks, self.world, self.L, self.E, self.D)
        self._aux5 = blocks[..., self.obs_floats:].view(self.n_chunks, self.world, self.L, self.E, 4)
        self._term4 = self.store[..., self.term_off:self.term_off + self.T * self.D].view(self.n_chunks, self.world, self.T, self.D)
        self._i32 = self.store.view(torch.int32)
        self._count = self._i32[..., self.count_off]                      # [chunk][rank]
        self._ticket5 = self._i32[..., :self.term_off].view(self.n_chunks, self.world, self.L, self.block)[..., self.obs_floats:] \
            .view(self.n_chunks, self.world, self.L, self.E, 4)[..., 3]  # aux word 3 as the int32 it is
        if self.L == 1:      # the plain [slot][rank][env] views
            self.obs = self._obs5.squeeze(2)
            self.aux = self._aux5.squeeze(2)
        self._env = None
        self.head = 0                 # next slot to write
        self.size = 0                 # number of sampleable slots behind the head
        self._pending = [None] * self.n_chunks     # outstanding collective per chunk
        self._comm_stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self._used_comm = False

    # ---- the terminal section of a chunk ---------------------------------------------------------------
    def local_terminal_section(self, chunk=None):
        """(rows [T, D], counter int32 [1]) of THIS rank's terminal section of a chunk: what
        BatchedUAVEnv.set_terminal_pool takes."""
        c = self.head // self.L if chunk is None else int(chunk)
        return self._term4[c, self.rank], self._i32[c, self.rank, self.count_off:self.count_off + 1]

    def _point_env(self, slot=None):
        """Aim the attached environment's aux / terminal outputs at a slot (default: the head)."""
        slot = self.head if slot is None else slot
        pool, counter = self.local_terminal_section(slot // self.L)
        self._env.set_terminal_pool(pool, counter, None)
        self._env.set_aux_output(self.local_aux_slot(slot))

    def attach(self, env):
        """Let `env` (BatchedUAVEnv) write terminal observations into the head chunk's terminal section and the
        (action, reward, done, terminal ticket) part of every transition straight into the ring."""
        self._env = env
        self._point_env()

    # ---- producer side -----------------------------------------------------------------------
    def _cj(self, slot):
        return slot // self.L, slot % self.L

    def obs_at(self, slot, r, e):
        """Observation rows by (slot, rank, env) index tensors (or ints / slices)."""
        return self._obs5[slot // self.L, r, slot % self.L, e]

    def aux_at(self, slot, r, e):
        return self._aux5[slot // self.L, r, slot % self.L, e]

    def tickets_at(self, slot, r, e):
        """Terminal tickets (int32; -1 where the step into `slot` did not end an episode)."""
        return self._ticket5[slot // self.L, r, slot % self.L, e]

    def terminal_at(self, slot, r, ticket):
        """Terminal observation rows of the chunk holding `slot` for tickets of that chunk (row = ticket mod T)."""
        return self._term4[slot // self.L, r, ticket % self.T]

    def local_obs_slot(self, slot=None):
        """[E, D] view of THIS rank's part of a slot: pass it as `obs_out` to BatchedUAVEnv.step*()."""
        c, j = self._cj(self.head if slot is None else slot)
        return self._obs5[c, self.rank, j]

    def local_aux_slot(self, slot=None):
        c, j = self._cj(self.head if slot is None else slot)
        return self._aux5[c, self.rank, j]

    def wait_chunk(self, c):
        w = self._pending[c]
        if w is not None:
            w.wait()
            self._pending[c] = None

    def wait_slot(self, slot):
        self.wait_chunk(slot // self.L)

    def _gather_chunk(self, c):
        """One in-place all-gather of this rank's part of chunk c (RCCL on a side stream / gloo on CPU)."""
        out, inp = self.store[c].view(-1), self.store[c, self.rank].view(-1)
        if self._comm_stream is not None:
            self._used_comm = True
            self._comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._comm_stream):
                w = dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)
        else:   # CPU / gloo (tests)
            w = dist.all_gather_into_tensor(out, inp.clone(), group=self.group, async_op=True)
        self._pending[c] = w

    def _advance(self, n):
        """Move the head by n committed slots; gather every chunk that was completed; recycle the chunk the head
        enters (its previous gather must be over; its old slots leave the sampling window, and with them the tickets
        of its previous cycle -- the terminal counter just keeps counting)."""
        for _ in range(n):
            slot = self.head
            c, j = self._cj(slot)
            if self.exchange and j == self.L - 1:
                self._gather_chunk(c)
            self.head = (slot + 1) % self.capacity
            if self.head % self.L == 0:
                nc = self.head // self.L
                self.wait_chunk(nc)
            self.size = min(self.size + 1, self.capacity - self.L + self.head % self.L)
        self.wait_slot(self.head)

    def commit(self, actions=None, reward=None, done=None, terminal_obs=None):
        """Publish this rank's block of the current slot.  With an attached env the kernel has already written
        the aux part and the terminal rows; otherwise (tests, foreign producers) pass actions / reward / done
        (and terminal_obs [E, D] for the rows where done) to fill them here the way the kernel does."""
        slot = self.head
        if actions is not None:
            c, j = self._cj(slot)
            aux = self.local_aux_slot(slot)
            aux[:, 0] = actions.to(torch.float32)
            aux[:, 1] = reward.to(torch.float32)
            aux[:, 2] = done.to(torch.float32)
            tick = self._ticket5[c, self.rank, j]
            tick.fill_(-1)
            if terminal_obs is not None:
                idx = torch.nonzero(done.to(torch.bool)).flatten()
                if idx.numel():
                    base = int(self._count[c, self.rank])
                    t = base + torch.arange(idx.numel(), device=self.device)
                    tick[idx] = t.to(torch.int32)
                    self._term4[c, self.rank, t % self.T] = terminal_obs[idx].to(torch.float32)
                    self._count[c, self.rank] = base + idx.numel()
        self._advance(1)
        if self._env is not None:
            self._point_env()
        return slot

    # ---- launch-bound producer loops: one HIP graph per chunk -----------------------------------------------
    def capture_chunks(self, step_fn):
        """Capture, for every chunk, "L x (re-point outputs, `step_fn(obs_slot)`)" into one
        HIP graph (the head must stand at the start of a chunk).  `replay_chunk(graphs)` then costs one graph launch
        (plus the chunk's collective when ranks share the ring, issued from the host after the replay) instead of L
        Python -> ctypes -> hipLaunchKernel round trips of ~12 us each -- more than the step kernel itself at 4096
        environments.  `step_fn(obs_out)` must only enqueue work on the current stream (e.g. `env.step_random`)."""
        assert self.device.type == "cuda" and self._env is not None and self.head % self.L == 0
        self.drain()
        torch.cuda.synchronize(self.device)
        head0, graphs = self.head, _ChunkGraphs([None] * self.n_chunks)
        graphs.epoch = self._env.launch_epoch
        for k in range(self.n_chunks):
            c = (head0 // self.L + k) % self.n_chunks
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for j in range(self.L):
                    slot = c * self.L + j
                    self._point_env(slot)
                    step_fn(self.local_obs_slot(slot))
            graphs[c] = g
        self._point_env()          # capturing executed nothing
        return graphs

    def run_chunk_random(self):
        """The chunk at the head stepped with the in-kernel random policy: L single-step launches issued by ONE call into the
        C library (BatchedUAVEnv.step_random_n) -- no graph to capture, and the first kernel starts after one packet."""
        assert self._env is not None and self.head % self.L == 0
        c = self.head // self.L
        self.wait_chunk(c)
        self._point_env(c * self.L)                     # terminal section of this chunk
        self._env.step_random_n(self.L, self._obs5[c, self.rank, 0], self.block, self._aux5[c, self.rank, 0], self.block)
        self._advance(self.L)
        self._point_env()

    def replay_chunk(self, graphs):
        """Replay the graph of the chunk at the head and commit its L slots."""
        c = self.head // self.L
        assert self.head % self.L == 0
        if getattr(graphs, "epoch", self._env.launch_epoch) != self._env.launch_epoch:
            raise RuntimeError("these chunk graphs were captured before env.seed() / set_noise_tape() / enable_terminal_snapshot(): "
                               "their launches still carry the old values -- capture_chunks() again")
        self.wait_chunk(c)            # its previous gather (a full revolution ago) must be done before it is overwritten;
        graphs[c].replay()            # the gather of the chunk just before this one keeps running on the side stream
        self._advance(self.L)
        self._point_env()

    def drain(self):
        for c in range(self.n_chunks):
            self.wait_chunk(c)
        if self._comm_stream is not None and self._used_comm:        # (a cross-stream wait costs several us of GPU time even
            torch.cuda.current_stream(self.device).wait_stream(self._comm_stream)   # when the side stream never ran anything)

    # ---- consumer side -----------------------------------------------------------------------
    def sampleable(self):
        """Slots behind the head that can be sampled on this rank: all committed ones, minus -- when ranks share a
        chunked ring -- those of the head chunk, whose other-rank parts have not been gathered yet."""
        return self.size - (self.head % self.L if (self.exchange and self.L > 1) else 0)

    def window_state(self):
        """(n, oldest): the number of sampleable slots and the ring position of the oldest one -- the two host integers a
        draw depends on.  `_draw(..., window=(n_dev, oldest_dev))` takes them as 0-d int64 device tensors instead, which
        makes a sampling step replayable as a captured graph (the caller refreshes the two tensors before each replay)."""
        n = self.sampleable()
        assert self.size - n >= 0
        return n, (self.head - self.size) % self.capacity

    def _draw(self, batch_size, generator, window=None):
        if window is not None:
            n_dev, oldest_dev = window
            # j uniform on 0 .. n-2 (the transition out of the newest sampleable slot has no successor yet)
            u = torch.rand(batch_size, generator=generator, device=self.device, dtype=torch.float64)
            j = torch.minimum((u * (n_dev - 1).to(torch.float64)).long(), n_dev - 2)
            r = torch.randint(0, self.world, (batch_size,), generator=generator, device=self.device)
            e = torch.randint(0, self.E, (batch_size,), generator=generator, device=self.device)
            return j, (oldest_dev + j) % self.capacity, r, e
        n = self.sampleable()
        skip = self.size - n                       # committed but not yet visible slots right behind the head
        oldest = (self.head - self.size) % self.capacity
        j = torch.randint(0, n - 1, (batch_size,), generator=generator, device=self.device)
        r = torch.randint(0, self.world, (batch_size,), generator=generator, device=self.device)
        e = torch.randint(0, self.E, (batch_size,), generator=generator, device=self.device)
        assert skip >= 0
        return j, (oldest + j) % self.capacity, r, e

    def _next_frame(self, slot, r, e):
        """(action, reward, done, newest frame of next_obs, valid) of the transitions slot -> slot+1."""
        nxt = (slot + 1) % self.capacity
        c, j = nxt // self.L, nxt % self.L
        aux = self._aux5[c, r, j, e]               # the aux row stored WITH an observation describes the step INTO it
        done = aux[:, 2] > 0.5
        ticket = self._ticket5[c, r, j, e].long()
        count = self._count[c, r].long()
        have_term = done & (ticket >= 0) & (count - ticket <= self.T)      # the row has not been overwritten since
        row = ticket.clamp(min=0) % self.T
        last = torch.where(have_term.unsqueeze(1), self._term4[c, r, row], self._obs5[c, r, j, e])
        return aux[:, 0].long(), aux[:, 1], done, last, ~done | have_term

    def _stacked_batch_hip(self, j, slot, r, e, k, batch_size):
        """stacked_batch_at in ONE launch (csrc/uavenv_replay.hip:uavenv_ring_gather_stacked) instead of ~40 indexing launches;
        same bits (tests/test_gpu_api.py compares the two)."""
        import ctypes as C
        from . import _native as N
        if self.__dict__.get("_layout") is None:
            self._layout = N.UavRingLayout(section=self.section, num_chunks=self.n_chunks, world=self.world, slots_per_chunk=self.L,
                                           envs=self.E, obs_dim=self.D, terminal_rows=self.T, block=self.block,
                                           obs_floats=self.obs_floats, term_off=self.term_off, count_off=self.count_off)
            self._glib = N.lib()
        dev = self.device
        idx = [t.to(torch.int64).contiguous() for t in (j, slot, r, e)]
        obs = torch.empty(batch_size, k * self.D, dtype=torch.float32, device=dev)
        nxt = torch.empty(batch_size, k * self.D, dtype=torch.float32, device=dev)
        action = torch.empty(batch_size, dtype=torch.int64, device=dev)
        reward = torch.empty(batch_size, dtype=torch.float32, device=dev)
        done = torch.empty(batch_size, dtype=torch.bool, device=dev)
        valid = torch.empty(batch_size, dtype=torch.bool, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = self._glib.uavenv_ring_gather_stacked(p(self.store), C.byref(self._layout), p(idx[0]), p(idx[1]), p(idx[2]), p(idx[3]),
                                                   batch_size, k, p(obs), p(nxt), p(action), p(reward), p(done), p(valid),
                                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc:
            raise RuntimeError(f"uavenv_ring_gather_stacked failed ({rc})")
        return dict(obs=obs, action=action, reward=reward, done=done, next_obs=nxt, valid=valid, index=(j, slot, r, e))

    def sample_stacked_keyed(self, batch_size, n_stack, window, counter, seed, out=None):
        """sample_stacked with the draw made inside the gather kernel (uavenv_ring_sample_stacked): `window` = int64 cuda tensor [2]
        (sampleable slots, position of the oldest), `counter` = float32 cuda scalar that differs from draw to draw, `seed` = the
        Philox key.  One launch in all; replayable as part of a captured graph (the caller drains pending gathers and refreshes
        `window` before each replay).  `out`: a dict of preallocated output tensors from an earlier call (reused in place)."""
        import ctypes as C
        from . import _native as N
        if self.__dict__.get("_layout") is None:
            self._layout = N.UavRingLayout(section=self.section, num_chunks=self.n_chunks, world=self.world, slots_per_chunk=self.L,
                                           envs=self.E, obs_dim=self.D, terminal_rows=self.T, block=self.block,
                                           obs_floats=self.obs_floats, term_off=self.term_off, count_off=self.count_off)
            self._glib = N.lib()
        dev, k, B = self.device, int(n_stack), int(batch_size)
        assert window.is_cuda and window.dtype == torch.int64 and window.numel() == 2 and counter.is_cuda and counter.dtype == torch.float32
        if out is None:
            out = dict(obs=torch.empty(B, k * self.D, dtype=torch.float32, device=dev), next_obs=torch.empty(B, k * self.D, dtype=torch.float32, device=dev),
                       action=torch.empty(B, dtype=torch.int64, device=dev), reward=torch.empty(B, dtype=torch.float32, device=dev),
                       done=torch.empty(B, dtype=torch.bool, device=dev), valid=torch.empty(B, dtype=torch.bool, device=dev),
                       index_block=torch.empty(4, B, dtype=torch.int64, device=dev))
            out["index"] = tuple(out["index_block"][i] for i in range(4))
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = self._glib.uavenv_ring_sample_stacked(p(self.store), C.byref(self._layout), p(window), p(counter), int(seed) & 0xFFFFFFFFFFFFFFFF, B, k,
                                                   p(out["obs"]), p(out["next_obs"]), p(out["action"]), p(out["reward"]), p(out["done"]),
                                                   p(out["valid"]), p(out["index_block"]),
                                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc:
            raise RuntimeError(f"uavenv_ring_sample_stacked failed ({rc})")
        return out

    def sample(self, batch_size, generator=None):
        """Uniform sample of transitions (obs, action, reward, done, next_obs, valid) over all ranks' envs.
        Slot s holds the observation s_t together with (a, r, done) of the step that PRODUCED it, so the
        transition out of slot s reads its action / reward / done from slot s+1."""
        assert self.sampleable() >= 2
        self.drain()
        _, slot, r, e = self._draw(batch_size, generator)
        action, reward, done, last, valid = self._next_frame(slot, r, e)
        return dict(obs=self.obs_at(slot, r, e), action=action, reward=reward, done=done, next_obs=last, valid=valid)

    def sample_stacked(self, batch_size, n_stack, generator=None, window=None):
        """Like sample(), but observations are frame stacks of `n_stack` frames gathered from the ring on the
        fly (SB3 VecFrameStack layout: [oldest | ... | newest], frames from before the episode start zeroed),
        so the replay stores each frame ONCE instead of n_stack times (dqn.py:1085 budgets 2 x 612 floats per
        transition for the stacked copies)."""
        assert self.sampleable() >= 2
        if window is None:
            self.drain()            # (a captured draw -- `window` given -- is replayed later: its caller drains before each replay)
        return self.stacked_batch_at(*self._draw(batch_size, generator, window), n_stack)

    def stacked_batch_at(self, j, slot, r, e, n_stack):
        """The stacked transitions at drawn positions: j = age rank inside the sampling window (0 = oldest), slot = ring slot,
        r = rank, e = environment (int64 tensors [B]).  The result carries them back as `index` (tests replay a draw)."""
        k, batch_size = int(n_stack), int(slot.numel())
        if self.store.is_cuda and k <= 16:
            return self._stacked_batch_hip(j, slot, r, e, k, batch_size)
        return self.stacked_batch_at_torch(j, slot, r, e, n_stack)

    def stacked_batch_at_torch(self, j, slot, r, e, n_stack):
        """stacked_batch_at as tensor expressions: what CPU rings run, and the statement the HIP gather is tested against."""
        k, batch_size = int(n_stack), int(slot.numel())
        back = torch.arange(k - 1, -1, -1, device=self.device)                      # k-1 ... 0 (oldest first)
        fs = (slot.unsqueeze(1) - back.unsqueeze(0)) % self.capacity                # [B, k] frame slots
        in_ring = (j.unsqueeze(1) - back.unsqueeze(0)) >= 0                         # frame older than the ring start?
        rr, ee = r.unsqueeze(1).expand(-1, k), e.unsqueeze(1).expand(-1, k)
        frames = self.obs_at(fs, rr, ee)                                            # [B, k, D]
        dn = self.aux_at(fs, rr, ee)[..., 2] > 0.5                                  # frame is the first of an episode
        # frame i (i < k-1) is valid iff no episode start among frames i+1 .. k-1
        later_start = torch.flip(torch.cumsum(torch.flip(dn[:, 1:], [1]).int(), 1), [1]) > 0     # [B, k-1]
        valid_f = torch.cat([~later_start, torch.ones(batch_size, 1, dtype=torch.bool, device=self.device)], 1) & in_ring
        frames = frames * valid_f.unsqueeze(2)
        action, reward, done, last, valid = self._next_frame(slot, r, e)
        # next stack = [frames 1..k-1 | newest]; after an auto-reset the real next state is the terminal observation
        next_obs = torch.cat([frames[:, 1:].reshape(batch_size, (k - 1) * self.D), last], 1)
        return dict(obs=frames.reshape(batch_size, k * self.D), action=action, reward=reward, done=done,
                    next_obs=next_obs, valid=valid, index=(j, slot, r, e))
