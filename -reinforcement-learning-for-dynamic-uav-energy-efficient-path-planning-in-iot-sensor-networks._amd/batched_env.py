"""BatchedUAVEnv: E independent UAV-IoT environments stepped by ONE HIP kernel launch.

Host-side mirror of the reference's environment interface (uav_env.py:240-488) in batched,
device-resident form.  All arrays handed out are torch tensors on the GPU (views of persistent
buffers owned by this object); the numpy-returning adapters that SB3 / gymnasium callers expect
live in vec_env.py and gym_env.py on top of this class.

torch is plumbing here (device memory + the current HIP stream); every computation of the path
happens in libuavenv_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N

# reference constructor kwargs (uav_env.py:266-287) -> UavEnvConfig fields
_REF_KWARGS = {
    "data_generation_rate": "data_generation_rate", "max_buffer_size": "max_buffer_size",
    "rssi_threshold": "rssi_threshold", "sensor_duty_cycle": "duty_cycle", "max_battery": "max_battery",
    "collection_duration": "collection_duration", "max_steps": "max_steps",
    "penalty_data_loss": "penalty_data_loss", "reward_urgency_reduction": "reward_urgency_reduction",
    "penalty_battery": "penalty_battery", "reward_movement": "reward_movement",
}
# accepted and ignored, like the reference does: lora_spreading_factor is overwritten to 12 by every
# reset (iot_sensors.py:309), path_loss_exponent is stored but never read (iot_sensors.py:72 vs :147-197)
_REF_IGNORED = ("lora_spreading_factor", "path_loss_exponent", "render_mode")


def config_from_kwargs(**kw):
    """Build a UavEnvConfig from the reference's kwargs and/or raw UavEnvConfig field names."""
    kw = dict(kw)
    over = {}
    for k in _REF_IGNORED:
        kw.pop(k, None)
    if "grid_size" in kw:
        over["grid_size"] = kw.pop("grid_size")
    if "uav_start_position" in kw:
        sp = kw.pop("uav_start_position")
        if sp is not None:
            over["start_x"], over["start_y"] = float(sp[0]), float(sp[1])
    if "include_sensor_positions" in kw:
        over["include_sensor_positions"] = int(bool(kw.pop("include_sensor_positions")))
    for ref, mine in _REF_KWARGS.items():
        if ref in kw:
            over[mine] = kw.pop(ref)
    over.update(kw)          # anything else must be a UavEnvConfig field (TypeError otherwise)
    return N.default_config(**over)


class BatchedUAVEnv:
    """E environment instances on one GPU.

    Parameters
    ----------
    num_envs : number of environment instances of this shard
    env_index_base : global index of the first instance (multi-GPU sharding keeps results invariant)
    auto_reset : SB3 VecEnv semantics -- an environment that truncates is reset inside the same step
    sensor_positions : optional [E, N, 2] (or [N, 2] shared) array replacing the Philox layouts
    **config : the reference's constructor kwargs (uav_env.py:266-287) and/or UavEnvConfig fields
    """

    def __init__(self, num_envs, device=None, env_index_base=0, auto_reset=True, sensor_positions=None,
                 flags=0, **config):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedUAVEnv needs a HIP device; there is no CPU fallback")
        self.L = N.lib()
        if sensor_positions is not None and "num_sensors" not in config:
            config["num_sensors"] = int(np.asarray(sensor_positions).shape[-2])
        cfg = config_from_kwargs(**config)
        cfg.flags = int(flags) | int(cfg.flags) | (N.FLAG_AUTO_RESET if auto_reset else 0)
        self.cfg = cfg
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.num_envs = int(num_envs)
        self.env_index_base = int(env_index_base)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            N.check(self.L.uavenv_create(C.byref(cfg), self.num_envs, self.env_index_base, self.device.index, C.byref(h)))
        self._h = h
        self.lane_stride = self.L.uavenv_lane_stride(h)
        self.obs_dim = self.L.uavenv_env_obs_dim(h)
        self.num_sensors = cfg.num_sensors
        E, D = self.num_envs, self.obs_dim
        dev = self.device
        # what a host caller wants back after every step -- observation, float32 reward, done flag -- lives in ONE
        # allocation, so that the numpy-returning adapters move it with a single device-to-host copy (vec_env.py)
        nb = E * D * 4 + E * 4 + E
        self.out_block = torch.zeros((nb + 15) // 16 * 16, dtype=torch.uint8, device=dev)
        self.obs = self.out_block[:E * D * 4].view(torch.float32).view(E, D)
        self.reward32 = self.out_block[E * D * 4:E * D * 4 + E * 4].view(torch.float32)
        self.done = self.out_block[E * D * 4 + E * 4:E * D * 4 + E * 4 + E]
        self.terminal_obs = torch.zeros(E, D, dtype=torch.float32, device=dev)
        self.reward = torch.zeros(E, dtype=torch.float64, device=dev)
        self.actions_taken = torch.zeros(E, dtype=torch.int32, device=dev)
        self._tapes = (None, None)
        # counts the calls that change what a step launch carries BY VALUE and a captured HIP graph therefore keeps in its
        # old form (include/uavenv.h:uavenv_set_seed): holders of captured graphs compare it (replay.py, learner.py)
        self.launch_epoch = 0
        if sensor_positions is not None:
            self.set_positions(sensor_positions)

    # ---- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            torch.cuda.synchronize(self.device)
            self.L.uavenv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t):
        return C.c_void_p(0 if t is None else t.data_ptr())

    # ---- configuration ----------------------------------------------------------------------
    def set_positions(self, pos):
        """pos: [E, N, 2] or [N, 2] (uav_env.py:269 sensor_positions)."""
        pos = np.asarray(pos, dtype=np.float32)
        if pos.ndim == 2:
            pos = np.broadcast_to(pos, (self.num_envs,) + pos.shape)
        E, G = self.num_envs, self.lane_stride
        px = np.zeros((E, G), np.float32); py = np.zeros((E, G), np.float32)
        px[:, :pos.shape[1]] = pos[:, :, 0]; py[:, :pos.shape[1]] = pos[:, :, 1]
        torch.cuda.synchronize(self.device)
        N.check(self.L.uavenv_set_positions(self._h, px.ctypes.data_as(C.c_void_p), py.ctypes.data_as(C.c_void_p)), self._h)

    def set_env_params(self, grid_w=None, grid_h=None, num_sensors=None):
        a = lambda x: None if x is None else np.ascontiguousarray(np.broadcast_to(np.asarray(x, np.int32), (self.num_envs,)))
        gw, gh, ns = a(grid_w), a(grid_h), a(num_sensors)
        p = lambda x: C.c_void_p(0) if x is None else x.ctypes.data_as(C.c_void_p)
        torch.cuda.synchronize(self.device)
        N.check(self.L.uavenv_set_env_params(self._h, p(gw), p(gh), p(ns)), self._h)

    def seed(self, seed):
        N.check(self.L.uavenv_set_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF), self._h)
        self.launch_epoch += 1

    def set_config(self, **overrides):
        """Change constants of the LIVE environments (uavenv_set_config): the reference's kwargs and/or UavEnvConfig field names,
        e.g. `set_config(shadowing_std_db=8.0)` for what sim_to_real_sweep.py:113-117 does to every sensor.  Takes effect
        with the next step; captured graphs must be captured again (launch_epoch)."""
        cfg = N.UavEnvConfig()
        N.check(self.L.uavenv_get_config(self._h, C.byref(cfg)), self._h)
        over = {}
        for k, v in overrides.items():
            if k in _REF_IGNORED:
                continue
            over[_REF_KWARGS.get(k, k)] = v
        N.apply_overrides(cfg, over)
        N.check(self.L.uavenv_set_config(self._h, C.byref(cfg)), self._h)
        self.cfg = cfg
        self.launch_epoch += 1

    def set_grid_choices(self, grids):
        w = (C.c_int32 * 8)(*[int(g[0]) for g in grids]); h = (C.c_int32 * 8)(*[int(g[1]) for g in grids])
        N.check(self.L.uavenv_set_grid_choices(self._h, len(grids), w, h), self._h)

    def set_noise_tape(self, step_tape=None, reset_tape=None):
        """step_tape: float32 cuda [E, 7, lane_stride]; reset_tape: [E, 4, lane_stride]; None = Philox."""
        for t, slots in ((step_tape, 7), (reset_tape, 4)):
            if t is not None:
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                assert tuple(t.shape) == (self.num_envs, slots, self.lane_stride), tuple(t.shape)
        self._tapes = (step_tape, reset_tape)        # keep alive
        N.check(self.L.uavenv_set_noise_tape(self._h, self._p(step_tape), self._p(reset_tape)), self._h)
        self.launch_epoch += 1

    def set_terminal_pool(self, pool=None, counter=None, index_out=None):
        """pool float32 cuda [rows, D]; counter int32/uint32 cuda [1]; index_out int32 cuda [E] (or all None)."""
        if pool is not None:
            assert pool.is_cuda and pool.dtype == torch.float32 and pool.is_contiguous() and pool.shape[1] == self.obs_dim
            assert counter.is_cuda and counter.numel() == 1 and counter.element_size() == 4
            assert index_out is None or (index_out.is_cuda and index_out.dtype == torch.int32 and index_out.numel() == self.num_envs)
        self._term_pool = (pool, counter, index_out)     # keep alive
        N.check(self.L.uavenv_set_terminal_pool(self._h, self._p(pool), 0 if pool is None else pool.shape[0],
                                                self._p(counter), self._p(index_out)), self._h)

    def set_aux_output(self, aux=None):
        """aux float32 cuda [E, 4] (or [K, E, 4] for rollouts of up to K steps): (action, reward, done, terminal ticket as
        int32 bits) per step; None disables."""
        cap = 0
        if aux is not None:
            assert aux.is_cuda and aux.dtype == torch.float32 and aux.is_contiguous() and aux.shape[-1] == 4
            cap = aux.numel() // (4 * self.num_envs)
            assert cap >= 1 and aux.numel() == cap * 4 * self.num_envs, tuple(aux.shape)
        self._aux = aux
        N.check(self.L.uavenv_set_aux_output(self._h, self._p(aux), cap), self._h)

    def enable_terminal_snapshot(self, enable=True):
        """Keep, for every environment that ends an episode, the state its terminal step's `info` is made of
        (uavenv_enable_terminal_snapshot): read it with `terminal_snapshot(indices)`."""
        N.check(self.L.uavenv_enable_terminal_snapshot(self._h, 1 if enable else 0), self._h)
        self._term_snapshot = bool(enable)
        self.launch_epoch += 1

    def terminal_snapshot(self, indices):
        """(records [k] structured, sensors [k, 3, lane_stride] float64 = buffer / generated / transmitted) of the last
        terminal step of the environments `indices`."""
        idx = torch.as_tensor(np.asarray(indices, np.int64), device=self.device)
        rec = self.get_state(N.F_TERM_RECORD)[idx].cpu().numpy().view(N.record_dtype()).reshape(len(idx))
        sens = self.get_state(N.F_TERM_SENSORS).view(self.num_envs, 3, self.lane_stride)[idx].cpu().numpy()
        return rec, sens

    def dump_noise(self):
        E, G = self.num_envs, self.lane_stride
        st = torch.empty(E, 7, G, dtype=torch.float32, device=self.device)
        rt = torch.empty(E, 4, G, dtype=torch.float32, device=self.device)
        N.check(self.L.uavenv_dump_noise(self._h, self._p(st), self._p(rt), self._stream()), self._h)
        return st, rt

    # ---- the hot path -------------------------------------------------------------------------
    def reset(self, mask=None):
        """uav_env.py:400-427 for every (masked) environment; returns the observation tensor [E, D]."""
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        N.check(self.L.uavenv_reset(self._h, self._p(mask), self._p(self.obs), self._stream()), self._h)
        return self.obs

    def _obs_target(self, obs_out):
        if obs_out is None:
            return self.obs
        assert obs_out.is_cuda and obs_out.dtype == torch.float32 and obs_out.is_contiguous()
        assert tuple(obs_out.shape) == (self.num_envs, self.obs_dim), tuple(obs_out.shape)
        return obs_out

    def _hot_args(self, obs):
        """(obs, reward, reward32, done, terminal_obs, stream) as plain integers: the per-step launches go through here,
        and building eight ctypes objects per call costs more host time than the argument marshalling itself."""
        fixed = self.__dict__.get("_fixed_ptrs")
        if fixed is None or fixed[0] is not self.terminal_obs:
            fixed = (self.terminal_obs, self.reward.data_ptr(), self.reward32.data_ptr(), self.done.data_ptr(),
                     self.terminal_obs.data_ptr(), self.actions_taken.data_ptr())
            self._fixed_ptrs = fixed
        return fixed, obs.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream

    def step(self, actions, obs_out=None):
        """uav_env.py:429-488 for all E environments in one launch.
        actions: int32 cuda tensor [E].  Returns (obs, reward float64, done uint8); with auto_reset the
        rows of `self.terminal_obs` are valid where done.  obs_out: optional [E, D] destination (e.g. a
        slot of a replay ring) written directly by the kernel instead of `self.obs`."""
        assert actions.is_cuda and actions.dtype == torch.int32 and actions.numel() == self.num_envs
        obs = self._obs_target(obs_out)
        f, po, st = self._hot_args(obs)
        rc = self.L.uavenv_step(self._h, actions.data_ptr(), po, f[1], f[2], f[3], f[4], st)
        if rc:
            N.check(rc, self._h)
        return obs, self.reward, self.done

    def step_random(self, obs_out=None):
        """Same with the uniform-random policy drawn in-kernel; the actions land in `self.actions_taken`."""
        obs = self._obs_target(obs_out)
        f, po, st = self._hot_args(obs)
        rc = self.L.uavenv_step_random(self._h, f[5], po, f[1], f[2], f[3], f[4], st)
        if rc:
            N.check(rc, self._h)
        return obs, self.reward, self.done

    def step_random_n(self, num_steps, obs_first, obs_stride, aux_first=None, aux_stride=0):
        """`num_steps` random-policy steps issued by ONE call into the C library (uavenv_step_random_n): step k writes its
        observations at obs_first + k * obs_stride floats (and its aux block at aux_first + k * aux_stride), e.g. the slots of a
        replay-ring chunk.  Same results as calling step_random() num_steps times; `self.reward32` / `self.done` hold the last
        step's values."""
        assert obs_first.is_cuda and obs_first.dtype == torch.float32
        rc = self.L.uavenv_step_random_n(self._h, int(num_steps), obs_first.data_ptr(), int(obs_stride),
                                         0 if aux_first is None else aux_first.data_ptr(), int(aux_stride),
                                         self.reward32.data_ptr(), self.done.data_ptr(),
                                         torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            N.check(rc, self._h)

    def step_policy(self, policy, obs_out=None):
        """Step with the action chosen in the kernel: N.POLICY_RANDOM, N.POLICY_NEAREST (NearestSensorGreedy) or
        N.POLICY_MAX_THROUGHPUT_V2 (greedy_agents.py); the actions land in `self.actions_taken`."""
        obs = self._obs_target(obs_out)
        N.check(self.L.uavenv_step_policy(self._h, int(policy), self._p(self.actions_taken), self._p(obs), self._p(self.reward),
                                          self._p(self.reward32), self._p(self.done), self._p(self.terminal_obs),
                                          self._stream()), self._h)
        return obs, self.reward, self.done

    def rollout(self, num_steps, actions=None, obs_out=None, with_terminal=False, policy=None):
        """K steps in ONE launch (uavenv_rollout): `actions` int32 cuda [K, E], or an in-kernel `policy`
        (default N.POLICY_RANDOM when no actions are given).  Returns dict(obs [K,E,D], reward [K,E] f64, reward32, done [K,E] u8, actions [K,E] i32
        (+ terminal_obs [K,E,D] when with_terminal)).  Bit-identical to K step() calls."""
        K, E, D, dev = int(num_steps), self.num_envs, self.obs_dim, self.device
        if actions is not None:
            assert actions.is_cuda and actions.dtype == torch.int32 and tuple(actions.shape) == (K, E) and actions.is_contiguous()
        obs = obs_out if obs_out is not None else torch.empty(K, E, D, dtype=torch.float32, device=dev)
        assert obs.is_contiguous() and tuple(obs.shape) == (K, E, D) and obs.dtype == torch.float32
        out = dict(obs=obs, reward=torch.empty(K, E, dtype=torch.float64, device=dev),
                   reward32=torch.empty(K, E, dtype=torch.float32, device=dev),
                   done=torch.empty(K, E, dtype=torch.uint8, device=dev),
                   actions=torch.empty(K, E, dtype=torch.int32, device=dev))
        term = torch.zeros(K, E, D, dtype=torch.float32, device=dev) if with_terminal else None
        if policy is None:
            policy = N.POLICY_ACTIONS if actions is not None else N.POLICY_RANDOM
        N.check(self.L.uavenv_rollout(self._h, K, int(policy), self._p(actions), self._p(out["actions"]), self._p(obs),
                                      self._p(out["reward"]), self._p(out["reward32"]), self._p(out["done"]),
                                      self._p(term), self._stream()), self._h)
        if actions is not None:
            out["actions"] = actions
        if with_terminal:
            out["terminal_obs"] = term
        if obs_out is None:
            self.obs.copy_(obs[K - 1])
        return out

    def time_steps(self, steps):
        """Average milliseconds per step launch, HIP events on the launch stream (bench.py)."""
        ms = C.c_float()
        N.check(self.L.uavenv_time_steps(self._h, int(steps), self._p(self.obs), self._p(self.reward), self._p(self.done),
                                         self._stream(), C.byref(ms)), self._h)
        return ms.value

    def check_actions(self):
        """Raise ValueError if any step since the last check saw an action outside 0..4 (uav_env.py:468)."""
        rec = self.records()
        if (rec["status"] & 1).any():
            rec["status"] &= ~np.uint32(1)
            self.set_state(N.F_RECORD, torch.from_numpy(rec.view(np.uint8).reshape(self.num_envs, -1)))
            raise ValueError("Invalid action: outside 0..4")

    # ---- state access -------------------------------------------------------------------------
    _FIELD_DTYPES = {N.F_POS_X: torch.float32, N.F_POS_Y: torch.float32, N.F_BUFFER: torch.float64,
                     N.F_GEN: torch.float64, N.F_TX: torch.float64, N.F_LOST: torch.float64,
                     N.F_AVG_RSSI: torch.float64, N.F_FLAGS: torch.int32, N.F_TERM_SENSORS: torch.float64}

    def get_state(self, field):
        """Device copy of one state field: [E, lane_stride] tensor (or raw bytes [E, 128] for records)."""
        nbytes = self.L.uavenv_state_bytes(self._h, field)
        if field in self._FIELD_DTYPES:
            cols = self.lane_stride * (3 if field == N.F_TERM_SENSORS else 1)
            t = torch.empty(self.num_envs, cols, dtype=self._FIELD_DTYPES[field], device=self.device)
        else:
            t = torch.empty(self.num_envs, nbytes // self.num_envs, dtype=torch.uint8, device=self.device)
        N.check(self.L.uavenv_get_state(self._h, field, self._p(t), nbytes, 1, self._stream()), self._h)
        return t

    def set_state(self, field, t):
        nbytes = self.L.uavenv_state_bytes(self._h, field)
        t = t.to(self.device).contiguous()
        assert t.numel() * t.element_size() == nbytes, (t.shape, nbytes)
        N.check(self.L.uavenv_set_state(self._h, field, self._p(t), nbytes, 1, self._stream()), self._h)
        torch.cuda.current_stream(self.device).synchronize()

    def records(self):
        """numpy structured array [E] of UavEnvRecord (include/uavenv.h)."""
        raw = self.get_state(N.F_RECORD).cpu().numpy()
        return raw.view(N.record_dtype()).reshape(self.num_envs).copy()

    def episode_stats(self):
        raw = self.get_state(N.F_EPISODE_STATS).cpu().numpy()
        return raw.view(N.episode_stats_dtype()).reshape(self.num_envs).copy()

    def state_dict(self):
        """Everything needed to resume (checkpoint) or to compare against the oracle."""
        names = dict(pos_x=N.F_POS_X, pos_y=N.F_POS_Y, buffer=N.F_BUFFER, gen=N.F_GEN, tx=N.F_TX, lost=N.F_LOST,
                     avg_rssi=N.F_AVG_RSSI, flags=N.F_FLAGS, record=N.F_RECORD, episode_stats=N.F_EPISODE_STATS)
        return {k: self.get_state(f).cpu() for k, f in names.items()}

    def load_state_dict(self, sd):
        names = dict(pos_x=N.F_POS_X, pos_y=N.F_POS_Y, buffer=N.F_BUFFER, gen=N.F_GEN, tx=N.F_TX, lost=N.F_LOST,
                     avg_rssi=N.F_AVG_RSSI, flags=N.F_FLAGS, record=N.F_RECORD, episode_stats=N.F_EPISODE_STATS)
        for k, f in names.items():
            self.set_state(f, sd[k])

    def sensor_state(self, env=None):
        """Decoded per-sensor state as numpy (for one env or all): the IoTSensor attributes callers read."""
        sl = slice(None) if env is None else env
        n = self.num_sensors
        g = lambda f: self.get_state(f).cpu().numpy()[sl][..., :n]
        flags = g(N.F_FLAGS)
        avg = g(N.F_AVG_RSSI).copy()
        avg[(flags & 16) == 0] = np.nan
        return dict(pos_x=g(N.F_POS_X), pos_y=g(N.F_POS_Y), buffer=g(N.F_BUFFER), gen=g(N.F_GEN), tx=g(N.F_TX),
                    lost=g(N.F_LOST), avg_rssi=avg, sf=(flags & 15).astype(np.int32),
                    visited=((flags >> 5) & 1).astype(np.uint8), data_collected=((flags >> 6) & 1).astype(np.uint8))
