"""SB3-compatible vectorised environment over the HIP kernel.

`UAVVecEnv` honours the Stable-Baselines3 `VecEnv` contract the reference's trainer relies on
(agents/dqn/dqn.py:1276-1278, :924, :946): `num_envs`, `observation_space`, `action_space`,
`reset() -> obs[E, D]`, `step_async(actions)` / `step_wait() -> (obs, rewards, dones, infos)` with
auto-reset, `infos[i]["terminal_observation"]` and `infos[i]["TimeLimit.truncated"]` on episode
ends, plus `get_attr / set_attr / env_method / env_is_wrapped / seed / close`.  It replaces
`DummyVecEnv([Monitor(DomainRandEnv(...))] * 4)`: Monitor's `info["episode"] = {"r", "l", "t"}` and
DomainRandEnv's `last_episode_stats` are produced by the kernel when an episode ends.

The numpy arrays SB3 wants are host copies made at this outermost adapter only: observation, reward and
done flags of a step sit in ONE device allocation and cross PCIe in ONE copy into pinned host memory
(`host_copies=False` hands out views of a small rotation of those pinned buffers instead of fresh arrays).
`step_tensors()` is the zero-copy device API for GPU-resident learners.
"""
import time

import numpy as np
import torch

from . import _native as N
from . import spaces
from .batched_env import BatchedUAVEnv
from .gym_env import CURRICULUM_STAGES, MAX_SENSORS_LIMIT
from .info import build_info

try:  # pragma: no cover
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv
except Exception:
    _SB3VecEnv = None


class _VecEnvBase:
    """Duck-typed stand-in for stable_baselines3.common.vec_env.VecEnv when SB3 is not installed."""

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space
        self.render_mode = None

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()


_Base = _SB3VecEnv if _SB3VecEnv is not None else _VecEnvBase


class UAVVecEnv(_Base):
    """E UAV-IoT environments behind the SB3 VecEnv interface.

    domain_rand=True reproduces what the reference trainer drives (DomainRandEnv, dqn.py:177-451):
    curriculum grid sampling, fresh layouts, far start, shaping, Jain's bonus, obs padded to 50 slots.
    """

    def __init__(self, num_envs, domain_rand=False, curriculum_stage=0, max_sensors_limit=MAX_SENSORS_LIMIT,
                 device=None, env_index_base=0, n_stack=1, num_sensors=None, host_copies=True, host_buffers=4, **config):
        """num_sensors: one count for every environment, or a sequence of per-environment counts (cycled over the
        environments) -- the reference trainer pins each of its workers to a different count, WORKER_SENSOR_COUNTS =
        [10, 20, 30, 40] (dqn.py:1065, :1223-1234); observations then carry max(counts) slots (zero padded, or
        `max_sensors_limit` slots under domain_rand).
        host_copies: True = step() returns fresh numpy arrays like SB3's DummyVecEnv; False = views of `host_buffers`
        rotating pinned buffers: an array stays valid until `host_buffers - 1` further steps (SB3's own loops copy what
        they keep at once), which saves a 2.5 MB host memcpy per step at 4096 x 50."""
        flags = 0
        per_env = None
        if num_sensors is not None:
            if np.ndim(num_sensors) == 0:
                config["num_sensors"] = int(num_sensors)
            else:
                counts = [int(c) for c in num_sensors]
                per_env = np.array([counts[i % len(counts)] for i in range(num_envs)], np.int32)
                config["num_sensors"] = int(per_env.max())
        if domain_rand:
            flags = N.FLAG_RANDOM_LAYOUT | N.FLAG_FAR_START | N.FLAG_PROX_SHAPING | N.FLAG_JAIN_BONUS
            config.setdefault("pad_sensors", max_sensors_limit)
            config.setdefault("grid_choices", CURRICULUM_STAGES[min(curriculum_stage, len(CURRICULUM_STAGES) - 1)][0])
            config.setdefault("grid_size", CURRICULUM_STAGES[0][0][0])
        self.env = BatchedUAVEnv(num_envs, device=device, env_index_base=env_index_base, auto_reset=True,
                                 flags=flags, **config)
        if per_env is not None:
            self.env.set_env_params(num_sensors=per_env)
        self.env.enable_terminal_snapshot(True)          # episode-end infos carry the terminal step's _get_info() keys
        self._domain_rand = domain_rand
        self._curriculum_stage = curriculum_stage
        # n_stack > 1 folds SB3's VecFrameStack(n_stack) (dqn.py:1278) into this env: the stack is kept on the GPU
        # by uavenv_frame_stack and only the stacked observation is copied to the host.
        self.n_stack = int(n_stack)
        self._fs = None
        if self.n_stack > 1:
            from .frame_stack import FrameStack
            self._fs = FrameStack(num_envs, self.env.obs_dim, self.n_stack, self.env.device)
        D = self.env.obs_dim * self.n_stack
        if domain_rand:
            obs_space = spaces.Box(low=-np.inf, high=np.inf, shape=(D,), dtype=np.float32)      # dqn.py:252-254
        else:
            obs_space = spaces.Box(low=np.full(D, -1.0, np.float32), high=np.ones(D, np.float32), dtype=np.float32)
        super().__init__(num_envs, obs_space, spaces.Discrete(5))
        self._actions = torch.zeros(num_envs, dtype=torch.int32, device=self.env.device)
        # host staging: pinned buffers the step's output block (obs | reward32 | done) is copied into with ONE transfer;
        # with a frame stack the stacked observation is a second, separate transfer
        self._host_copies = bool(host_copies)
        nbuf = 1 if self._host_copies else max(2, int(host_buffers))
        self._stage = [torch.empty(self.env.out_block.numel(), dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
        self._stage_obs = ([torch.empty(num_envs, D, dtype=torch.float32).pin_memory() for _ in range(nbuf)]
                           if self._fs is not None else None)
        self._actions_host = torch.empty(num_envs, dtype=torch.int32).pin_memory()
        self._k = 0
        self._empty_infos = [{} for _ in range(num_envs)]
        self._t0 = time.time()
        self._last_stats = [None] * num_envs
        self._pending = False

    # ---- SB3 VecEnv contract --------------------------------------------------------------------
    def reset(self):
        obs = self.env.reset()
        if self._fs is not None:
            obs = self._fs.reset(obs)
        return obs.cpu().numpy()

    def step_async(self, actions):
        a = np.asarray(actions)
        if a.shape != (self.num_envs,):
            a = a.reshape(self.num_envs)
        if ((a < 0) | (a > 4)).any():
            raise ValueError(f"Invalid action: {a[(a < 0) | (a > 4)][0]}")           # uav_env.py:468
        self._actions_host.numpy()[:] = a
        self._actions.copy_(self._actions_host, non_blocking=True)
        self.env.step(self._actions)                                                  # one kernel launch, async
        self._pending = True

    def _to_host(self):
        """One device-to-host transfer of (obs | reward32 | done) into pinned memory; returns numpy (obs, rewards, dones)."""
        env, E = self.env, self.num_envs
        D1 = env.obs_dim
        k = self._k % len(self._stage)
        self._k += 1
        stage = self._stage[k]
        stage.copy_(env.out_block, non_blocking=True)
        obs_t = None
        if self._fs is not None:
            obs_t = self._stage_obs[k]
            obs_t.copy_(self._fs.step(env.obs, env.done, env.terminal_obs), non_blocking=True)
        torch.cuda.current_stream(env.device).synchronize()
        raw = stage.numpy()
        o = E * D1 * 4
        obs = obs_t.numpy() if obs_t is not None else raw[:o].view(np.float32).reshape(E, D1)
        rews = raw[o:o + E * 4].view(np.float32)
        dones = raw[o + E * 4:o + E * 4 + E].view(np.bool_)
        if self._host_copies:
            return obs.copy(), rews.copy(), dones.copy()
        return obs, rews, dones

    def step_wait(self):
        assert self._pending
        self._pending = False
        env = self.env
        obs, rews, dones = self._to_host()
        # infos of environments that did not end an episode are empty: handing out the SAME empty dicts every step
        # (a fresh list of them) instead of creating num_envs dicts saves ~100 us per step at 4096 environments --
        # callers must treat the info of a running environment as read-only
        infos = self._empty_infos.copy()
        if dones.any():
            idx = np.nonzero(dones)[0]
            src = self._fs.terminal_stacked if self._fs is not None else env.terminal_obs
            didx = torch.from_numpy(idx).to(env.device)
            term = src[didx].cpu().numpy()
            stats = env.episode_stats()
            trec, tsens = env.terminal_snapshot(idx)
            now = round(time.time() - self._t0, 6)
            for j, i in enumerate(idx):
                st = stats[i]
                n = int(st["num_sensors"])
                # the reference's own info of the terminal step (uav_env.py:676-700: what BestByMetricCallback reads at
                # dqn.py:1150-1155 -- total_data_collected, battery, sensor_collection_ratios -- and the other 14 keys) ...
                info = build_info(env.cfg, trec[j], tsens[j, 0], tsens[j, 1], tsens[j, 2], int(st["sensors_visited"]), n)
                # ... plus what SB3's VecEnv / Monitor add on an episode end
                info["terminal_observation"] = term[j]
                info["TimeLimit.truncated"] = True           # `terminated` is always False (uav_env.py:471)
                info["episode"] = {"r": float(st["episode_return"]), "l": int(st["length"]), "t": now}
                les = self._episode_stats_dict(st)
                info["last_episode_stats"] = les
                self._last_stats[i] = les
                infos[i] = info
        return obs, rews, dones, infos

    def step_tensors(self, actions):
        """Device-resident fast path: int32 cuda tensor in, (obs, reward32, done) cuda tensors out, no sync."""
        obs, _, done = self.env.step(actions)
        return obs, self.env.reward32, done

    def _episode_stats_dict(self, st):
        """dqn.py:316-331 last_episode_stats keys."""
        tg, tc = float(st["total_generated"]), float(st["total_collected"])
        used = float(self.env.cfg.max_battery) - float(st["battery_remaining"])
        cov = int(st["first_full_coverage_step"])
        return {"total_generated": tg, "total_collected": tc, "total_lost": float(st["total_lost"]),
                "battery_remaining": float(st["battery_remaining"]),
                "ndr": int(st["sensors_visited"]) / int(st["num_sensors"]) * 100,
                "fairness_std": float(st["fairness_std"]), "jains_index": float(st["jains_index"]),
                "grid_size": (int(st["grid_w"]), int(st["grid_h"])), "num_sensors": int(st["num_sensors"]),
                "data_efficiency": (tc / tg * 100) if tg > 0 else 0.0,
                "bytes_per_wh": (tc / used) if used > 0 else 0.0,
                # dqn.py:302 clears `_first_full_coverage_step` before the snapshot at :330 reads it: always None in the
                # reference; the step the kernel tracked (dqn.py:428-431) is kept under its own key
                "time_to_coverage": None, "first_full_coverage_step": cov if cov >= 0 else None}

    def close(self):
        self.env.close()

    def seed(self, seed=None):
        if seed is not None:
            self.env.seed(seed)
        return [None if seed is None else seed + i for i in range(self.num_envs)]

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, int):
            return [indices]
        return list(indices)

    def get_attr(self, attr_name, indices=None):
        idx = self._indices(indices)
        if attr_name == "last_episode_stats":                                         # dqn.py:924
            return [self._last_stats[i] for i in idx]
        rec = None
        if attr_name in ("current_step", "total_reward", "total_data_collected", "boundary_hits", "edge_steps"):
            rec = self.env.records()
            return [rec[attr_name][i].item() for i in idx]
        if attr_name == "num_sensors":
            return [int(self.env.records()["num_sensors"][i]) for i in idx]
        if attr_name == "grid_size":
            rec = self.env.records()
            return [(int(rec["grid_w"][i]), int(rec["grid_h"][i])) for i in idx]
        if attr_name == "render_mode":
            return [None for _ in idx]
        if hasattr(self, attr_name):
            return [getattr(self, attr_name) for _ in idx]
        raise AttributeError(attr_name)

    def set_attr(self, attr_name, value, indices=None):
        raise AttributeError(f"UAVVecEnv has no settable per-env attribute {attr_name!r}")

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        idx = self._indices(indices)
        if method_name == "set_curriculum_stage":                                     # dqn.py:946
            stage = int(np.clip(method_args[0], 0, len(CURRICULUM_STAGES) - 1))
            self._curriculum_stage = stage
            self.env.set_grid_choices(CURRICULUM_STAGES[stage][0])
            return [None for _ in idx]
        raise AttributeError(method_name)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode=None):
        return None
