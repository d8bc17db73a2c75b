"""Single-environment view with the reference's own interface.

`UAVEnvironment` mirrors /root/reference/src/environment/uav_env.py:240-488: same constructor
kwargs (:266-287), `reset(seed, options) -> (obs, info)`, `step(action) -> (obs, reward, terminated,
truncated, info)`, `action_space = Discrete(5)`, `observation_space = Box(-1, 1, (3 + fps*N,))`, and the
attributes callers of the reference reach into (`env.uav`, `env.sensors[i]`, `env.sensors_visited`,
`env.total_data_collected`, ... -- SURVEY.md section 1).  `DomainRandEnv` mirrors
agents/dqn/dqn.py:177-451.  Both are thin host code over a 1-environment `BatchedUAVEnv`: every
step is one launch of the same HIP kernel that steps 4096 environments.
"""
import numpy as np
import torch

from . import _native as N
from . import spaces
from .batched_env import BatchedUAVEnv
from .info import build_info

try:  # pragma: no cover
    import gymnasium as _gym
    _Base = _gym.Env
except Exception:
    class _Base:          # gymnasium is absent in the build container
        metadata = {}


# agents/dqn/dqn.py:63-69
CURRICULUM_STAGES = [
    ([(100, 100)], [20, 30, 40], "Stage 0 - 100x100 only"),
    ([(100, 100), (200, 200)], [20, 30, 40], "Stage 1 - up to 200x200"),
    ([(100, 100), (200, 200), (300, 300)], [20, 30, 40], "Stage 2 - up to 300x300"),
    ([(100, 100), (200, 200), (300, 300), (400, 400)], [10, 30, 40], "Stage 3 - up to 400x400"),
    ([(100, 100), (200, 200), (300, 300), (400, 400), (500, 500)], [10, 20, 30, 40], "Stage 4 - full feasible range"),
]
MAX_SENSORS_LIMIT = 50                    # dqn.py:122
SF_DATA_RATES = {7: 5470 / 8, 8: 3125 / 8, 9: 1760 / 8, 10: 980 / 8, 11: 440 / 8, 12: 250 / 8}   # iot_sensors.py:13-20


class UAVView:
    """What callers read from `env.uav` (uav.py:63-263)."""

    def __init__(self, env):
        self._env = env

    @property
    def position(self):
        r = self._env._record()
        return np.array([r["uav_x"], r["uav_y"]], dtype=np.float32)

    @property
    def start_position(self):
        r = self._env._record()
        return np.array([r["start_x"], r["start_y"]], dtype=np.float32)

    @property
    def battery(self):
        return float(self._env._record()["battery"])

    @property
    def max_battery(self):
        return float(self._env._cfg.max_battery)

    @property
    def power_move(self):
        return float(self._env._cfg.power_move)

    @property
    def power_hover(self):
        return float(self._env._cfg.power_hover)

    @property
    def battery_drain_hover(self):
        return self.power_hover / (60 * 60)

    def is_alive(self):
        return self.battery > (self._env._cfg.alive_fraction * self.max_battery)      # uav.py:224

    def get_battery_percentage(self):
        return (self.battery / self.max_battery) * 100

    def __repr__(self):
        p = self.position
        return f"UAV(position=({p[0]}, {p[1]}), battery={self.battery:.2f}Wh/{self.max_battery}Wh)"


class SensorView:
    """What callers read from `env.sensors[i]` (iot_sensors.py:66-103)."""

    def __init__(self, env, i):
        self._env, self.sensor_id = env, i

    def _g(self, key):
        return self._env._sensors_snapshot()[key][self.sensor_id]

    position = property(lambda s: np.array([s._g("pos_x"), s._g("pos_y")], dtype=np.float32))
    data_buffer = property(lambda s: float(s._g("buffer")))
    total_data_generated = property(lambda s: float(s._g("gen")))
    total_data_transmitted = property(lambda s: float(s._g("tx")))
    total_data_lost = property(lambda s: float(s._g("lost")))
    spreading_factor = property(lambda s: int(s._g("sf")))
    data_collected = property(lambda s: bool(s._g("data_collected")))
    data_rate = property(lambda s: SF_DATA_RATES[int(s._g("sf"))])
    max_buffer_size = property(lambda s: float(s._env._cfg.max_buffer_size))
    data_generation_rate = property(lambda s: float(s._env._cfg.data_generation_rate))
    rssi_threshold = property(lambda s: float(s._env._cfg.rssi_threshold))
    duty_cycle = property(lambda s: float(s._env._cfg.duty_cycle))
    duty_cycle_probability = property(lambda s: float(s._env._cfg.duty_cycle) / 100.0)

    # The channel constants live once per environment, not per sensor: writing one through ANY sensor changes it for all of
    # them -- which is what the reference's sweeps do anyway (`for s in env.sensors: s.shadowing_std_db = sigma`,
    # sim_to_real_sweep.py:113-117).
    def _cfg_property(field):
        def get(s):
            return float(getattr(s._env._cfg, field))

        def set_(s, value):
            s._env._set_config(**{field: float(value)})
        return property(get, set_)

    shadowing_std_db = _cfg_property("shadowing_std_db")
    transmit_power_dbm = _cfg_property("tx_power_dbm")
    noise_floor_dbm = _cfg_property("noise_floor_dbm")
    adr_lambda = _cfg_property("adr_lambda")
    path_loss_exponent = property(lambda s: 3.8, lambda s, v: None)     # stored by IoTSensor, read by nothing (iot_sensors.py:72)
    del _cfg_property

    @property
    def avg_rssi(self):
        v = self._g("avg_rssi")
        return None if np.isnan(v) else float(v)

    def calculate_rssi(self, uav_position):
        """iot_sensors.py:147-197 on the host (used by heuristic baselines between steps; draws its
        shadowing sample from the env's host-side np_random, not from the device noise stream)."""
        c = self._env._cfg
        sp = self.position
        up = np.array(uav_position, dtype=np.float32)
        dx = (up[0] - sp[0]) * np.float32(10)
        dy = (up[1] - sp[1]) * np.float32(10)
        ground = np.sqrt(dx * dx + dy * dy)
        d = np.sqrt(ground * ground + np.float32(c.uav_altitude * c.uav_altitude))
        d_break = (4 * np.pi * c.sensor_height * c.uav_altitude) / c.wavelength
        l10 = np.float32(np.log10(np.float64(d)))
        if d < d_break:
            pl = (np.float64(np.float32(20) * l10) + 20 * np.log10(c.freq_mhz)) - c.fspl_offset_db
        else:
            pl = (np.float64(np.float32(40) * l10) - 20 * np.log10(c.sensor_height)) - 20 * np.log10(c.uav_altitude)
        return float(c.tx_power_dbm - pl + c.shadowing_std_db * self._env.np_random.standard_normal())

    def is_in_range(self, uav_position):
        return self.calculate_rssi(uav_position) >= self.rssi_threshold             # iot_sensors.py:214-219


class UAVEnvironment(_Base):
    """Drop-in for the reference's `UAVEnvironment` (uav_env.py:240)."""

    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 4}

    def __init__(self, grid_size=(10, 10), sensor_positions=None, num_sensors=20, data_generation_rate=22.0 / 10,
                 max_buffer_size=1000.0, lora_spreading_factor=7, path_loss_exponent=2.0, rssi_threshold=-85.0,
                 sensor_duty_cycle=10.0, uav_start_position=None, max_battery=274.0, collection_duration=1.0,
                 max_steps=2100, render_mode=None, penalty_data_loss=-1.0, reward_urgency_reduction=20.0,
                 penalty_battery=-0.5, reward_movement=10.0, include_sensor_positions=False,
                 device=None, seed=0, env_index=0, **extra_config):
        self.grid_size = tuple(grid_size)
        self.max_steps = max_steps
        self.render_mode = render_mode
        self.collection_duration = collection_duration
        if sensor_positions is not None:
            num_sensors = len(sensor_positions)
        self.num_sensors = int(num_sensors)
        self.include_sensor_positions = bool(include_sensor_positions)
        self._features_per_sensor = 5 if include_sensor_positions else 3
        kw = dict(grid_size=self.grid_size, num_sensors=self.num_sensors, data_generation_rate=data_generation_rate,
                  max_buffer_size=max_buffer_size, rssi_threshold=rssi_threshold, sensor_duty_cycle=sensor_duty_cycle,
                  uav_start_position=uav_start_position, max_battery=max_battery, collection_duration=collection_duration,
                  max_steps=max_steps, penalty_data_loss=penalty_data_loss,
                  reward_urgency_reduction=reward_urgency_reduction, penalty_battery=penalty_battery,
                  reward_movement=reward_movement, include_sensor_positions=include_sensor_positions, seed=seed)
        kw.update(extra_config)
        pos = None if sensor_positions is None else np.asarray(sensor_positions, np.float32).reshape(1, -1, 2)
        self._benv = BatchedUAVEnv(1, device=device, env_index_base=env_index, auto_reset=False,
                                   sensor_positions=pos, **kw)
        self._cfg = self._benv.cfg
        self.action_space = spaces.Discrete(5)                                      # uav_env.py:347
        D = self._benv.obs_dim
        self.observation_space = spaces.Box(low=np.full(D, -1.0, np.float32), high=np.ones(D, np.float32),
                                            dtype=np.float32)                       # uav_env.py:348-355
        self.uav = UAVView(self)
        self.sensors = [SensorView(self, i) for i in range(self.num_sensors)]
        self.np_random = np.random.default_rng(seed)
        self.last_action = None
        self._rec = None
        self._snap = None
        self._act = torch.zeros(1, dtype=torch.int32, device=self._benv.device)

    def _set_config(self, **overrides):
        self._benv.set_config(**overrides)
        self._cfg = self._benv.cfg

    # ---- cached device state --------------------------------------------------------------------
    def _invalidate(self):
        self._rec = None
        self._snap = None

    def _record(self):
        if self._rec is None:
            self._rec = self._benv.records()[0]
        return self._rec

    def _sensors_snapshot(self):
        if self._snap is None:
            self._snap = {k: v[0] for k, v in self._benv.sensor_state().items()}
        return self._snap

    sensor_positions = property(lambda s: [(float(x), float(y)) for x, y in
                                           zip(s._sensors_snapshot()["pos_x"], s._sensors_snapshot()["pos_y"])])
    current_step = property(lambda s: int(s._record()["current_step"]))
    total_reward = property(lambda s: float(s._record()["total_reward"]))
    total_data_collected = property(lambda s: float(s._record()["total_data_collected"]))
    capture_effect_triggers = property(lambda s: int(s._record()["capture_triggers"]))
    boundary_hits = property(lambda s: int(s._record()["boundary_hits"]))
    edge_steps = property(lambda s: int(s._record()["edge_steps"]))
    last_step_bytes_collected = property(lambda s: float(s._record()["last_step_bytes"]))
    sensors_visited = property(lambda s: set(int(i) for i in np.nonzero(s._sensors_snapshot()["visited"])[0]))

    # ---- gymnasium API ----------------------------------------------------------------------------
    def reset(self, seed=None, options=None):
        """uav_env.py:400-427.  `seed` re-keys the device noise (replaces gymnasium's np_random seeding)."""
        if seed is not None:
            self._benv.seed(seed)
            self.np_random = np.random.default_rng(seed)
        obs = self._benv.reset()
        self.last_action = None
        self._invalidate()
        return obs[0].cpu().numpy().copy(), self._get_info()

    def step(self, action):
        """uav_env.py:429-488; `terminated` is always False, truncation = battery <= 2 % or max_steps."""
        self.last_action = action
        a = int(action) if isinstance(action, (int, np.integer)) or np.ndim(action) == 0 else -1
        self._act[0] = a if -2**31 <= a < 2**31 else -1
        obs, rew, done = self._benv.step(self._act)
        self._invalidate()
        if not (0 <= a <= 4):
            self._benv.check_actions()                      # raises ValueError after the ageing, like uav_env.py:468
            raise ValueError(f"Invalid action: {action}")
        out = obs[0].cpu().numpy().copy()
        return out, float(rew[0].item()), False, bool(done[0].item()), self._get_info()

    def _get_sensor_urgencies(self):
        s = self._sensors_snapshot()
        r = self._cfg.data_generation_rate
        return (s["buffer"] / r).astype(np.float32) if r > 0 else np.zeros(self.num_sensors, np.float32)   # uav_env.py:386-394

    def _get_info(self):
        """uav_env.py:676-700 (17 keys)."""
        s = self._sensors_snapshot()
        return build_info(self._cfg, self._record(), s["buffer"], s["gen"], s["tx"], int(s["visited"].sum()), self.num_sensors)

    def render(self):
        """Visualisation (uav_env.py:706-895) is outside the hot path: no-op."""
        return None

    def close(self):
        self._benv.close()


class DomainRandEnv(UAVEnvironment):
    """Drop-in for agents/dqn/dqn.py:177-451: per-reset grid randomisation from the curriculum stage,
    fresh uniform layout with empty buffers, far-start rejection sampling, observation zero-padded to
    `max_sensors_limit` slots, proximity shaping and the per-step Jain's bonus -- all executed inside
    the HIP kernel (flags of the C ABI)."""

    def __init__(self, fixed_num_sensors, max_sensors_limit=MAX_SENSORS_LIMIT, curriculum_stage=0, base_config=None,
                 **kwargs):
        self.max_sensors_limit = max_sensors_limit
        self._curriculum_stage = min(curriculum_stage, len(CURRICULUM_STAGES) - 1)
        self._fixed_num_sensors = fixed_num_sensors
        cfg = dict(base_config or {})
        cfg.update(kwargs)
        cfg.setdefault("grid_size", CURRICULUM_STAGES[0][0][0])
        flags = N.FLAG_RANDOM_LAYOUT | N.FLAG_FAR_START | N.FLAG_PROX_SHAPING | N.FLAG_JAIN_BONUS
        super().__init__(num_sensors=fixed_num_sensors, pad_sensors=max_sensors_limit, flags=flags,
                         grid_choices=CURRICULUM_STAGES[self._curriculum_stage][0], **cfg)
        D = self._benv.obs_dim
        self.observation_space = spaces.Box(low=-np.inf, high=np.inf, shape=(D,), dtype=np.float32)   # dqn.py:252-254
        self.last_episode_stats = None
        self.first_full_coverage_step = None

    def set_curriculum_stage(self, stage):
        """dqn.py:258-277: only the grid list sampled at the next reset changes."""
        self._curriculum_stage = int(np.clip(stage, 0, len(CURRICULUM_STAGES) - 1))
        self._benv.set_grid_choices(CURRICULUM_STAGES[self._curriculum_stage][0])

    @staticmethod
    def _jains(rates):                                                              # dqn.py:446-451
        n, s1, s2 = len(rates), sum(rates), sum(x ** 2 for x in rates)
        return (s1 ** 2) / (n * s2) if n > 0 and s2 > 0 else 1.0

    def reset(self, **kwargs):
        if self.current_step > 0:                                                   # dqn.py:305-331
            s, r = self._sensors_snapshot(), self._record()
            rates = [t / g * 100 for t, g in zip(s["tx"], s["gen"]) if g > 0]
            tg, tc = float(s["gen"].sum()), float(s["tx"].sum())
            used = self._cfg.max_battery - float(r["battery"])
            cov = int(r["first_full_coverage_step"])
            self.first_full_coverage_step = cov if cov >= 0 else None      # what dqn.py:428-431 tracked for the episode
            self.last_episode_stats = {
                "total_generated": tg, "total_collected": tc, "total_lost": float(s["lost"].sum()),
                "battery_remaining": float(r["battery"]),
                "ndr": int(s["visited"].sum()) / self.num_sensors * 100,
                "fairness_std": float(np.std(rates)) if rates else 0.0,
                "jains_index": self._jains(rates),
                "grid_size": (int(r["grid_w"]), int(r["grid_h"])), "num_sensors": self.num_sensors,
                "data_efficiency": (tc / tg * 100) if tg > 0 else 0.0,
                "bytes_per_wh": (tc / used) if used > 0 else 0.0,
                # dqn.py:302 sets `_first_full_coverage_step = None` BEFORE the snapshot at :330 reads it, so the reference's
                # dict always carries None here; the tracked step is kept in `self.first_full_coverage_step`
                "time_to_coverage": None,
            }
        out = super().reset(**kwargs)
        r = self._record()
        self.grid_size = (int(r["grid_w"]), int(r["grid_h"]))
        return out
