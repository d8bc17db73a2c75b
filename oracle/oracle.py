"""ctypes binding of the CPU oracle (oracle/uavenv_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does (tests/test_boundary.py greps for that).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
def _host_has_fma():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return " fma " in (line + " ")
    except OSError:
        pass
    return False


# same source, same results; the -mfma build inlines the specification's fma() calls (oracle/Makefile)
LIB_NAME = "liborc.so" if _host_has_fma() else "liborc_nofma.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
MAX_SENSORS = 64

FLAG_RANDOM_LAYOUT, FLAG_FAR_START, FLAG_PROX_SHAPING, FLAG_JAIN_BONUS = 1, 2, 4, 8
POLICY_NEAREST, POLICY_MAX_THROUGHPUT_V2 = 2, 3

_D = C.c_double


class OrcConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("grid_w", C.c_int32), ("grid_h", C.c_int32),
        ("num_sensors", C.c_int32), ("max_steps", C.c_int32), ("include_sensor_positions", C.c_int32),
        ("pad_sensors", C.c_int32), ("flags", C.c_uint32), ("max_start_tries", C.c_int32),
        ("use_ema_adr", C.c_int32), ("num_grid_choices", C.c_int32),
        ("grid_choices_w", C.c_int32 * 8), ("grid_choices_h", C.c_int32 * 8),
        ("seed", C.c_uint64),
        ("data_generation_rate", _D), ("max_buffer_size", _D), ("rssi_threshold", _D), ("duty_cycle", _D),
        ("start_x", _D), ("start_y", _D), ("max_battery", _D), ("collection_duration", _D),
        ("tx_power_dbm", _D), ("noise_floor_dbm", _D), ("uav_altitude", _D), ("sensor_height", _D),
        ("wavelength", _D), ("freq_mhz", _D), ("fspl_offset_db", _D), ("adr_lambda", _D),
        ("shadowing_std_db", _D), ("capture_threshold_db", _D),
        ("sf_thresholds", _D * 4), ("fill_lo", _D), ("fill_hi", _D),
        ("power_move", _D), ("power_hover", _D), ("alive_fraction", _D),
        ("reward_per_byte", _D), ("reward_new_sensor", _D), ("reward_completion", _D),
        ("reward_urgency_reduction", _D), ("reward_movement", _D), ("penalty_revisit", _D),
        ("penalty_boundary", _D), ("penalty_collision", _D), ("penalty_battery", _D),
        ("penalty_hover", _D), ("penalty_step", _D), ("penalty_data_loss", _D),
        ("penalty_starvation", _D), ("penalty_unvisited", _D), ("penalty_starved", _D),
        ("starvation_cr_threshold", _D),
        ("min_start_dist", _D), ("prox_eta", _D), ("jain_weight", _D),
    ]


_F64S = C.c_double * MAX_SENSORS


class OrcEnv(C.Structure):
    _fields_ = [
        ("cfg", OrcConfig), ("n", C.c_int32), ("grid_w", C.c_int32), ("grid_h", C.c_int32),
        ("env_index", C.c_uint32), ("episode", C.c_uint32),
        ("pos_x", C.c_float * MAX_SENSORS), ("pos_y", C.c_float * MAX_SENSORS),
        ("buffer", _F64S), ("gen", _F64S), ("tx", _F64S), ("lost", _F64S),
        ("avg_rssi", _F64S), ("cur_rssi", _F64S),
        ("sf", C.c_int32 * MAX_SENSORS),
        ("avg_valid", C.c_uint8 * MAX_SENSORS), ("visited", C.c_uint8 * MAX_SENSORS),
        ("data_collected", C.c_uint8 * MAX_SENSORS),
        ("uav_x", C.c_float), ("uav_y", C.c_float), ("start_x", C.c_float), ("start_y", C.c_float),
        ("battery", _D), ("previous_data_loss", _D), ("total_reward", _D),
        ("total_data_collected", _D), ("last_step_bytes", _D), ("prev_dist_nearest", _D),
        ("current_step", C.c_int32), ("capture_triggers", C.c_int32), ("boundary_hits", C.c_int32),
        ("edge_steps", C.c_int32), ("collisions_total", C.c_int32),
        ("first_full_coverage_step", C.c_int32),
    ]


class OrcEpisodeStats(C.Structure):
    _fields_ = [(k, _D) for k in ("total_generated", "total_collected", "total_lost", "battery_remaining", "ndr",
                                  "fairness_std", "jains_index", "data_efficiency", "bytes_per_wh")] + \
               [(k, C.c_int32) for k in ("grid_w", "grid_h", "num_sensors", "rated", "length", "first_full_coverage_step")]


_lib = None


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "uavenv_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-B", "all"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.orc_default_config.argtypes = [C.POINTER(OrcConfig)]
        L.orc_obs_dim.argtypes = [C.POINTER(OrcConfig), C.c_int]
        L.orc_obs_dim.restype = C.c_int
        L.orc_init.argtypes = [C.POINTER(OrcEnv), C.POINTER(OrcConfig), C.c_uint32, fp, fp]
        L.orc_reset_tape.argtypes = [C.POINTER(OrcEnv), fp, fp]
        L.orc_step_tape.argtypes = [C.POINTER(OrcEnv), C.c_int, fp, fp, C.POINTER(_D), C.POINTER(C.c_int)]
        L.orc_step_tape.restype = C.c_int
        L.orc_reset_keyed.argtypes = [C.POINTER(OrcEnv), fp]
        L.orc_step_keyed.argtypes = [C.POINTER(OrcEnv), C.c_int, fp, C.POINTER(_D), C.POINTER(C.c_int)]
        L.orc_step_keyed.restype = C.c_int
        L.orc_noise_step_tape.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, fp]
        L.orc_noise_reset_tape.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, fp]
        L.orc_noise_positions.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, fp, fp]
        L.orc_noise_action.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_noise_action.restype = C.c_int
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_philox4x32.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32)]
        L.orc_philox_rounds.restype = C.c_int
        L.orc_normal_pair.argtypes = [C.c_uint32, C.c_uint32, fp, fp]
        L.orc_run_random_policy.argtypes = [C.POINTER(OrcConfig), C.c_int, C.c_uint32, C.c_int, C.POINTER(_D)]
        L.orc_run_random_policy.restype = C.c_long
        L.orc_policy_action.argtypes = [C.POINTER(OrcEnv), C.c_int, fp]
        L.orc_policy_action.restype = C.c_int
        L.orc_step_policy_tape.argtypes = [C.POINTER(OrcEnv), C.c_int, fp, fp, C.POINTER(_D), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_step_policy_tape.restype = C.c_int
        L.orc_step_policy_keyed.argtypes = [C.POINTER(OrcEnv), C.c_int, fp, C.POINTER(_D), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_step_policy_keyed.restype = C.c_int
        L.orc_trace_keyed.argtypes = [C.POINTER(OrcConfig), C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_trace_keyed.restype = C.c_long
        L.orc_rssi_deterministic.argtypes = [C.POINTER(OrcConfig), C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_rssi_deterministic.restype = _D
        L.orc_episode_stats.argtypes = [C.POINTER(OrcEnv), C.POINTER(OrcEpisodeStats)]
        L.orc_noise_words.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.POINTER(C.c_uint32)]
        assert C.sizeof(OrcConfig) > 0
        _lib = L
    return _lib


def default_config(**overrides):
    cfg = OrcConfig()
    lib().orc_default_config(C.byref(cfg))
    assert cfg.struct_size == C.sizeof(OrcConfig), (cfg.struct_size, C.sizeof(OrcConfig))
    apply_overrides(cfg, overrides)
    return cfg


def apply_overrides(cfg, overrides):
    for k, v in overrides.items():
        if k == "grid_size":
            cfg.grid_w, cfg.grid_h = int(v[0]), int(v[1])
        elif k == "grid_choices":
            cfg.num_grid_choices = len(v)
            for i, (w, h) in enumerate(v):
                cfg.grid_choices_w[i], cfg.grid_choices_h[i] = int(w), int(h)
        elif k == "sf_thresholds":
            for i, t in enumerate(v):
                cfg.sf_thresholds[i] = float(t)
        else:
            if not hasattr(cfg, k):
                raise AttributeError(k)
            setattr(cfg, k, v)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleEnv:
    """One oracle environment instance (scalar; mirrors one reference `UAVEnvironment`)."""

    def __init__(self, cfg, env_index=0, pos_x=None, pos_y=None):
        self.L = lib()
        self.cfg = cfg
        self.e = OrcEnv()
        n = cfg.num_sensors
        if pos_x is None:
            pos_x = np.zeros(n, np.float32); pos_y = np.zeros(n, np.float32)
            self.L.orc_noise_positions(cfg.seed, env_index, 0xFFFFFFFF, n, cfg.grid_w, cfg.grid_h, _fp(pos_x), _fp(pos_y))
        px = np.ascontiguousarray(pos_x, np.float32); py = np.ascontiguousarray(pos_y, np.float32)
        self.L.orc_init(C.byref(self.e), C.byref(cfg), env_index, _fp(px), _fp(py))
        self.n = n
        self.obs_dim = self.L.orc_obs_dim(C.byref(cfg), n)

    def reset_tape(self, rt):
        rt = np.ascontiguousarray(rt, np.float32); assert rt.shape in ((3, self.n), (4, self.n))
        if rt.shape[0] == 3:            # no zS row (only DomainRandEnv's fresh sensors read it): zeros
            rt = np.concatenate([rt, np.zeros((1, self.n), np.float32)])
        obs = np.empty(self.obs_dim, np.float32)
        self.L.orc_reset_tape(C.byref(self.e), _fp(rt), _fp(obs))
        return obs

    def step_policy_tape(self, policy, tp):
        """tp float32[7, n] incl. the zP row; returns (action, obs, reward, truncated)."""
        tp = np.ascontiguousarray(tp, np.float32); assert tp.shape == (7, self.n)
        obs = np.empty(self.obs_dim, np.float32)
        r = _D(); tr = C.c_int(); a = C.c_int()
        self.L.orc_step_policy_tape(C.byref(self.e), int(policy), _fp(tp), _fp(obs), C.byref(r), C.byref(tr), C.byref(a))
        return a.value, obs, r.value, bool(tr.value)

    def step_tape(self, action, tp):
        tp = np.ascontiguousarray(np.asarray(tp)[:6], np.float32); assert tp.shape == (6, self.n)
        obs = np.empty(self.obs_dim, np.float32)
        r = _D(); tr = C.c_int()
        rc = self.L.orc_step_tape(C.byref(self.e), int(action), _fp(tp), _fp(obs), C.byref(r), C.byref(tr))
        if rc != 0:
            raise ValueError(f"Invalid action: {action}")
        return obs, r.value, bool(tr.value)

    def reset_keyed(self):
        obs = np.empty(self.obs_dim, np.float32)
        self.L.orc_reset_keyed(C.byref(self.e), _fp(obs))
        return obs

    def step_keyed(self, action):
        obs = np.empty(self.obs_dim, np.float32)
        r = _D(); tr = C.c_int()
        rc = self.L.orc_step_keyed(C.byref(self.e), int(action), _fp(obs), C.byref(r), C.byref(tr))
        if rc != 0:
            raise ValueError(f"Invalid action: {action}")
        return obs, r.value, bool(tr.value)

    def episode_stats(self):
        """dqn.py:316-331 `last_episode_stats` of the episode the current state ends (call before the next reset)."""
        return episode_stats_of(self.e)

    def next_random_action(self):
        return self.L.orc_noise_action(self.cfg.seed, self.e.env_index, self.e.episode, self.e.current_step + 1)

    def state(self):
        e, n = self.e, self.n
        arr = lambda f, dt: np.ctypeslib.as_array(getattr(e, f))[:n].astype(dt)
        avg = arr("avg_rssi", np.float64)
        avg[arr("avg_valid", np.uint8) == 0] = np.nan
        return dict(
            buffer=arr("buffer", np.float64), gen=arr("gen", np.float64), tx=arr("tx", np.float64),
            lost=arr("lost", np.float64), avg_rssi=avg, sf=arr("sf", np.int32),
            visited=arr("visited", np.uint8), data_collected=arr("data_collected", np.uint8),
            pos_x=arr("pos_x", np.float32), pos_y=arr("pos_y", np.float32),
            uav_x=np.float32(e.uav_x), uav_y=np.float32(e.uav_y), battery=np.float64(e.battery),
            step=np.int32(e.current_step), total_reward=np.float64(e.total_reward),
            total_collected=np.float64(e.total_data_collected),
            capture_triggers=np.int32(e.capture_triggers), boundary_hits=np.int32(e.boundary_hits),
            edge_steps=np.int32(e.edge_steps), last_bytes=np.float64(e.last_step_bytes),
            episode=np.uint32(e.episode), grid_w=np.int32(e.grid_w), grid_h=np.int32(e.grid_h),
            start_x=np.float32(e.start_x), start_y=np.float32(e.start_y),
        )


def episode_stats_of(e):
    st = OrcEpisodeStats()
    lib().orc_episode_stats(C.byref(e), C.byref(st))
    d = {k: getattr(st, k) for k, _ in OrcEpisodeStats._fields_}
    d["grid_size"] = (d.pop("grid_w"), d.pop("grid_h"))
    return d


def noise_words(seed, env_index, episode, step, lane, call):
    o = (C.c_uint32 * 4)()
    lib().orc_noise_words(int(seed), env_index & 0xFFFFFFFF, episode & 0xFFFFFFFF, step, lane, call, o)
    return list(o)


def philox(ctr, key, rounds=None):
    """Philox4x32 with `rounds` rounds (default: the noise specification's ORC_PHILOX_ROUNDS)."""
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c, k, lib().orc_philox_rounds() if rounds is None else int(rounds), o)
    return list(o)


def normal_pair(a, b):
    z0, z1 = C.c_float(), C.c_float()
    lib().orc_normal_pair(a, b, C.byref(z0), C.byref(z1))
    return z0.value, z1.value


def run_random_policy(cfg, num_envs, steps, env_index_base=0):
    s = _D()
    n = lib().orc_run_random_policy(C.byref(cfg), num_envs, env_index_base, steps, C.byref(s))
    return n, s.value


def _env_state(e, n):
    arr = lambda f, dt: np.ctypeslib.as_array(getattr(e, f))[:n].astype(dt)
    avg = arr("avg_rssi", np.float64)
    avg[arr("avg_valid", np.uint8) == 0] = np.nan
    return dict(buffer=arr("buffer", np.float64), gen=arr("gen", np.float64), tx=arr("tx", np.float64),
                lost=arr("lost", np.float64), avg_rssi=avg, sf=arr("sf", np.int32),
                visited=arr("visited", np.uint8), data_collected=arr("data_collected", np.uint8),
                pos_x=arr("pos_x", np.float32), pos_y=arr("pos_y", np.float32),
                uav_x=np.float32(e.uav_x), uav_y=np.float32(e.uav_y), battery=np.float64(e.battery),
                step=np.int32(e.current_step), total_reward=np.float64(e.total_reward),
                total_collected=np.float64(e.total_data_collected), episode=np.uint32(e.episode),
                capture_triggers=np.int32(e.capture_triggers), boundary_hits=np.int32(e.boundary_hits),
                edge_steps=np.int32(e.edge_steps), collisions_total=np.int32(e.collisions_total),
                grid_w=np.int32(e.grid_w), grid_h=np.int32(e.grid_h),
                start_x=np.float32(e.start_x), start_y=np.float32(e.start_y))


def trace_keyed(cfg, num_envs, steps, base=0, actions=None, auto_reset=True, policy=0):
    """Run the keyed oracle for a batch; returns dict(obs[steps,E,D], reward, done, term_obs, actions,
    reset_obs[E,D], final=[per-env state dicts])."""
    L = lib()
    D = L.orc_obs_dim(C.byref(cfg), cfg.num_sensors)
    E = num_envs
    obs = np.zeros((steps, E, D), np.float32); term = np.zeros((steps, E, D), np.float32)
    rew = np.zeros((steps, E), np.float64); done = np.zeros((steps, E), np.uint8)
    acts_out = np.zeros((steps, E), np.int32); reset_obs = np.zeros((E, D), np.float32)
    finals = (OrcEnv * E)()
    ap = None
    if actions is not None:
        actions = np.ascontiguousarray(actions, np.int32); assert actions.shape == (steps, E)
        ap = actions.ctypes.data_as(C.c_void_p)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    L.orc_trace_keyed(C.byref(cfg), E, base, steps, ap, int(policy), int(auto_reset), vp(obs), vp(rew), vp(done), vp(term),
                      vp(acts_out), vp(reset_obs), C.cast(finals, C.c_void_p))
    return dict(obs=obs, reward=rew, done=done, term_obs=term, actions=acts_out, reset_obs=reset_obs,
                final=[_env_state(finals[i], cfg.num_sensors) for i in range(E)])
