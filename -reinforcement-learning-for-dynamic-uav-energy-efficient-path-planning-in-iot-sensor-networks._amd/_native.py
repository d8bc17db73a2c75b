"""ctypes binding of the C ABI declared in include/uavenv.h (libuavenv_hip.so).

There is NO fallback: if the shared library is missing or fails to load, importing the product
raises.  (The CPU oracle under oracle/ is test infrastructure and is never reachable from here.)
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libuavenv_hip.so")

_D = C.c_double

FLAG_RANDOM_LAYOUT, FLAG_FAR_START, FLAG_PROX_SHAPING, FLAG_JAIN_BONUS, FLAG_AUTO_RESET = 1, 2, 4, 8, 16

(F_POS_X, F_POS_Y, F_BUFFER, F_GEN, F_TX, F_LOST, F_AVG_RSSI, F_FLAGS, F_RECORD, F_EPISODE_STATS, F_TERM_RECORD, F_TERM_SENSORS) = range(12)

E_INVALID, E_HIP, E_ACTION, E_ALLOC = -1, -2, -3, -4
POLICY_ACTIONS, POLICY_RANDOM, POLICY_NEAREST, POLICY_MAX_THROUGHPUT_V2 = 0, 1, 2, 3
GEMM_A_RELU, GEMM_A_MASK, GEMM_B_RELU, GEMM_BIAS, GEMM_ROWSUM, GEMM_SUMSQ = 1, 2, 4, 8, 16, 32


def gemm_sumsq_count(M, N):
    """include/uavenv.h:UAVENV_GEMM_SUMSQ_COUNT"""
    return ((M + 15) // 16) * ((N + 31) // 32)


UPD_LOSS, UPD_NORM2, UPD_STEP, UPD_BC1, UPD_BC2, UPD_LR, UPD_COUNT, UPD_WORKSPACE = 0, 1, 2, 3, 4, 5, 8, 256
ABI_VERSION = 2


class UavEnvConfig(C.Structure):
    """include/uavenv.h: UavEnvConfig (field for field)."""
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


# numpy dtypes of the two HBM record types (include/uavenv.h)
def record_dtype():
    import numpy as np
    return np.dtype([
        ("battery", "<f8"), ("total_reward", "<f8"), ("total_data_collected", "<f8"), ("last_step_bytes", "<f8"),
        ("prev_dist_nearest", "<f8"), ("episode_return", "<f8"),
        ("uav_x", "<f4"), ("uav_y", "<f4"), ("start_x", "<f4"), ("start_y", "<f4"),
        ("current_step", "<i4"), ("episode", "<u4"), ("capture_triggers", "<i4"), ("boundary_hits", "<i4"),
        ("edge_steps", "<i4"), ("collisions_total", "<i4"), ("first_full_coverage_step", "<i4"),
        ("grid_w", "<i4"), ("grid_h", "<i4"), ("num_sensors", "<i4"), ("env_index", "<u4"), ("status", "<u4"),
        ("inv_grid_w", "<f8"), ("inv_grid_h", "<f8"),
    ])


def episode_stats_dtype():
    import numpy as np
    return np.dtype([
        ("episode_return", "<f8"), ("total_reward", "<f8"), ("total_generated", "<f8"), ("total_collected", "<f8"),
        ("total_lost", "<f8"), ("battery_remaining", "<f8"), ("jains_index", "<f8"), ("fairness_std", "<f8"),
        ("length", "<i4"), ("sensors_visited", "<i4"), ("num_sensors", "<i4"), ("grid_w", "<i4"), ("grid_h", "<i4"),
        ("first_full_coverage_step", "<i4"), ("episode", "<u4"), ("valid", "<u4"),
    ])


class UavRingLayout(C.Structure):
    """include/uavenv.h:UavRingLayout (the transition ring of replay.py, for uavenv_ring_gather_stacked)."""
    _fields_ = [("section", C.c_int64), ("num_chunks", C.c_int32), ("world", C.c_int32), ("slots_per_chunk", C.c_int32),
                ("envs", C.c_int32), ("obs_dim", C.c_int32), ("terminal_rows", C.c_int32), ("block", C.c_int32),
                ("obs_floats", C.c_int32), ("term_off", C.c_int32), ("count_off", C.c_int32)]


class UavGemm(C.Structure):
    """include/uavenv.h:UavGemm (one product of uavenv_gemm_f32)."""
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("a_mask", C.c_void_p), ("row_sum", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("flags", C.c_int32),
                ("a_sm", C.c_int64), ("a_sk", C.c_int64), ("b_sk", C.c_int64), ("b_sn", C.c_int64), ("ldc", C.c_int64), ("sumsq", C.c_void_p)]


EXPORTS = [
    "uavenv_abi_version", "uavenv_default_config", "uavenv_obs_dim", "uavenv_create", "uavenv_destroy",
    "uavenv_last_error", "uavenv_num_envs", "uavenv_lane_stride", "uavenv_env_obs_dim", "uavenv_set_env_params",
    "uavenv_set_positions", "uavenv_set_seed", "uavenv_get_config", "uavenv_set_config", "uavenv_set_grid_choices", "uavenv_set_noise_tape",
    "uavenv_dump_noise", "uavenv_reset", "uavenv_step", "uavenv_step_random", "uavenv_step_random_n", "uavenv_step_policy", "uavenv_rollout", "uavenv_frame_stack", "uavenv_ring_gather_stacked", "uavenv_ring_sample_stacked", "uavenv_set_terminal_pool", "uavenv_set_aux_output", "uavenv_enable_terminal_snapshot", "uavenv_attention_weight_floats", "uavenv_attention_features", "uavenv_gemm_f32", "uavenv_td_loss", "uavenv_clip_adam", "uavenv_epsilon_greedy", "uavenv_q_head_select", "uavenv_attn_core_forward", "uavenv_attn_core_backward", "uavenv_get_state",
    "uavenv_set_state", "uavenv_state_bytes", "uavenv_reset_host", "uavenv_step_host", "uavenv_time_steps",
]

_lib = None


class UavEnvError(RuntimeError):
    pass


def lib():
    """Load libuavenv_hip.so (building it first when hipcc is available and it is stale)."""
    global _lib
    if _lib is not None:
        return _lib
    override = os.environ.get("UAVENV_LIB")          # dev tools (tools/exp.sh): load an experimental build of the library
    if override:
        return _load(override)
    from . import build as _b
    try:
        _b.hipcc()
        have_compiler = True
    except RuntimeError:
        have_compiler = False
    if have_compiler:
        _b.build()          # no-op unless a source / header is newer than the library (a stale binary would load silently)
    elif not os.path.exists(LIB_PATH):
        raise UavEnvError("libuavenv_hip.so is missing and hipcc is not available to build it")
    return _load(LIB_PATH)


def _load(path):
    global _lib
    L = C.CDLL(path)       # raises OSError loudly if missing / unloadable: no fallback path exists
    vp, i32, u32, u64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64
    cfgp = C.POINTER(UavEnvConfig)
    sig = {
        "uavenv_abi_version": (C.c_int, []),
        "uavenv_default_config": (C.c_int, [cfgp]),
        "uavenv_obs_dim": (C.c_int, [cfgp]),
        "uavenv_create": (C.c_int, [cfgp, i32, u32, i32, C.POINTER(vp)]),
        "uavenv_destroy": (C.c_int, [vp]),
        "uavenv_last_error": (C.c_char_p, [vp]),
        "uavenv_num_envs": (C.c_int, [vp]),
        "uavenv_lane_stride": (C.c_int, [vp]),
        "uavenv_env_obs_dim": (C.c_int, [vp]),
        "uavenv_set_env_params": (C.c_int, [vp, vp, vp, vp]),
        "uavenv_set_positions": (C.c_int, [vp, vp, vp]),
        "uavenv_set_seed": (C.c_int, [vp, u64]),
        "uavenv_get_config": (C.c_int, [vp, cfgp]),
        "uavenv_set_config": (C.c_int, [vp, cfgp]),
        "uavenv_set_grid_choices": (C.c_int, [vp, i32, vp, vp]),
        "uavenv_set_noise_tape": (C.c_int, [vp, vp, vp]),
        "uavenv_dump_noise": (C.c_int, [vp, vp, vp, vp]),
        "uavenv_reset": (C.c_int, [vp, vp, vp, vp]),
        "uavenv_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
        "uavenv_step_random": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
        "uavenv_step_random_n": (C.c_int, [vp, i32, vp, C.c_int64, vp, C.c_int64, vp, vp, vp]),
        "uavenv_step_policy": (C.c_int, [vp, i32, vp, vp, vp, vp, vp, vp, vp]),
        "uavenv_rollout": (C.c_int, [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
        "uavenv_set_terminal_pool": (C.c_int, [vp, vp, i32, vp, vp]),
        "uavenv_set_aux_output": (C.c_int, [vp, vp, i32]),
        "uavenv_enable_terminal_snapshot": (C.c_int, [vp, i32]),
        "uavenv_attention_weight_floats": (C.c_int, [i32]),
        "uavenv_attention_features": (C.c_int, [vp, vp, vp, i32, i32, vp]),
        "uavenv_gemm_f32": (C.c_int, [C.POINTER(UavGemm), C.POINTER(UavGemm), vp]),
        "uavenv_td_loss": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp, vp]),
        "uavenv_q_head_select": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, C.c_uint64, C.c_int32, vp, vp, vp]),
        "uavenv_attn_core_forward": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]),
        "uavenv_attn_core_backward": (C.c_int, [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]),
        "uavenv_clip_adam": (C.c_int, [vp, vp, vp, vp, C.c_int64, vp, vp, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
        "uavenv_epsilon_greedy": (C.c_int, [vp, i32, i32, vp, vp, u64, i32, vp, vp]),
        "uavenv_frame_stack": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
        "uavenv_ring_gather_stacked": (C.c_int, [vp, C.POINTER(UavRingLayout), vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
        "uavenv_ring_sample_stacked": (C.c_int, [vp, C.POINTER(UavRingLayout), vp, vp, u64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
        "uavenv_get_state": (C.c_int, [vp, i32, vp, C.c_size_t, i32, vp]),
        "uavenv_set_state": (C.c_int, [vp, i32, vp, C.c_size_t, i32, vp]),
        "uavenv_state_bytes": (C.c_size_t, [vp, i32]),
        "uavenv_reset_host": (C.c_int, [vp, vp, vp]),
        "uavenv_step_host": (C.c_int, [vp, vp, vp, vp, vp, vp]),
        "uavenv_time_steps": (C.c_int, [vp, i32, vp, vp, vp, vp, C.POINTER(C.c_float)]),
    }
    for name in EXPORTS:
        fn = getattr(L, name)     # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = sig[name]
    if L.uavenv_abi_version() != ABI_VERSION:
        raise UavEnvError("libuavenv_hip.so ABI version mismatch")
    _lib = L
    return L


def default_config(**overrides):
    cfg = UavEnvConfig()
    rc = lib().uavenv_default_config(C.byref(cfg))
    if rc != 0 or cfg.struct_size != C.sizeof(UavEnvConfig):
        raise UavEnvError("UavEnvConfig layout mismatch between _native.py and include/uavenv.h")
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
        elif not hasattr(cfg, k):
            raise TypeError(f"unknown UavEnvConfig field {k!r}")
        else:
            setattr(cfg, k, v)
    return cfg


def check(rc, handle=None):
    if rc == 0:
        return
    msg = lib().uavenv_last_error(handle)
    msg = msg.decode() if msg else ""
    if rc == E_ACTION:
        raise ValueError(msg or "Invalid action")
    raise UavEnvError(f"uavenv error {rc}: {msg}")
