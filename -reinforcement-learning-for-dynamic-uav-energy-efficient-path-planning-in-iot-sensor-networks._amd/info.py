"""The `info` dict of the reference's `UAVEnvironment._get_info()` (uav_env.py:676-700), built on the host from the
device state: one place for the single-environment view (gym_env.py) and for the episode-end infos of the vectorised
environment (vec_env.py), where the state of the terminal step comes from the kernel's terminal snapshot."""
import numpy as np


def build_info(cfg, rec, buffer, gen, tx, sensors_collected, num_sensors):
    """rec: one UavEnvRecord (numpy structured scalar); buffer / gen / tx: float64 arrays of the env's `num_sensors`
    sensors; sensors_collected: len(env.sensors_visited)."""
    n = int(num_sensors)
    rate = float(cfg.data_generation_rate)
    # uav_env.py:386-394 _get_sensor_urgencies: AoI approximation, stored as float32
    urg = (np.asarray(buffer, np.float64)[:n] / rate).astype(np.float32) if rate > 0 else np.zeros(n, np.float32)
    battery = float(rec["battery"])
    nvis = int(sensors_collected)
    return {
        "uav_position": np.array([rec["uav_x"], rec["uav_y"]], dtype=np.float32),
        "battery": battery,
        "battery_percent": battery / float(cfg.max_battery) * 100,
        "sensors_collected": nvis,
        "current_step": int(rec["current_step"]),
        "total_reward": float(rec["total_reward"]),
        "total_data_collected": float(rec["total_data_collected"]),
        "coverage_percentage": (nvis / n) * 100,
        "is_alive": bool(battery > float(cfg.alive_fraction) * float(cfg.max_battery)),
        "max_urgency": float(np.max(urg)),
        "avg_urgency": float(np.mean(urg)),
        "high_urgency_sensors": int(np.sum(urg > 0.8)),
        "capture_effect_triggers": int(rec["capture_triggers"]),
        "boundary_hits": int(rec["boundary_hits"]),
        "edge_steps": int(rec["edge_steps"]),
        "last_step_bytes_collected": float(rec["last_step_bytes"]),
        "sensor_collection_ratios": [float(t / max(g, 1e-6)) for t, g in zip(np.asarray(tx)[:n], np.asarray(gen)[:n])],
    }
