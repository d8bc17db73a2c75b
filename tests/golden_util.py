"""Helpers shared by the CPU (oracle) and GPU (HIP) replays of the committed golden fixtures."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixture_names():
    """Environment fixtures (make_golden.py)."""
    return sorted(n for n in (os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
                  if not n.startswith(("policy_", "domainrand_")))


def policy_fixture_names():
    """Heuristic-policy fixtures (make_golden_policies.py)."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "policy_*.npz")))


def domainrand_fixture_names():
    """Fixtures recorded from the REAL DomainRandEnv (make_golden_domainrand.py): keyed noise, no tape."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "domainrand_*.npz")))


FLAGS_DOMAIN_RAND = 1 | 2 | 4 | 8          # RANDOM_LAYOUT | FAR_START | PROX_SHAPING | JAIN_BONUS


def domainrand_overrides(meta):
    """DomainRandEnv(fixed_num_sensors, curriculum_stage, base_config) (dqn.py:204-235) -> config field names shared by
    OrcConfig and UavEnvConfig."""
    kw = dict(meta["base"])
    grids = [tuple(g) for g in meta["grids"]]
    over = dict(num_sensors=meta["n"], grid_size=grids[0], grid_choices=grids, pad_sensors=50, flags=FLAGS_DOMAIN_RAND,
                seed=meta["seed"])
    if "sensor_duty_cycle" in kw:
        over["duty_cycle"] = kw.pop("sensor_duty_cycle")
    kw.pop("path_loss_exponent", None)      # carried by IoTSensor, read by nothing on the path
    kw.pop("render_mode", None)
    if "include_sensor_positions" in kw:
        kw["include_sensor_positions"] = int(bool(kw["include_sensor_positions"]))
    over.update(kw)
    return over


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


def config_overrides(meta):
    """Reference constructor kwargs (uav_env.py:266-287) -> config field names shared by the oracle's
    OrcConfig and the product's UavEnvConfig."""
    kw = dict(meta["kwargs"])
    over = dict(grid_size=tuple(meta["grid"]), num_sensors=meta["n"])
    if "sensor_duty_cycle" in kw:
        over["duty_cycle"] = kw.pop("sensor_duty_cycle")
    if "uav_start_position" in kw:
        over["start_x"], over["start_y"] = kw.pop("uav_start_position")
    over.update(kw)
    if meta["sigma"] is not None:
        over["shadowing_std_db"] = meta["sigma"]
    return over
