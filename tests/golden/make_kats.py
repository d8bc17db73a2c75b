"""Build-container-only: capture scalar known-answer values from the REAL reference classes
(IoTSensor.calculate_rssi at sigma=0, RewardFunction, UAV energy arithmetic) into kats.json."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as R  # noqa: E402

uav_env, iot = R.import_reference()
from uav import UAV  # noqa: E402
from rewards.reward_function import RewardFunction  # noqa: E402

out = {}
s = iot.IoTSensor(position=(0.0, 0.0), shadowing_std_db=0.0)
out["rssi_sigma0"] = [dict(grid_dist=d, rssi=s.calculate_rssi((float(d), 0.0)))
                      for d in (0, 1, 10, 11, 45, 115, 150, 182, 183, 211, 212, 300, 500, 1400)]
rf = RewardFunction(penalty_data_loss=-1.0, reward_urgency_reduction=20.0, penalty_battery=-0.5, reward_movement=10.0)
out["move_reward_ok"] = rf.calculate_movement_reward(True, 274.0 - (274.0 - 500.0 / 3600))
out["move_reward_blocked"] = rf.calculate_movement_reward(False, 274.0 - (274.0 - 250.0 / 3600))
out["collect_reward_nothing_500_250_1000"] = rf.calculate_collection_reward(
    bytes_collected=0.0, was_new_sensor=False, was_empty=False, all_sensors_collected=False,
    battery_used=700.0 / 3600, collision_count=0, data_loss=0.0, urgency_reduced=0.0,
    sensor_buffers=[500.0, 250.0, 1000.0], sensor_urgency=0.0)
out["collect_reward_full"] = rf.calculate_collection_reward(
    bytes_collected=683.75, was_new_sensor=True, was_empty=True, all_sensors_collected=False,
    battery_used=700.0 / 3600, collision_count=3, data_loss=12.5, urgency_reduced=310.79544,
    sensor_buffers=[316.25, 0.0, 1000.0, 730.0], sensor_urgency=0.61)
u = UAV(start_position=(0.0, 0.0), max_battery=274.0)
trace = []
for d in ("UP", "LEFT", "RIGHT", "DOWN", "DOWN"):
    ok = u.move(d, (500, 500))
    trace.append(dict(dir=d, ok=bool(ok), x=float(u.position[0]), y=float(u.position[1]), battery=u.battery))
u.hover(1.0)
trace.append(dict(dir="HOVER", ok=True, x=float(u.position[0]), y=float(u.position[1]), battery=u.battery))
out["uav_trace"] = trace
json.dump(out, open(os.path.join(HERE, "kats.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
