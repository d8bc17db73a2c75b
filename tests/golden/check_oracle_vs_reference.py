"""Build-container-only: step the REAL reference and the C oracle side by side under the same
noise tape and report every divergence.  Usage: python tests/golden/check_oracle_vs_reference.py

Not collected by pytest (no test_ prefix) because /root/reference does not exist on the GPU box.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness as R  # noqa: E402
import tape as T  # noqa: E402
from oracle import oracle as O  # noqa: E402

STATE_EXACT = ["sf", "visited", "data_collected", "uav_x", "uav_y", "step", "capture_triggers",
               "boundary_hits", "edge_steps"]
STATE_F64 = ["buffer", "gen", "tx", "lost", "battery", "total_reward", "total_collected", "last_bytes"]


def compare_states(a, b, where, rtol=1e-12):
    bad = []
    for k in STATE_EXACT:
        if not np.array_equal(a[k], b[k]):
            bad.append((where, k, a[k], b[k]))
    for k in STATE_F64:
        if not np.allclose(a[k], b[k], rtol=rtol, atol=1e-9):
            bad.append((where, k, a[k], b[k]))
    # avg_rssi carries the 1-ulp float32 log10 fuzz of the reference platform (see oracle header)
    av, bv = a["avg_rssi"], b["avg_rssi"]
    if not (np.array_equal(np.isnan(av), np.isnan(bv)) and np.allclose(av, bv, rtol=0, atol=4e-5, equal_nan=True)):
        bad.append((where, "avg_rssi", av, bv))
    return bad


def oracle_config(n, grid, kw):
    """Map the reference's constructor kwargs (uav_env.py:266-287) onto the oracle config."""
    kw = dict(kw)
    over = {}
    if "sensor_duty_cycle" in kw:
        over["duty_cycle"] = kw.pop("sensor_duty_cycle")
    if "uav_start_position" in kw:
        over["start_x"], over["start_y"] = kw.pop("uav_start_position")
    over.update(kw)
    return O.default_config(grid_size=grid, num_sensors=n, **over)


def run_case(n, grid, tape_seed, steps, p_collect=0.2, sigma=None, verbose=False, **kw):
    """Returns (num_divergent_steps, max |obs diff|, max rel reward diff)."""
    cfg = oracle_config(n, grid, kw)
    if sigma is not None:
        cfg.shadowing_std_db = sigma
    acts = T.actions(tape_seed, 0, steps, p_collect)
    diverged = 0
    max_obs = 0.0
    max_rew = 0.0
    with R.TapedReference(n, grid, tape_seed, sigma=sigma, **kw) as ref:
        orc = O.OracleEnv(cfg, 0, ref.pos_x, ref.pos_y)
        episode = 0
        ro, _ = ref.reset()
        oo = orc.reset_tape(T.reset_tape(tape_seed, 0, episode, n))
        assert np.array_equal(ro, oo), ("reset obs", ro, oo)
        for s, a in enumerate(acts):
            tp = T.step_tape(tape_seed, 0, s, n)
            ro, rr, _, rtr, _ = ref.step(a)
            oo, orr, otr = orc.step_tape(a, tp)
            bad = compare_states(ref.state(), orc.state(), s)
            dobs = float(np.max(np.abs(ro.astype(np.float64) - oo.astype(np.float64))))
            drew = abs(rr - orr) / max(1.0, abs(rr))
            max_obs, max_rew = max(max_obs, dobs), max(max_rew, drew)
            if bad or dobs > 0 or drew > 1e-12 or rtr != otr:
                diverged += 1
                if verbose:
                    print(f"  step {s} action {a}: obs diff {dobs:.3e} reward {rr!r} vs {orr!r} trunc {rtr}/{otr}")
                    for b in bad[:4]:
                        print("    ", b[1], b[2], b[3])
                return diverged, max_obs, max_rew, s
            if rtr:
                episode += 1
                ro, _ = ref.reset()
                oo = orc.reset_tape(T.reset_tape(tape_seed, 0, episode, n))
                assert np.array_equal(ro, oo)
    return diverged, max_obs, max_rew, steps


STRESS = [
    dict(n=20, grid=(100, 100), steps=300, p_collect=0.6, kw=dict(sensor_duty_cycle=60.0)),
    dict(n=50, grid=(100, 100), steps=300, p_collect=0.6, kw=dict(sensor_duty_cycle=100.0)),
    dict(n=20, grid=(100, 100), steps=300, p_collect=0.5, kw=dict(max_steps=40, sensor_duty_cycle=50.0)),
    dict(n=10, grid=(60, 60), steps=300, p_collect=0.7, kw=dict(max_battery=6.0, sensor_duty_cycle=80.0)),
    dict(n=20, grid=(100, 100), steps=200, p_collect=0.5, sigma=0.0, kw=dict(sensor_duty_cycle=70.0)),
    dict(n=20, grid=(100, 100), steps=200, p_collect=0.5, kw=dict(collection_duration=2.5, sensor_duty_cycle=70.0,
                                                                 include_sensor_positions=True)),
    dict(n=3, grid=(30, 30), steps=600, p_collect=0.8, kw=dict(sensor_duty_cycle=100.0, max_buffer_size=400.0,
                                                               uav_start_position=(15.0, 15.0))),
    dict(n=1, grid=(20, 20), steps=300, p_collect=0.8, kw=dict(sensor_duty_cycle=100.0, uav_start_position=(5.0, 5.0))),
    dict(n=50, grid=(150, 150), steps=300, p_collect=0.5, kw=dict(sensor_duty_cycle=100.0, penalty_data_loss=-3.0,
                                                                 reward_urgency_reduction=7.0, penalty_battery=-1.5,
                                                                 reward_movement=4.0, data_generation_rate=9.0,
                                                                 rssi_threshold=-80.0)),
]

if __name__ == "__main__":
    total = 0
    bad = 0
    for i, c in enumerate(STRESS):
        for seed in range(3):
            d, mo, mr, at = run_case(c["n"], c["grid"], 77000 + 10 * i + seed, c["steps"], c["p_collect"],
                                     sigma=c.get("sigma"), verbose=True, **c["kw"])
            total += 1
            bad += d
            print(f"stress {i} seed={seed}: divergent={d} (at step {at}) max_obs={mo:.2e} max_rel_rew={mr:.2e}")
    if "--quick" in sys.argv:
        print(f"{bad} of {total} runs diverged")
        sys.exit(0)
    for n in (10, 20, 50):
        for grid in ((250, 250), (500, 500), (1000, 1000)):
            for seed in range(1, 5):
                d, mo, mr, at = run_case(n, grid, 1000 * n + grid[0] + seed, 400, verbose=True)
                total += 1
                bad += d
                print(f"N={n} grid={grid} seed={seed}: divergent={d} (at step {at}) max_obs={mo:.2e} max_rel_rew={mr:.2e}")
    print(f"{bad} of {total} runs diverged")
