"""GPU box: full-length episodes against the keyed oracle, step by step (the suite's keyed cases stop after <= 260 steps; a
default-constants episode ends by battery after 1 381-1 933 steps and its float64 totals have accumulated thousands of
additions by then).  Reuses tests/test_gpu_parity.py:test_keyed_rollout_matches_oracle on longer cases."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T

CASES = [
    ("soak_n50_default_full", 32, dict(num_sensors=50), 4300, 0),
    ("soak_n20_default_full", 32, dict(num_sensors=20), 4300, 0),
    ("soak_n10_default_full", 32, dict(num_sensors=10), 4300, 0),
    ("soak_domain_rand_full", 32, dict(num_sensors=20, pad_sensors=50, grid_choices=[(100, 100), (300, 300), (500, 500)]), 4300, 1 | 2 | 4 | 8),
    ("soak_n50_duty100_small_grid", 48, dict(num_sensors=50, duty_cycle=100.0, grid_size=(60, 60)), 3000, 0),
]
for c in CASES:
    t0 = time.perf_counter()
    T.test_keyed_rollout_matches_oracle(c)
    print("%-30s %d envs x %d steps: identical actions / done, obs <= %g, rewards <= %g rel, final state equal   (%.1f s)" % (
        c[0], c[1], c[3], T.OBS_ATOL, T.REW_RTOL, time.perf_counter() - t0), flush=True)
