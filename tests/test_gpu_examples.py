"""GPU: the scripts under examples/ run end to end on small settings and print the JSON they promise (they are what the
numbers in DESIGN.md / profiles/ were made with; a script that has rotted would silently invalidate them)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "examples", script), *args], text=True, stderr=subprocess.DEVNULL,
                                  cwd=ROOT, timeout=240)
    return json.loads([l for l in out.splitlines() if l.strip().startswith("{")][-1])


def test_train_dqn_example_both_update_ratios():
    d = _run("train_dqn.py", "--envs", "256", "--sensors", "20", "--timesteps", "40000")
    assert d["timesteps"] >= 40000 and d["graph_replay"] is True and d["gradient_steps"] >= 10     # (learning_starts 25 000)
    assert abs(d["updates_per_transition"] - 1.0 / (4 * 256)) < 1e-12                    # SB3's gradient_steps = 1
    d = _run("train_dqn.py", "--envs", "256", "--sensors", "20", "--timesteps", "40000", "--updates-per-transition", "0.0625",
             "--reward-scale", "0.001")
    assert d["updates_per_transition"] == 0.0625 and d["gradient_steps"] >= 0.9 * (40000 - 25000) / 16   # one update per 16 transitions


def test_train_dqn_example_saves_and_resumes(tmp_path):
    ck = str(tmp_path / "dqn.pt")
    a = _run("train_dqn.py", "--envs", "256", "--sensors", "20", "--timesteps", "30000", "--save", ck)
    assert os.path.exists(ck) and a["timesteps"] >= 30000
    b = _run("train_dqn.py", "--envs", "256", "--sensors", "20", "--timesteps", "45000", "--load", ck)
    # resumed at a's counters: reaches the new total in (45000 - 30000) / 256 more vector steps, and keeps a's update count
    assert b["timesteps"] >= 45000 and b["vector_steps"] == -(-45000 // 256) and b["gradient_steps"] > a["gradient_steps"]


def test_learn_config3_example_reports_all_four_policies():
    d = _run("learn_config3.py", "--envs", "64", "--eval-envs", "64", "--timesteps", "40000", "--no-tune")
    assert d["train"]["updates_per_transition"] == 0.0625 and d["train"]["updates"] > 800
    for pol in ("dqn_greedy", "uniform_random", "max_throughput_greedy_v2", "nearest_sensor_greedy"):
        assert d[pol]["episodes"] == 64 and 0.0 <= d[pol]["ndr"] <= 100.0 and 0.0 < d[pol]["jains"] <= 1.0
        assert 1381 <= d[pol]["mean_episode_length"] <= 2100
    assert d["max_throughput_greedy_v2"]["ndr"] > d["uniform_random"]["ndr"]              # the curriculum gate's heuristic beats random


def test_greedy_benchmark_example():
    d = _run("greedy_benchmark.py", "--envs", "128")
    assert d["episodes"] == 128 and 40.0 < d["ndr"] <= 100.0 and 0.2 < d["jains"] <= 1.0
