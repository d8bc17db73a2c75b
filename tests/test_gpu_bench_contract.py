"""GPU: bench.py prints ONE JSON line with the fields the driver's contract names (short run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "96", "--warmup", "16",
                                   "--fused", "4", "--no-cpu-baseline"], text=True, stderr=subprocess.DEVNULL, cwd=ROOT)
    lines = [l for l in out.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == baseline["metric"] and d["unit"] == "env-steps/s"
    assert d["n_gpus"] == 1 and d["steps"] == 96 and d["warmup"] == 16 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "f64" and d["data"] == "synthetic"
    # the reference publishes no throughput; the ratio is to its own step() as measured for BASELINE.md section 2 (764 env-steps/s)
    assert abs(d["vs_baseline"] - d["value"] / 764.0) < 1e-6 * d["vs_baseline"] and "BASELINE.md section 2" in d["vs_baseline_basis"]
    assert abs(d["ms_per_step"] * 1e-3 * d["value"] - 4096) < 1e-3 * 4096          # value = envs / time per step
    assert d["value"] > 2e6                                                       # north_star target on one GPU
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["algorithmic_bytes_per_env_step"] == 68 * 50 + 104
    assert "cpu_baseline" in d and d["fused_rollout"]["steps_per_launch"] == 4


def test_bench_two_rank_path_rehearsed_on_one_gpu():
    """The N > 1 path of bench.py exactly as the driver launches it (`python -m torch.distributed.run --nproc-per-node N ... bench.py
    --gpus N`), rehearsed with two ranks on this one GPU over gloo (UAVENV_BENCH_REHEARSE=gloo: RCCL needs a GPU per rank): sharded
    environments, the chunk all-gather into the shared ring, barrier + max-over-ranks timing, ONE JSON line from rank 0 whose value
    counts the environment steps of BOTH ranks.  No scaling number is claimed from it (both ranks share the GPU)."""
    env = dict(os.environ, UAVENV_BENCH_REHEARSE="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "64", "--warmup", "16", "--fused", "0"]
    r = subprocess.run(cmd, text=True, capture_output=True, cwd=ROOT, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 64 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert abs(d["ms_per_step"] * 1e-3 * d["value"] - 2 * 4096) < 1e-3 * 2 * 4096     # whole-job aggregate: both shards' environments
    assert "all_gather" in d["config"]["exchange"] and "exchange_error" not in d["config"]
    assert "REHEARSAL" in d["config"]["parallelism"]
    assert d["shard_only"]["env_steps_per_s"] > 0
