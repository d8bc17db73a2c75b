#!/bin/bash
# GPU box: everything profiles/<tag>_* is made of, in one call (about 3 minutes): the rocprofv3 kernel stats + HBM traffic of the
# bench command, the SQ counters of the step kernel, the bench line at the default 2000 steps and at the driver's 20, and the
# batch-size sweep.  Usage (build container): tools/gpu.sh --timeout 900 'tools/regen_profiles.sh r02n'; then copy
# gpurun_out/profiles/<tag>_* into profiles/.
set -e
TAG=${1:?tag}
mkdir -p gpurun_out/profiles
P=gpurun_out/profiles
timeout -k 10 300 tools/profile_bench.sh $TAG > gpurun_out/pb.log 2>&1
timeout -k 10 200 tools/pmc_sq.sh $TAG > gpurun_out/sq.log 2>&1
python3 bench.py > $P/${TAG}_bench.json 2> gpurun_out/bench.err
python3 bench.py --steps 20 --warmup 5 > $P/${TAG}_bench_steps20.json 2>> gpurun_out/bench.err
python3 tools/scale_envs.py > $P/${TAG}_scale_envs.json
python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
for f in ("bench", "bench_steps20", "bench_under_rocprof"):
    d = json.load(open("gpurun_out/profiles/%s_%s.json" % (tag, f))); r = d["roofline"]
    print("%-22s %.1f M env-steps/s  %.2f us/step  launch %.2f us (back-to-back %.2f)  frac %.4f  fused %s" % (
        f, d["value"] / 1e6, d["ms_per_step"] * 1e3, r["avg_launch_ms"] * 1e3, r.get("back_to_back_ms", 0) * 1e3, r["frac"],
        d.get("fused_rollout", {}).get("env_steps_per_s")))
PY
head -4 $P/${TAG}_kernel_stats.md
