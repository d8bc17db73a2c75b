#!/bin/bash
# GPU box: dynamic instruction counts of the step and rollout kernels (SQ counters), per wave.
set -e
export TMPDIR=/tmp
D=gpurun_out/prof_sq_rollout
rm -rf $D
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $D -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-graph > /dev/null 2> gpurun_out/pmc_rollout.err || { tail -5 gpurun_out/pmc_rollout.err; exit 1; }
python3 - "$D" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = "rollout" if "uav_rollout_kernel" in r["Kernel_Name"] else ("step" if "uav_step_kernel" in r["Kernel_Name"] else None)
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    w = sum(acc[k]["SQ_WAVES"]) / len(acc[k]["SQ_WAVES"])
    for cn, v in sorted(acc[k].items()):
        print(f"{k:8s} {cn:20s} per wave {sum(v)/len(v)/w:10.1f}" + (f"   per step {sum(v)/len(v)/w/16:8.1f}" if k == "rollout" else ""))
PY
