#!/bin/bash
# GPU box: average memory-instruction latencies of the step kernel from the SQ "level" counters
# (SQ_INST_LEVEL_x accumulates the number of outstanding x instructions per cycle: level / count = cycles in flight each).
export TMPDIR=/tmp
for E in ${@:-256 4096}; do
  D=gpurun_out/prof_lat_$E; rm -rf $D
  rocprofv3 --pmc SQ_WAVES SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR --output-format csv -d $D -- python3 tools/step_driver.py $E 300 > /dev/null 2>&1
  python3 - "$D" "$E" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "uav_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m["SQ_WAVES"]
print(f"envs={sys.argv[2]}: per wave: SMEM {m['SQ_INSTS_SMEM']/w:.1f} insts, level {m['SQ_INST_LEVEL_SMEM']/w:.0f} -> {m['SQ_INST_LEVEL_SMEM']/m['SQ_INSTS_SMEM']:.0f} per inst;"
      f"  VMEM rd {m['SQ_INSTS_VMEM_RD']/w:.1f} wr {m['SQ_INSTS_VMEM_WR']/w:.1f}, level {m['SQ_INST_LEVEL_VMEM']/w:.0f} -> {m['SQ_INST_LEVEL_VMEM']/(m['SQ_INSTS_VMEM_RD']+m['SQ_INSTS_VMEM_WR']):.0f} per inst;"
      f"  cycles_vmem_rd {m['SQ_INST_CYCLES_VMEM_RD']/w:.0f} wr {m['SQ_INST_CYCLES_VMEM_WR']/w:.0f}")
PY
done
