#!/bin/bash
# GPU box: SQ counters for the step kernel -> dynamic instruction mix and issue / wait cycle split per wave.
# Usage: tools/pmc_sq.sh <tag> [envs ...]      (default envs: 256 4096; one rocprofv3 --pmc pass per counter set and size)
set -e
TAG=${1:-sq}; shift || true
SIZES=${@:-256 4096}
export TMPDIR=/tmp
mkdir -p gpurun_out/profiles
OUT=gpurun_out/profiles/${TAG}_sq_counters.txt
echo "# SQ counters of uav_step_kernel, per wave (tools/pmc_sq.sh; tools/step_driver.py <envs> 300; commit $(cat .build_commit 2>/dev/null))" > $OUT
echo "# SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4 = shader cycles), MI355X_MICROARCH.md" >> $OUT
for E in $SIZES; do
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_WAVES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM"; do
  D=gpurun_out/prof_sq_${E}_$(echo $SET | md5sum | cut -c1-6)
  rm -rf $D
  rocprofv3 --pmc $SET --output-format csv -d $D -- python3 tools/step_driver.py $E 300 > /dev/null 2> gpurun_out/profiles/${TAG}_pmc.err || { echo "pass failed: $SET"; tail -3 gpurun_out/profiles/${TAG}_pmc.err; continue; }
  python3 - "$D" "$E" >> $OUT <<'PY'
import csv, glob, sys, collections
d, E = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "uav_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(acc["SQ_WAVES"]) / max(1, len(acc["SQ_WAVES"]))
for k, v in sorted(acc.items()):
    print(f"step envs={E:>6s} {k:28s} per wave {sum(v)/len(v)/w:12.1f}   launches {len(v)}   waves/launch {w:.0f}")
PY
done
done
cat $OUT
