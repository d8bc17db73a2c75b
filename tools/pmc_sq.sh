#!/bin/bash
# GPU box: SQ counters for the step kernel -> dynamic instruction mix per wave.  Usage: tools/pmc_sq.sh <tag>
set -e
TAG=${1:-sq}
export TMPDIR=/tmp
mkdir -p gpurun_out/profiles
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_FLAT"; do
  D=gpurun_out/prof_sq_$(echo $SET | md5sum | cut -c1-6)
  rm -rf $D
  rocprofv3 --pmc $SET --output-format csv -d $D -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2> gpurun_out/profiles/${TAG}_pmc.err || { tail -5 gpurun_out/profiles/${TAG}_pmc.err; continue; }
  python3 - "$D" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(list)
for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "uav_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:36s} avg/launch {sum(v)/len(v):16.1f}   per-wave(4096) {sum(v)/len(v)/4096:12.2f}   n={len(v)}")
PY
done
