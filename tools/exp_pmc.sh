#!/bin/bash
# GPU box: dynamic instruction counts per wave of the step kernel for tools/_exp/lib_<name>.so variants (built by tools/exp.sh).
#   tools/exp_pmc.sh <envs> name1 name2 ...
export TMPDIR=/tmp
E=$1; shift
for n in "$@"; do
  D=gpurun_out/prof_exp_$n; rm -rf $D
  UAVENV_LIB=$PWD/tools/_exp/lib_$n.so rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $D -- python3 tools/step_driver.py $E 200 > /dev/null 2>&1
  python3 - "$D" "$n" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "uav_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(acc["SQ_WAVES"]) / max(1, len(acc["SQ_WAVES"]))
print("%-10s" % sys.argv[2], "  ".join(f"{k[9:]}={sum(v)/len(v)/w:7.1f}" for k, v in sorted(acc.items()) if k != "SQ_WAVES"))
PY
done
