#!/bin/bash
# GPU box: dynamic VALU mix per wave of the step kernel for tools/_exp/lib_<name>.so variants (built by tools/exp.sh):
# total, int32, int64, conversions, float64 and float32 arithmetic, transcendental -- and what is left ("other": moves,
# selects, compares, DPP, lane reads, bit operations not counted as int32).
#   EXP_CMD="tools/exp_pmc_mix.sh 4096" tools/exp.sh full= exit3=-DUAV_ABL_EXIT=3 ...
export TMPDIR=/tmp
E=$1; shift
for n in "$@"; do
  i=0
  for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" \
             "SQ_WAVES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM"; do
    i=$((i+1)); D=gpurun_out/prof_mix_${n}_$i; rm -rf $D
    UAVENV_LIB=$PWD/tools/_exp/lib_$n.so rocprofv3 --pmc $SET --output-format csv -d $D -- python3 tools/step_driver.py $E 200 > /dev/null 2>&1
  done
  python3 - "$n" gpurun_out/prof_mix_${n}_1 gpurun_out/prof_mix_${n}_2 <<'PY'
import csv, glob, sys, collections
tot = {}
for d in sys.argv[2:]:
    acc = collections.defaultdict(list)
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "uav_step_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    w = sum(acc["SQ_WAVES"]) / max(1, len(acc["SQ_WAVES"]))
    for k, v in acc.items():
        if k != "SQ_WAVES":
            tot[k[9:]] = sum(v) / len(v) / w
g = lambda k: tot.get(k, 0.0)
f64 = g("VALU_FMA_F64") + g("VALU_ADD_F64") + g("VALU_MUL_F64")
f32 = g("VALU_FMA_F32") + g("VALU_ADD_F32") + g("VALU_MUL_F32")
tr = g("VALU_TRANS_F32") + g("VALU_TRANS_F64")
other = g("VALU") - g("VALU_INT32") - g("VALU_INT64") - g("VALU_CVT") - f64 - f32 - tr
print("%-8s VALU %6.1f = int32 %5.1f + int64 %4.1f + cvt %4.1f + f64 %5.1f + f32 %5.1f + trans %4.1f + other %6.1f   | SALU %6.1f SMEM %4.1f" % (
    sys.argv[1], g("VALU"), g("VALU_INT32"), g("VALU_INT64"), g("VALU_CVT"), f64, f32, tr, other, g("SALU"), g("SMEM")))
PY
done
