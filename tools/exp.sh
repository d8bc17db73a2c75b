#!/bin/bash
# Build container + GPU box: A/B timing of step-kernel variants.  Each "name=flags" argument is compiled HERE (hipcc
# cross-compiles) into gpurun_out-free tools/_exp/lib_<name>.so (they travel with the snapshot), then every variant is
# timed on an MI355X at 256 and 4096 environments x 50 sensors (uavenv_time_steps, 3 x 1000 launches each).
#   tools/exp.sh base= nostate=-DUAV_ABL_NOSTATE head=@/tmp/lib_head.so ...
set -e
cd "$(dirname "$0")/.."
PKG="./-reinforcement-learning-for-dynamic-uav-energy-efficient-path-planning-in-iot-sensor-networks._amd"
mkdir -p tools/_exp
rm -f tools/_exp/lib_*.so
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -Wno-bitwise-instead-of-logical -Wno-unused-variable -mllvm -amdgpu-kernarg-preload-count=8"
names=""
for v in "$@"; do
  n="${v%%=*}"; f="${v#*=}"
  names="$names $n"
  if [ "${f#@}" != "$f" ]; then cp "${f#@}" tools/_exp/lib_$n.so; continue; fi     # name=@/path/lib.so: a library built earlier
  ( /opt/rocm/bin/hipcc $FLAGS $f -o tools/_exp/lib_$n.so "$PKG/csrc/uavenv_kernels.hip" "$PKG/csrc/uavenv_capi.hip" "$PKG/csrc/uavenv_attention.hip" "$PKG/csrc/uavenv_replay.hip" "$PKG/csrc/uavenv_learner.hip" 2> tools/_exp/build_$n.log || echo "BUILD FAILED $n" ) &
done
wait
grep -l "error" tools/_exp/build_*.log 2>/dev/null && { grep -h "error" tools/_exp/build_*.log | head; exit 1; }
tools/gpu.sh --timeout 600 "${EXP_CMD:-python3 tools/exp_time.py} $names"
