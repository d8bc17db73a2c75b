#!/bin/bash
# Build container + GPU box: A/B timing of attention-kernel variants (csrc/uavenv_attention.hip).  Each "name=flags" argument
# is compiled HERE into tools/_exp/lib_<name>.so, then timed on an MI355X (tools/attn_time.py: HIP-graph replays of 40 launches,
# batch 16 / 256 / 4096 / 16384).     tools/attn_exp.sh base= exit0=-DATTN_EXIT=0 ...
set -e
cd "$(dirname "$0")/.."
PKG="./-reinforcement-learning-for-dynamic-uav-energy-efficient-path-planning-in-iot-sensor-networks._amd"
mkdir -p tools/_exp
rm -f tools/_exp/lib_*.so
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -Wno-bitwise-instead-of-logical -Wno-unused-variable -Wno-unused-but-set-variable -mllvm -amdgpu-kernarg-preload-count=8"
names=""
for v in "$@"; do
  n="${v%%=*}"; f="${v#*=}"
  names="$names $n"
  ( /opt/rocm/bin/hipcc $FLAGS $f -o tools/_exp/lib_$n.so "$PKG/csrc/uavenv_kernels.hip" "$PKG/csrc/uavenv_capi.hip" "$PKG/csrc/uavenv_attention.hip" "$PKG/csrc/uavenv_replay.hip" "$PKG/csrc/uavenv_learner.hip" 2> tools/_exp/build_$n.log || echo "BUILD FAILED $n" ) &
done
wait
grep -l "error" tools/_exp/build_*.log 2>/dev/null && { grep -h "error" tools/_exp/build_*.log | head; exit 1; }
tools/gpu.sh --timeout 600 "python3 tools/attn_time.py $names"
