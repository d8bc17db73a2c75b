#!/bin/bash
# Build container: compile csrc/uavenv_kernels.hip to gfx950 assembly and list, for one kernel instantiation, every scalar
# load and every "s_waitcnt lgkmcnt" with its line number (a load directly followed by its wait is one serialized round trip).
#   tools/isa_sloads.sh [extra hipcc flags]        -> /tmp/isa/k_t.s + the listing on stdout
set -e
cd "$(dirname "$0")/.."
PKG="./-reinforcement-learning-for-dynamic-uav-energy-efficient-path-planning-in-iot-sensor-networks._amd"
mkdir -p /tmp/isa
[ -n "$NOBUILD" ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -Wno-bitwise-instead-of-logical -Wno-unused-variable \
  -mllvm -amdgpu-kernarg-preload-count=8 "$@" -S --cuda-device-only -o /tmp/isa/k_t.s "$PKG/csrc/uavenv_kernels.hip"
K="${KERNEL:-_ZN6uavenv15uav_step_kernelILi64ELb1ELi16ELb1EE}"
awk -v k="$K" '
  index($0, k) && /^_Z.*:/ {on=1; start=NR}
  on && /^\.Lfunc_end/ {print "lines " start "-" NR; on=0}
  on && (/s_load|s_waitcnt.*lgkmcnt|s_buffer_load/) {print NR-start ": " $0}
' /tmp/isa/k_t.s
grep -A60 "\.amdhsa_kernel $K" /tmp/isa/k_t.s | grep -E "next_free_vgpr|next_free_sgpr|private_segment_fixed" || true
