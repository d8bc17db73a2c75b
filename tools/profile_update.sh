#!/bin/bash
# GPU box: per-kernel table of a DQN training run at the reference's update ratio (one update per 16 transitions, 4096 x 50):
# rocprofv3 kernel trace of examples/train_dqn.py, summary -> gpurun_out/profiles/<tag>_update_kernel_stats.md.
# Usage: tools/profile_update.sh <tag> [train_dqn.py arguments].  The raw trace goes to /tmp on the box (hundreds of MB; gpurun_out merges back at most 64 MiB).
set -e
TAG=${1:-r03b}; shift || true
ARGS=${@:---timesteps 400000 --updates-per-transition 0.0625 --reward-scale 0.001}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/profiles; mkdir -p $OUT
rm -rf /tmp/prof_update
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_update -- python3 examples/train_dqn.py $ARGS > $OUT/${TAG}_update_run.json 2> $OUT/${TAG}_update_rocprof.err || { tail -20 $OUT/${TAG}_update_rocprof.err; exit 1; }
python3 tools/parse_rocprof.py stats /tmp/prof_update $OUT/${TAG}_update_kernel_stats.md
head -${LINES_SHOWN:-12} $OUT/${TAG}_update_kernel_stats.md | cut -c1-200
