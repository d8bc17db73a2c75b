#!/bin/bash
# Build container: stamp the tree's commit into .build_commit (git-ignored; it travels to the GPU box, which has no .git:
# bench.py and tools/parse_rocprof.py copy it into what they write, so profiles say which code they measured), then
# hand the command to gpurun.   Usage: tools/gpu.sh [--timeout S] '<command run on the MI355X box>'
cd "$(dirname "$0")/.."
T=900
if [ "$1" = "--timeout" ]; then T=$2; shift 2; fi
c=$(git rev-parse --short=12 HEAD)
if ! git diff --quiet HEAD -- . ':!gpurun_out'; then c="$c+dirty"; fi
echo "$c" > .build_commit
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
