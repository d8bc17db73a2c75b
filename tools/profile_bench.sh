#!/bin/bash
# GPU box: rocprofv3 passes over bench.py; raw output under gpurun_out/prof_*, summaries under gpurun_out/profiles/
# (copy the summaries you want judged into profiles/).  Usage: tools/profile_bench.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/profiles; mkdir -p $OUT
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline "$@" > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_rocprof_stats.err || { tail -20 $OUT/${TAG}_rocprof_stats.err; exit 1; }
python3 tools/parse_rocprof.py stats gpurun_out/prof_stats $OUT/${TAG}_kernel_stats.md
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline "$@" > /dev/null 2> $OUT/${TAG}_rocprof_fetch.err || { tail -20 $OUT/${TAG}_rocprof_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline "$@" > /dev/null 2> $OUT/${TAG}_rocprof_write.err || { tail -20 $OUT/${TAG}_rocprof_write.err; exit 1; }
python3 tools/parse_rocprof.py pmc gpurun_out/prof_fetch gpurun_out/prof_write "4096x50@500" $OUT/${TAG}_traffic.json
ls gpurun_out/prof_stats | head; find gpurun_out/prof_stats -name '*.csv' | head
