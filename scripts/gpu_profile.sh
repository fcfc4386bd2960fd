#!/bin/bash
# rocprofv3 kernel-trace + stats of the benchmark command; summaries land in gpurun_out/ (copy to profiles/)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
TAG=${TAG:-r01}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o $TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" > gpurun_out/prof_bench_$TAG.log 2>&1
tail -2 gpurun_out/prof_bench_$TAG.log | cut -c1-600
find gpurun_out/prof -name "*stats*" | head
F=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -25 "$F"
